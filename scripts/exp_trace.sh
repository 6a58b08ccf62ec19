#!/bin/bash
# experiment: trace kernel launch shape (waves/SIMD cap, workgroup size), serial vs chunked schedule
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1),'Mcell/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['roofline_step']['phase_ms'].items()})"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run X=1
run RGPU_CHUNKS=1
for v in t4 t4b128 t1b128 t1b512; do run RGPU_LIB=$PWD/build/librgpu_$v.so; done
python scripts/gpu_probe.py --no-parity 256 2>&1 | grep -v "^ " | tail -8
