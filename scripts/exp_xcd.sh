#!/bin/bash
# experiment: XCD-aware workgroup order (sub-band size in cells; 0 = linear order)
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1),'Mcell/s', round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['roofline_step']['phase_ms'].items()})"; }
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for s in 0 1024 2048 4096 8192 16384 32768; do run RGPU_XCD_SUB=$s; done
run RGPU_XCD_SUB=4096 RGPU_CHUNKS=32
run RGPU_XCD_SUB=4096 RGPU_CHUNKS=16
python scripts/gpu_probe.py --no-parity 256 2>&1 | grep -v "^ " | tail -4
