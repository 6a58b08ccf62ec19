#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1),'Mcell/s', round(d['ms_per_step'],2),'ms')"; }
run X=1
run RGPU_LIB=$PWD/build/librgpu_fuse.so
run RGPU_LIB=$PWD/build/librgpu_fuse.so RGPU_CHUNKS=32
run RGPU_LIB=$PWD/build/librgpu_fuse.so RGPU_CHUNKS=128
run RGPU_CHUNKS=128
run RGPU_CHUNKS=256
