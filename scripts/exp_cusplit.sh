#!/bin/bash
# experiment: CU partition between the VALU-bound and HBM-bound streams, LDS occupancy cap of the Riemann kernels
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1),'Mcell/s', round(d['ms_per_step'],2),'ms')"; }
run X=1
for n in 160 176 192 208 224; do run RGPU_CU_SPLIT=$n; done
for l in 16384 24576 32768 40960 65536; do run RGPU_ALU_LDS=$l; done
run RGPU_CU_SPLIT=192 RGPU_CHUNKS=32
run RGPU_CU_SPLIT=192 RGPU_CHUNKS=128
