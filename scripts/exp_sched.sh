#!/bin/bash
# schedule experiments at 512^3 with the specialised kernels: chunk counts, LDS pad capping the Riemann occupancy
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
run() { env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --timeline-only 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$*', 'ms/step %.2f' % d['ms_per_step'])"; }
run A=0
run RGPU_CHUNKS=1
run RGPU_CHUNKS=32
run RGPU_CHUNKS=128
run RGPU_HEAVY_LDS=40000
run RGPU_HEAVY_LDS=40000 RGPU_CHUNKS=32
run RGPU_HEAVY_LDS=40000 RGPU_CHUNKS=128
run RGPU_HEAVY_LDS=27000
run RGPU_XCD_SUB=8192
run RGPU_XCD_SUB=2048
run A=1
