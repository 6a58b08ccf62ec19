#!/bin/bash
cd "$(dirname "$0")/.."
for s in 0 2048 4096 8192; do echo "== RGPU_XCD_SUB=$s"; RGPU_XCD_SUB=$s python scripts/gpu_probe.py --no-parity 256 2>&1 | grep "\^3" ; done
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1),'Mcell/s', round(d['ms_per_step'],2),'ms')"; }
run RGPU_XCD_SUB=4096
run RGPU_XCD_SUB=3072
