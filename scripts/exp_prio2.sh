#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline --timeline-only 2>/dev/null | tail -1; }
run X=1
run RGPU_ALU_PRIO=1
run RGPU_ALU_PRIO=-1
run RGPU_CHUNKS=1
