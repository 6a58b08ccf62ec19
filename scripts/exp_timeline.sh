#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/timeline; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o b -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --timeline-only > /dev/null 2> $OUT/err.txt
python $R/scripts/timeline.py $(find $OUT/tr -name "*kernel_trace.csv" | head -1) 0.5 | tee $OUT/timeline.txt
rm -rf $OUT/tr
