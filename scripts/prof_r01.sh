set -x
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/prof_r01; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]+|TCC_[A-Z_0-9]+|FETCH_SIZE|WRITE_SIZE|GRBM_[A-Z_]+|TCP_[A-Z_0-9]+|VALU[A-Za-z]+|MemUnit[A-Za-z]+|L2CacheHit|LDSBankConflict)\b" | sort -u | tr '\n' ' ' > $OUT/counters_available.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
ls -R $OUT/trace | head -20
