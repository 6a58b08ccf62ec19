set -x
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/prof_r01; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_csv -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/trace_csv.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20; du -sh $OUT
