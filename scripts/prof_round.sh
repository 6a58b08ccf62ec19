#!/bin/bash
# Profile run on the GPU box (through gpurun): bench line, rocprofv3 kernel stats (default two-stream schedule
# and serial schedule), three separate --pmc passes.  Results land in gpurun_out/prof_$TAG; summarise with
# scripts/summarize_prof.py into profiles/.
TAG=${1:-r01b}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
BENCH="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -o bench -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace_default.err
RGPU_CHUNKS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -o bench -- $BENCH > $OUT/bench_serial_under_rocprof.json 2> $OUT/trace_serial.err
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
export RGPU_CHUNKS=1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq.err
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write.err
rm -f $OUT/*/bench_kernel_trace.csv     # per-dispatch rows: large, not needed
du -sh $OUT; cat $OUT/bench.json | cut -c1-300
