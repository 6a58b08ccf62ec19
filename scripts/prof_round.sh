#!/bin/bash
# Profile run on the GPU box (through gpurun): bench lines, rocprofv3 kernel stats, three separate --pmc passes -- for the
# headline workload (512^3 MRI) and for the 256^3 hydro implosion.  Results land in gpurun_out/prof_$TAG; summarise with
# scripts/summarize_prof.py $TAG into profiles/.
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# which state of the kernels the counters below belong to (bench.py compares it before quoting pmc_traffic.json)
python -c "import sys; sys.path.insert(0, '$R'); from ramsesgpu_amd import build as rb; print(rb.kernel_source_hash())" > $OUT/kernel_source_sha.txt
python $R/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
python $R/bench.py --workload implode3d --steps 100 --warmup 10 > $OUT/bench_implode3d.json 2> $OUT/bench_implode3d.err
python $R/bench.py --workload orszag-tang --steps 50 --warmup 5 > $OUT/bench_orszag-tang.json 2> $OUT/bench_orszag-tang.err
for W in mri_contracted mri implode3d_contracted implode3d orszag-tang_contracted orszag-tang; do
  WL=${W%_contracted}; AR="--arith exact"
  if [ $W != $WL ]; then AR="--arith contracted"; fi
  if [ $WL = mri ]; then ST="--steps 10 --warmup 2"; PST="--steps 2 --warmup 1"; elif [ $WL = implode3d ]; then ST="--steps 50 --warmup 5"; PST="--steps 5 --warmup 2"; else ST="--steps 200 --warmup 20"; PST="--steps 20 --warmup 5"; fi
  BENCH="python $R/bench.py --workload $WL $AR $ST --no-cpu-baseline --no-second-arith --no-other-workloads"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$W -o bench -- $BENCH > $OUT/bench_${W}_under_rocprof.json 2> $OUT/trace_$W.err
  CMD="python $R/bench.py --workload $WL $AR $PST --no-cpu-baseline --no-second-arith --no-other-workloads"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_sq_$W -o pmc -- $CMD > /dev/null 2> $OUT/pmc_sq_$W.err
  rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_fetch_$W -o pmc -- $CMD > /dev/null 2> $OUT/pmc_fetch_$W.err
  rocprofv3 --pmc WRITE_SIZE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $OUT/pmc_write_$W -o pmc -- $CMD > /dev/null 2> $OUT/pmc_write_$W.err
done
rm -f $OUT/*/bench_kernel_trace.csv $OUT/*/*/bench_kernel_trace.csv    # per-dispatch rows: large, not needed
du -sh $OUT; cut -c1-300 $OUT/bench.json
