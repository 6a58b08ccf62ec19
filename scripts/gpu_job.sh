#!/bin/bash
# One parameterised GPU job (through gpurun) instead of a script per session.  Steps, in the order given:
#   variants        parity at the bench's launch geometry + 512^3 MRI timing of librgpu.so, librgpu_fast.so and every librgpu_exp_*.so
#                   (scripts/exp_variants.sh; EXP_NO_PARITY=1 skips the parity half)
#   golden          tests/test_gpu_parity.py -k golden and tests/test_contracted.py
#   suite           the whole GPU suite (pytest -m gpu)
#   bench           bench.py at its defaults
#   prof TAG        scripts/prof_round.sh TAG (bench lines, rocprofv3 kernel stats, PMC passes)
#   final           what the driver runs at the end of a round: the whole GPU suite, the smoke check, the default bench line
#   toggles         parts of the suite with the environment / option switches of include/rgpu.h set the other way
#   cmd "..."       an arbitrary command
# Output: gpurun_out/$JOB_OUT (default job).   usage: gpurun -- 'JOB_OUT=r6a bash scripts/gpu_job.sh variants golden'
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
cd $R
OUT=$R/gpurun_out/${JOB_OUT:-job}; mkdir -p $OUT
while [ $# -gt 0 ]; do
  step=$1; shift
  echo "##### $step  $(date +%T)" | tee -a $OUT/log.txt
  case $step in
    variants) EXP_OUT=${JOB_OUT:-job} bash scripts/exp_variants.sh > $OUT/variants.log 2>&1; cat $OUT/summary.txt | tee -a $OUT/log.txt ;;
    golden) ( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k golden 2>&1 | tail -3
              timeout 1200 python -m pytest tests/test_contracted.py -x -q -m gpu 2>&1 | tail -3 ) | tee -a $OUT/log.txt ;;
    suite) timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 > $OUT/suite.log 2>&1; tail -40 $OUT/suite.log | tee -a $OUT/log.txt ;;
    bench) python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json | tee -a $OUT/log.txt ;;
    prof) tag=$1; shift; bash scripts/prof_round.sh $tag 2>&1 | tail -5 | tee -a $OUT/log.txt ;;
    final) ( time timeout 2700 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -30 ) > $OUT/final_tests.log 2>&1; tail -4 $OUT/final_tests.log | tee -a $OUT/log.txt
           ( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) 2>&1 | tail -3 | tee -a $OUT/log.txt
           python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json | tee -a $OUT/log.txt ;;
    toggles) ( RGPU_COMM_PACK=0 timeout 900 python -m pytest tests/test_comm_driver.py -x -q -m gpu -k "not whole_box" 2>&1 | tail -2
               RGPU_COMM_ONE_STREAM=1 timeout 900 python -m pytest tests/test_comm_driver.py -x -q -m gpu -k "not whole_box" 2>&1 | tail -2
               RGPU_TEST_OPTIONS=step_clock=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_steps" 2>&1 | tail -2
               RGPU_TEST_OPTIONS=ghost_images=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden" 2>&1 | tail -2
               RGPU_COMM_SCHEDULE=2 timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -2 ) 2>&1 | tee -a $OUT/log.txt ;;
    cmd) c=$1; shift; ( eval "$c" ) 2>&1 | tee -a $OUT/log.txt ;;
    *) echo "unknown step $step" | tee -a $OUT/log.txt ;;
  esac
done
echo "##### done $(date +%T)" | tee -a $OUT/log.txt
