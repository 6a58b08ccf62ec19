cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
( timeout 1500 python -m pytest tests/test_contracted.py -q -k "test_bench_launch_geometry_within_tolerance" 2>&1 | tail -5
  timeout 1500 python -m pytest tests/test_comm_driver.py -q -m gpu -k "config5_whole_box" -s 2>&1 | tail -12
  timeout 900 python -m pytest tests/test_bench_contract.py -q -k "both_slab_drivers or bench_line_single_gpu" 2>&1 | tail -5
  timeout 300 python -m pytest tests/test_kernel_resources.py -q 2>&1 | tail -3 ) > gpurun_out/r4a/tests.log 2>&1
EXP_OUT=r4a bash scripts/exp_variants.sh > gpurun_out/r4a/exp.log 2>&1
tail -5 gpurun_out/r4a/tests.log
