cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
( time timeout 1800 python -m pytest tests/test_comm_driver.py tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4p/tests.log 2>&1
cat gpurun_out/r4p/tests.log
