cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/toggles
( RGPU_COMM_PACK=0 timeout 900 python -m pytest tests/test_comm_driver.py -x -q -m gpu -k "not whole_box" 2>&1 | tail -3
  RGPU_NO_STEP_CLOCK=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_steps" 2>&1 | tail -3
  RGPU_COMM_SCHEDULE=2 timeout 900 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -3 ) > gpurun_out/toggles/tests.log 2>&1
cat gpurun_out/toggles/tests.log
