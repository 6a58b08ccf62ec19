cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "near_ties or longer_than_one" 2>&1 | tail -4 ) > gpurun_out/r4t/tests.log 2>&1
cat gpurun_out/r4t/tests.log
