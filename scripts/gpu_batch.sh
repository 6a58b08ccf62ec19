cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
( time timeout 2700 python -m pytest tests -q -m gpu --durations=30 2>&1 | tail -45 ) > gpurun_out/r4j/tests.log 2>&1
cat gpurun_out/r4j/tests.log
