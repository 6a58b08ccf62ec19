# what the builder runs on the GPU box before the round ends -- the same things the driver runs: the GPU suite, the smoke check, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
( time timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/final/tests.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > gpurun_out/final/smoke.log 2>&1
( time python bench.py 2>/dev/null | tail -1 ) > gpurun_out/final/bench.log 2>&1
cat gpurun_out/final/tests.log gpurun_out/final/smoke.log; cut -c1-600 gpurun_out/final/bench.log
