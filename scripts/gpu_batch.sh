cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
( time timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r4h/tests.log 2>&1
bash scripts/prof_round.sh r04 > gpurun_out/r4h/prof.log 2>&1
cd $GRAFT_REPO_ROOT
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4h/smoke.log 2>&1
tail -4 gpurun_out/r4h/tests.log; tail -3 gpurun_out/r4h/prof.log | cut -c1-400; cat gpurun_out/r4h/smoke.log
