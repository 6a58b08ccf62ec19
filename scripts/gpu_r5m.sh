# round 5: validation on the final kernels -- the whole GPU suite, then the profile round (kernel stats + PMC passes, profiles/r05_*)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5m
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r5m/tests.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5m/smoke.log 2>&1
bash scripts/prof_round.sh r05 > gpurun_out/r5m/prof.log 2>&1
cat gpurun_out/r5m/tests.log gpurun_out/r5m/smoke.log; tail -3 gpurun_out/r5m/prof.log
