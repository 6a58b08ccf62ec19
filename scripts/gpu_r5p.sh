# round 5, last validation on the frozen kernel sources: whole GPU suite, smoke, profile round (profiles/r05_*), default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r5p/tests.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5p/smoke.log 2>&1
bash scripts/prof_round.sh r05 > gpurun_out/r5p/prof.log 2>&1
cat gpurun_out/r5p/tests.log gpurun_out/r5p/smoke.log; tail -3 gpurun_out/r5p/prof.log | cut -c1-300
