# round 5, GPU call 2: whole GPU suite on the fused ghost fill / pair launches; slab probe + timeline of the N = 8 slab
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5b
( time timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > gpurun_out/r5b/tests.log 2>&1
( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60 40" python scripts/slab_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5b/probe.log 2>&1
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -70 ) > gpurun_out/r5b/timeline1.txt 2>&1
cat gpurun_out/r5b/tests.log; cat gpurun_out/r5b/probe.log
