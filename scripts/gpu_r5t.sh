# round 5, validation on the final kernel sources (MhLastX launch in): whole GPU suite, smoke, profile round (profiles/r05_*), N = 8 / 4 / 2 slab probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5t; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/tests.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > $O/smoke.log 2>&1
bash scripts/prof_round.sh r05 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
for s in 1 2; do ( RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=$s PROBE_LINK_GBPS="0 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s/" ) >> $O/probe.log 2>&1; done
( RGPU_COMM_EMULATE_MODE=parallel RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=1 PROBE_LINK_GBPS="60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched 1/" ) >> $O/probe.log 2>&1
for nz in 128 256; do ( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60" PROBE_NZ=$nz python scripts/slab_probe.py 2>&1 | grep "^nz" ) >> $O/probe.log 2>&1; done
( RGPU_ARITH=exact PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/exact  /" ) >> $O/probe.log 2>&1
cat $O/tests.log $O/smoke.log $O/probe.log; tail -3 $O/prof.log | cut -c1-300
