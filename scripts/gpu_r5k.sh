# round 5, GPU call: the slab probe table with the normal-priority halo stream (all N, both emulations, both schedules) + timelines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5k
O=gpurun_out/r5k
( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60 40" python scripts/slab_probe.py 2>&1 | grep "^nz" ) > $O/probe_serial.log 2>&1
( RGPU_COMM_EMULATE_MODE=parallel RGPU_ARITH=contracted PROBE_LINK_GBPS="80 60 40" python scripts/slab_probe.py 2>&1 | grep "^nz" ) > $O/probe_parallel.log 2>&1
( RGPU_ARITH=exact PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/exact s1/"; RGPU_COMM_SCHEDULE=2 RGPU_ARITH=exact PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/exact s2/" ) > $O/probe_exact.log 2>&1
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -40 ) > $O/timeline_s1.txt 2>&1
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 2 60 2>&1 | tail -44 ) > $O/timeline_s2.txt 2>&1
( RGPU_HALO_PRIO=normal GPU_MAX_HW_QUEUES=8 RGPU_ARITH=contracted PROBE_LINK_GBPS="0" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/hwq8+normal/" ) > $O/probe_hwq.log 2>&1
cat $O/probe_serial.log $O/probe_parallel.log $O/probe_exact.log $O/probe_hwq.log
