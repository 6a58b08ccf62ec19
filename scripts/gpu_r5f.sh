# round 5, GPU call 6: two-range sweep (schedule 2), link emulation beside the local copy, adopted contracted sweep; timelines of schedule 2
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5f
O=gpurun_out/r5f
( time timeout 2400 python -m pytest tests/test_comm_device.py tests/test_contracted.py tests/test_kernel_resources.py tests/test_comm_driver.py -x -q -m gpu 2>&1 | tail -12 ) > $O/tests.log 2>&1
for mode in serial parallel; do for s in 1 2; do
  ( RGPU_COMM_EMULATE_MODE=$mode RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=$s PROBE_LINK_GBPS="0 80 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s/" ) >> $O/probe64.log 2>&1
done; done
( RGPU_NO_SWEEP_PAIR=1 RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=2 PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched 2, two sweep launches/" ) >> $O/probe64.log 2>&1
( PROBE_HOST_LOOP=1 RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=2 PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched 2, host loop/" ) >> $O/probe64.log 2>&1
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 2 60 2>&1 | tail -50 ) > $O/timeline2.txt 2>&1
( RGPU_COMM_EMULATE_MODE=parallel RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -40 ) > $O/timeline1_parallel.txt 2>&1
( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60" python scripts/slab_probe.py 2>&1 | grep "^nz" ) > $O/probe_all.log 2>&1
cat $O/tests.log $O/probe64.log $O/probe_all.log
