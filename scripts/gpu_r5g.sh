# round 5, GPU call 7: why is the batched loop slower than the host loop on thin slabs?  kernel timelines of both, schedules 1 and 2, no link hold
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5g
O=gpurun_out/r5g
for s in 2 1; do
  ( RGPU_ARITH=contracted bash scripts/slab_timeline.sh $s 0 2>&1 | tail -34 ) > $O/timeline_s${s}_batch.txt 2>&1
  ( PROBE_HOST_LOOP=1 RGPU_ARITH=contracted bash scripts/slab_timeline.sh $s 0 2>&1 | tail -34 ) > $O/timeline_s${s}_host.txt 2>&1
done
for rep in 1 2; do for s in 1 2; do
  ( RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=$s PROBE_LINK_GBPS="0" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s batch/" ) >> $O/probe64.log 2>&1
  ( PROBE_HOST_LOOP=1 RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=$s PROBE_LINK_GBPS="0" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s host /" ) >> $O/probe64.log 2>&1
done; done
cat $O/probe64.log; tail -30 $O/timeline_s2_host.txt
