cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5l
O=gpurun_out/r5l
for s in 1 2; do
( RGPU_COMM_EMULATE_MODE=parallel RGPU_COMM_SCHEDULE=$s RGPU_ARITH=contracted PROBE_LINK_GBPS="80 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s/" ) >> $O/probe_parallel.log 2>&1
( RGPU_COMM_EMULATE_MODE=parallel RGPU_COMM_SCHEDULE=$s RGPU_ARITH=contracted PROBE_LINK_GBPS="60 40" PROBE_NZ=128 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s/" ) >> $O/probe_parallel.log 2>&1
( RGPU_COMM_SCHEDULE=$s RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s/" ) >> $O/probe_serial_again.log 2>&1
done
( RGPU_COMM_EMULATE_MODE=parallel RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -36 ) > $O/timeline_s1_parallel.txt 2>&1
cat $O/probe_parallel.log $O/probe_serial_again.log
