cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5j
O=gpurun_out/r5j
P="RGPU_ARITH=contracted PROBE_LINK_GBPS=0 PROBE_NZ=64"
run() { ( env $P "$@" python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/$LABEL/" ) >> $O/probe.log 2>&1; }
for s in 1 2; do
  LABEL="s$s host prio-high  " run RGPU_COMM_SCHEDULE=$s PROBE_HOST_LOOP=1
  LABEL="s$s batch prio-high " run RGPU_COMM_SCHEDULE=$s
  LABEL="s$s host prio-normal" run RGPU_COMM_SCHEDULE=$s PROBE_HOST_LOOP=1 RGPU_HALO_PRIO=normal
  LABEL="s$s batch prio-normal" run RGPU_COMM_SCHEDULE=$s RGPU_HALO_PRIO=normal
  LABEL="s$s batch prio-low  " run RGPU_COMM_SCHEDULE=$s RGPU_HALO_PRIO=low
  LABEL="s$s batch hwq8      " run RGPU_COMM_SCHEDULE=$s GPU_MAX_HW_QUEUES=8
  LABEL="s$s batch chunks1   " run RGPU_COMM_SCHEDULE=$s RGPU_CHUNKS=1
done
P="RGPU_ARITH=contracted PROBE_LINK_GBPS=60 PROBE_NZ=64"
for s in 1 2; do
  LABEL="s$s batch prio-high 60" run RGPU_COMM_SCHEDULE=$s
  LABEL="s$s batch prio-normal 60" run RGPU_COMM_SCHEDULE=$s RGPU_HALO_PRIO=normal
done
cat $O/probe.log
