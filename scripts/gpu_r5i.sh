cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
O=gpurun_out/r5i
P="RGPU_ARITH=contracted PROBE_LINK_GBPS=0 PROBE_NZ=64"
run() { ( env $P "$@" python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/$LABEL/" ) >> $O/probe.log 2>&1; }
for s in 1 2; do
  LABEL="s$s host       " run RGPU_COMM_SCHEDULE=$s PROBE_HOST_LOOP=1
  LABEL="s$s batch10    " run RGPU_COMM_SCHEDULE=$s
  LABEL="s$s batch5     " run RGPU_COMM_SCHEDULE=$s PROBE_BATCH=5
  LABEL="s$s batch2     " run RGPU_COMM_SCHEDULE=$s PROBE_BATCH=2
  LABEL="s$s batch1     " run RGPU_COMM_SCHEDULE=$s PROBE_BATCH=1
  LABEL="s$s batch10 notiming" run RGPU_COMM_SCHEDULE=$s RGPU_COMM_NO_TIMING=1
  LABEL="s$s host notiming   " run RGPU_COMM_SCHEDULE=$s RGPU_COMM_NO_TIMING=1 PROBE_HOST_LOOP=1
  LABEL="s$s batch10 nopack  " run RGPU_COMM_SCHEDULE=$s RGPU_COMM_PACK=0
  LABEL="s$s host nopack     " run RGPU_COMM_SCHEDULE=$s RGPU_COMM_PACK=0 PROBE_HOST_LOOP=1
done
cat $O/probe.log
