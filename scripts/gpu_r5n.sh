cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5n
O=gpurun_out/r5n
run() { ( env RGPU_ARITH=contracted PROBE_NZ=64 "$@" python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/$LABEL/" ) >> $O/probe.log 2>&1; }
for s in 1 2; do for pk in 1 0; do
  LABEL="s$s pack=$pk" run RGPU_COMM_SCHEDULE=$s RGPU_COMM_PACK=$pk PROBE_LINK_GBPS="0"
  LABEL="s$s pack=$pk" run RGPU_COMM_SCHEDULE=$s RGPU_COMM_PACK=$pk PROBE_LINK_GBPS="60" RGPU_COMM_EMULATE_MODE=parallel
done; done
cat $O/probe.log
