# pack kernel on the compute stream (RGPU_COMM_PACK_STREAM=compute) against the halo stream: N = 8 slab probe, schedule 1, contracted
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5z; mkdir -p $O; rm -f $O/ab.log
for rep in 1 2 3; do
  for v in halo compute; do
    export RGPU_COMM_PACK_STREAM=$v
    echo "== pack on $v rep=$rep" >> $O/ab.log
    RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=1 PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" >> $O/ab.log
    RGPU_COMM_EMULATE_MODE=parallel RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=1 PROBE_LINK_GBPS="60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" >> $O/ab.log
  done
done
cat $O/ab.log
