# ghost planes behind slab interfaces not updated: N = 8 / N = 4 probe A/B (the device tests of the slab driver ran green before: gpurun_out/r5r/tests.log)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5r
rm -f gpurun_out/r5r/probe.log
for rep in 1 2 3; do
for nz in 64 128; do
  for v in new old; do
    if [ $v = old ]; then export RGPU_COMM_UPDATE_GHOST_PLANES=1; else unset RGPU_COMM_UPDATE_GHOST_PLANES; fi
    echo "== $v nz=$nz rep=$rep" >> gpurun_out/r5r/probe.log
    PROBE_NZ=$nz PROBE_LINK_GBPS="0 60" python scripts/slab_probe.py 2>&1 | grep "ms/step" >> gpurun_out/r5r/probe.log
  done
done
done
unset RGPU_COMM_UPDATE_GHOST_PLANES
cat gpurun_out/r5r/probe.log
