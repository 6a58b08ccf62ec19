# run-to-run distribution of the 256^3 hydro implosion step (contracted / exact): eight processes each
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y; mkdir -p $O; rm -f $O/dist.log
for i in 1 2 3 4 5 6 7 8; do
  for a in contracted exact; do
    RGPU_ARITH=$a python scripts/probe_sweep.py implode3d 256 100 2>&1 | grep -v amdgpu.ids | tr '\n' ' ' >> $O/dist.log; echo >> $O/dist.log
  done
done
cat $O/dist.log
