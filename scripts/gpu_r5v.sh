# prologue of the MHD sweep: the three plane loads in flight at once (product build) against the serial prologue (-DRG_SERIAL_PROLOGUE, librgpu_xserial*.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; mkdir -p $O; rm -f $O/ab.log
for rep in 1 2 3; do
  for v in parallel serial; do
    for a in contracted exact; do
      lib=""; [ $v = serial ] && { [ $a = contracted ] && lib=ramsesgpu_amd/librgpu_xserial_fast.so || lib=ramsesgpu_amd/librgpu_xserial.so; }
      for nz in 64 512; do
        echo "== $v $a nz=$nz rep=$rep" >> $O/ab.log
        RGPU_LIB=$lib RGPU_ARITH=$a PROBE_NZ=$nz python scripts/probe_sweep.py mhd_mri_3d 512 10 2>&1 | grep "phases" >> $O/ab.log
      done
    done
  done
done
cat $O/ab.log
