# last x face column in 2 x 32 tiles (MhLastX): 3D MHD parity (GPU parity + contracted + slab device tests) and A/B of the 512^3 sweep
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_contracted.py -x -q -m gpu -k "mhd or mri or 3d or MHD or orszag or alfven or run_steps or bench" 2>&1 | tail -8 ) > $O/tests.log 2>&1
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export RGPU_NO_LASTX_TILES=1; else unset RGPU_NO_LASTX_TILES; fi
    for a in contracted exact; do
      echo "== $v $a rep=$rep" >> $O/ab.log
      RGPU_ARITH=$a python scripts/probe_sweep.py mhd_mri_3d 512 10 2>&1 | grep -v amdgpu.ids | tail -6 >> $O/ab.log
    done
  done
done
unset RGPU_NO_LASTX_TILES
cat $O/tests.log $O/ab.log
