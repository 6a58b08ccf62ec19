P="timeout 180 python scripts/probe_sweep.py mhd_mri_3d 512 10"
for rep in 1 2; do
for lib in librgpu_fast librgpu_exp_f_p1 librgpu_exp_f_p2 librgpu_exp_f_p3 librgpu_exp_f_p4 librgpu librgpu_exp_p3 librgpu_exp_p4; do
  echo "== $lib"; RGPU_LIB=$PWD/ramsesgpu_amd/$lib.so $P 2>&1 | grep -E "phases"
done; done
