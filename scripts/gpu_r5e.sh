# round 5, GPU call 5: clean A/B of the contracted sweep variants (round-4 kernel = fast_old); 2D hydro fold; slab probe
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5e
O=gpurun_out/r5e
for rep in 1 2; do
for lib in librgpu_fast.so librgpu_exp_fast_old.so librgpu_exp_fast_alf.so librgpu_exp_fast_e.so librgpu_exp_fast_alf_e.so; do
  ( echo "== $lib"; RGPU_LIB=$PWD/ramsesgpu_amd/$lib python scripts/probe_sweep.py mhd_mri_3d 512 10 2>&1 | grep -v amdgpu ) >> $O/sweep_ab.log 2>&1
done; done
( echo "== contracted"; RGPU_ARITH=contracted python scripts/probe_2d.py 2>&1 | grep -v amdgpu; echo "== exact"; RGPU_ARITH=exact python scripts/probe_2d.py 2>&1 | grep -v amdgpu ) > $O/probe_2d.log 2>&1
( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" ) > $O/probe64.log 2>&1
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -40 ) > $O/timeline1.txt 2>&1
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_comm_device.py tests/test_bench_contract.py tests/test_kernel_resources.py -x -q -m gpu -k "run_steps or comm or resources or bench" 2>&1 | tail -12 ) > $O/tests.log 2>&1
cat $O/sweep_ab.log $O/probe_2d.log $O/probe64.log $O/tests.log
