# round 5, GPU call 4: clock fold (2D kernels, rotating-path sweep), spill-free contracted sweep, Alfven selection in the contracted build
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5d
O=gpurun_out/r5d
( time timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_comm_device.py tests/test_contracted.py tests/test_kernel_resources.py -x -q -m gpu -k "run_steps or alfven or golden or comm or contracted or resources or fused" 2>&1 | tail -12 ) > $O/tests.log 2>&1
# sweep A/B at 512^3, contracted arithmetic: default (decodes per plane, no spill) / old (hoisted decodes, 6 spills) / + Alfven selection / + E decode per plane
for lib in librgpu_fast.so librgpu_exp_fast_old.so librgpu_exp_fast_alf.so librgpu_exp_fast_e.so librgpu_fast.so librgpu_exp_fast_old.so librgpu_exp_fast_alf.so; do
  ( echo "== $lib"; RGPU_LIB=$PWD/ramsesgpu_amd/$lib python scripts/probe_sweep.py mhd_mri_3d 512 10 2>&1 | grep -v amdgpu ) >> $O/sweep_ab.log 2>&1
done
( echo "== librgpu.so (exact)"; python scripts/probe_sweep.py mhd_mri_3d 512 10 2>&1 | grep -v amdgpu ) >> $O/sweep_ab.log 2>&1
# 2D: the clock folded into the step kernel against the separate clock kernel
for a in contracted exact; do
  ( echo "== $a, fold"; RGPU_ARITH=$a python scripts/probe_2d.py 2>&1 | grep -v amdgpu; echo "== $a, RGPU_NO_CLOCK_FOLD=1"; RGPU_NO_CLOCK_FOLD=1 RGPU_ARITH=$a python scripts/probe_2d.py 2>&1 | grep -v amdgpu ) >> $O/probe_2d.log 2>&1
done
( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" ) > $O/probe64.log 2>&1
( RGPU_NO_CLOCK_FOLD=1 RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/nofold /" ) >> $O/probe64.log 2>&1
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -40 ) > $O/timeline1.txt 2>&1
cat $O/tests.log $O/sweep_ab.log $O/probe_2d.log $O/probe64.log
