cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4r
export RGPU_ARITH=contracted
( PROBE_NZ=64 PROBE_LINK_GBPS="0 60" timeout 600 python scripts/slab_probe.py 2>&1 | grep "nz=\|rror"
  PROBE_NZ=128 PROBE_LINK_GBPS="0" timeout 600 python scripts/slab_probe.py 2>&1 | grep "nz=\|rror" ) > gpurun_out/r4r/probe.log 2>&1
( timeout 600 python scripts/probe_sweep.py mhd_mri_3d 512 10 2>&1 | grep -v amdgpu; timeout 600 python scripts/probe_sweep.py implode3d 256 50 2>&1 | grep -v amdgpu ) > gpurun_out/r4r/sweep.log 2>&1
bash scripts/slab_timeline.sh 1 60 > gpurun_out/r4r/tl.log 2>&1
unset RGPU_ARITH
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_comm_driver.py -x -q -m gpu -k "not whole_box and not long_runs" 2>&1 | tail -5 ) > gpurun_out/r4r/tests.log 2>&1
cat gpurun_out/r4r/probe.log gpurun_out/r4r/sweep.log gpurun_out/r4r/tests.log; grep -A34 "the last 56" gpurun_out/r4r/tl.log | cut -c1-100 | sed -n 14,36p
