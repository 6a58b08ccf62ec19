#!/bin/bash
# round-3 sweep experiments: each librgpu_exp_<tag>.so (built in the container with scripts/build_exp.py) is checked for parity
# on two MRI / 3D Orszag-Tang cases and timed at 512^3 (10 steps) with the phase timers.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/${EXP_OUT:-exp_r3}; mkdir -p $OUT
cd $R
for so in ramsesgpu_amd/librgpu.so $(ls ramsesgpu_amd/librgpu_exp_*.so 2>/dev/null); do
  tag=$(basename $so .so)
  echo "=== $tag" | tee -a $OUT/summary.txt
  RGPU_LIB=$R/$so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_medium_sizes_vs_oracle and (mhd_mri_3d or orszag-tang3d)" 2>&1 | tail -1 | tee -a $OUT/summary.txt
  RGPU_LIB=$R/$so timeout 600 python scripts/probe_sweep.py mhd_mri_3d ${EXP_N:-512} 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
