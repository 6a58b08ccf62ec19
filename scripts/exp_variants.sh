#!/bin/bash
# A/B runs of experiment builds of the library (ramsesgpu_amd/librgpu_exp_<tag>.so, built in the container with
# scripts/build_exp.py; tags starting with f_ are contracted-arithmetic builds): each is checked against the oracle at the
# bench's launch geometry (bit for bit / relative L2 < 1e-12) and timed on the 512^3 MRI box (10 steps + phase timers).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
OUT=$R/gpurun_out/${EXP_OUT:-exp_r4}; mkdir -p $OUT
cd $R
for so in ramsesgpu_amd/librgpu.so ramsesgpu_amd/librgpu_fast.so $(ls ramsesgpu_amd/librgpu_exp_*.so 2>/dev/null); do
  tag=$(basename $so .so)
  echo "=== $tag" | tee -a $OUT/summary.txt
  case $tag in
    librgpu_fast|librgpu_exp_f_*) T="tests/test_contracted.py -k test_bench_launch_geometry_within_tolerance" ;;
    *) T="tests/test_gpu_parity.py -k test_bench_launch_geometry_vs_oracle" ;;
  esac
  if [ -z "$EXP_NO_PARITY" ]; then
    RGPU_LIB=$R/$so timeout 900 python -m pytest $T -x -q -k "mhd_mri_3d and 512" 2>&1 | tail -1 | tee -a $OUT/summary.txt
  fi
  RGPU_LIB=$R/$so timeout 600 python scripts/probe_sweep.py mhd_mri_3d ${EXP_N:-512} 10 2>&1 | grep -v amdgpu.ids | tee -a $OUT/summary.txt
done
