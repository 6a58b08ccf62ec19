#!/bin/bash
# experiment: thread-tile shapes of the hydro sweep (needs ramsesgpu_amd/librgpu_exp.so built with -DRG_HYDRO_TILE_EXPERIMENT)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
for T in 32x8 32x16 64x8 64x4 16x16 32x8w3 64x4w3; do
  RGPU_LIB=$R/ramsesgpu_amd/librgpu_exp.so RGPU_HYDRO_TILE=$T python $R/scripts/probe_sweep.py implode3d 256 30 2>&1 | grep -v amdgpu.ids
done
for Z in 16 32 64; do RGPU_ZSEG=$Z RGPU_LIB=$R/ramsesgpu_amd/librgpu_exp.so RGPU_HYDRO_TILE=32x16 python $R/scripts/probe_sweep.py implode3d 256 30 2>&1 | grep -v amdgpu.ids; done
