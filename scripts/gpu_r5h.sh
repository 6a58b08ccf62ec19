cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5h
O=gpurun_out/r5h
for a in contracted; do
  ( RGPU_ARITH=$a python scripts/probe_batch.py 64 20; RGPU_ARITH=$a python scripts/probe_batch.py 64 20; RGPU_ARITH=$a python scripts/probe_batch.py 16 40; RGPU_ARITH=$a PROBE_BASE=implode3d python scripts/probe_batch.py 64 40; RGPU_ARITH=$a python scripts/probe_batch.py 256 10 ) 2>&1 | grep -v amdgpu >> $O/batch.log
done
( AMD_LOG_LEVEL=0 GPU_MAX_HW_QUEUES=8 RGPU_ARITH=contracted python scripts/probe_batch.py 64 20 ) 2>&1 | grep -v amdgpu | sed 's/^/GPU_MAX_HW_QUEUES=8: /' >> $O/batch.log
( HIP_FORCE_DEV_KERNARG=1 RGPU_ARITH=contracted python scripts/probe_batch.py 64 20 ) 2>&1 | grep -v amdgpu | sed 's/^/HIP_FORCE_DEV_KERNARG=1: /' >> $O/batch.log
cat $O/batch.log
