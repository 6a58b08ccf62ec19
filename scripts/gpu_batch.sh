cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
( time timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_steps or fused_hydro2d or orszag_tang_gate" 2>&1 | tail -15 ) > gpurun_out/r4g/tests.log 2>&1
( timeout 900 python scripts/probe_2d.py 2>&1 | grep -v amdgpu.ids; RGPU_ARITH=contracted timeout 900 python scripts/probe_2d.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4g/probe2d.log 2>&1
cat gpurun_out/r4g/tests.log gpurun_out/r4g/probe2d.log
