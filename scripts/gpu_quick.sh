cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden" 2>&1 | tail -2
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | cut -c1-200 ) > gpurun_out/quick/log.txt 2>&1
cat gpurun_out/quick/log.txt
