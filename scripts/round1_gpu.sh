set -x
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; tail -5 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; cat gpurun_out/bench_r01.json; tail -3 gpurun_out/bench_r01.err
