# the bench lines of the final state (pmc_traffic.json matches the kernel sources: roofline.traffic / valu_ceiling populated)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; mkdir -p $O
python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python bench.py --workload implode3d --no-cpu-baseline --no-other-workloads --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_implode3d.json
python bench.py --workload orszag-tang --no-cpu-baseline --no-other-workloads --steps 400 --warmup 10 2>/dev/null | tail -1 > $O/bench_orszag-tang.json
cut -c1-260 $O/bench.json
