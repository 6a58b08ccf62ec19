cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5q
( time python bench.py 2>gpurun_out/r5q/bench.err | tail -1 ) > gpurun_out/r5q/bench.json 2> gpurun_out/r5q/time.txt
for w in implode3d orszag-tang; do
  python bench.py --workload $w --no-cpu-baseline --no-other-workloads 2>gpurun_out/r5q/bench_$w.err | tail -1 > gpurun_out/r5q/bench_$w.json
done
cat gpurun_out/r5q/time.txt; cut -c1-300 gpurun_out/r5q/bench*.json
