cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5o
( time python bench.py 2>gpurun_out/r5o/bench.err | tail -1 ) > gpurun_out/r5o/bench.json 2> gpurun_out/r5o/time.txt
cat gpurun_out/r5o/time.txt; cut -c1-400 gpurun_out/r5o/bench.json
