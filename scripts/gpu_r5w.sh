# round 5, validation on the final kernel sources (last-column launch + exact-build prologue): whole GPU suite, smoke, profile round (profiles/r05_*), default bench lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/tests.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > $O/smoke.log 2>&1
bash scripts/prof_round.sh r05 > $O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python bench.py --workload implode3d --no-cpu-baseline --no-other-workloads --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_implode3d.json
python bench.py --workload orszag-tang --no-cpu-baseline --no-other-workloads --steps 400 --warmup 10 2>/dev/null | tail -1 > $O/bench_orszag-tang.json
cat $O/tests.log $O/smoke.log; cut -c1-260 $O/bench.json
