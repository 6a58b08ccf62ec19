# bench.py itself, A/B of the last-column launch on one box (contracted + exact), then the default line with everything
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5u; mkdir -p $O; rm -f $O/ab.log
for rep in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export RGPU_NO_LASTX_TILES=1; else unset RGPU_NO_LASTX_TILES; fi
    python bench.py --no-cpu-baseline --no-other-workloads 2>/dev/null | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read()); e=b['value_exact']
print('$v rep $rep: contracted %.1f Mcell/s %.3f ms/step sweep %.3f | exact %.1f %.3f sweep %.3f' % (b['value'], b['ms_per_step'], b['roofline']['avg_launch_ms'], e['value'], e['ms_per_step'], e['roofline']['avg_launch_ms']))" >> $O/ab.log
  done
done
unset RGPU_NO_LASTX_TILES
python bench.py 2>$O/bench.err | tail -1 > $O/bench.json
python bench.py --workload implode3d --no-cpu-baseline --no-other-workloads --steps 100 --warmup 10 2>/dev/null | tail -1 > $O/bench_implode3d.json
python bench.py --workload orszag-tang --no-cpu-baseline --no-other-workloads --steps 400 --warmup 10 2>/dev/null | tail -1 > $O/bench_orszag-tang.json
cat $O/ab.log; cut -c1-250 $O/bench.json
