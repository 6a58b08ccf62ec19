cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
( time timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r4n/tests.log 2>&1
( python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4n/smoke.log 2>&1
( python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', d['value'], 'ms', d['ms_per_step'], 'exact', d['value_exact']['value'], 'frac', r['frac'], 'traffic', r.get('traffic'), 'cpu', d['cpu_baseline']['value'], d.get('cpu_baseline_all_cores',{}).get('value'))
print({k:(v['value'],v['ms_per_step']) for k,v in d['other_workloads'].items()})
" ) > gpurun_out/r4n/bench.log 2>&1
cat gpurun_out/r4n/tests.log gpurun_out/r4n/smoke.log gpurun_out/r4n/bench.log
