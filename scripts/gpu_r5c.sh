# round 5, GPU call 3: device-side time step for 3D / slabs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5c
( time timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r5c/tests.log 2>&1
( RGPU_ARITH=contracted PROBE_LINK_GBPS="0 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r5c/probe64.log 2>&1
for s in 1 2; do ( RGPU_ARITH=contracted RGPU_COMM_SCHEDULE=$s PROBE_LINK_GBPS="0 60 40" PROBE_NZ=64 python scripts/slab_probe.py 2>&1 | grep "^nz" | sed "s/default/sched $s/" ) >> gpurun_out/r5c/probe64.log 2>&1; done
( RGPU_ARITH=contracted bash scripts/slab_timeline.sh 1 60 2>&1 | tail -60 ) > gpurun_out/r5c/timeline1.txt 2>&1
( python bench.py --workload implode3d --steps 100 --warmup 5 --no-cpu-baseline 2>gpurun_out/r5c/bench_implode.err | tail -1 ) > gpurun_out/r5c/bench_implode.json
( python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-workloads 2>gpurun_out/r5c/bench.err | tail -1 ) > gpurun_out/r5c/bench.json
cat gpurun_out/r5c/tests.log gpurun_out/r5c/probe64.log
python - <<'PY'
import json
for f in ("bench_implode", "bench"):
    try:
        d = json.loads(open("gpurun_out/r5c/%s.json" % f).read())
        print(f, d["value"], d["ms_per_step"], d.get("value_exact", {}).get("value"), d.get("value_exact", {}).get("ms_per_step"), d["config"]["fingerprint"]["state_sum_u64"])
    except Exception as e:
        print(f, "failed", e, open("gpurun_out/r5c/%s.err" % f.replace("bench_implode", "bench_implode")).read()[-1500:])
PY
