# round 5, first GPU call: the new N>1-on-one-GPU tests of the C++ slab driver + the bench fingerprint
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5a
( time timeout 2400 python -m pytest tests/test_comm_device.py -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r5a/tests.log 2>&1
( time python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads 2>gpurun_out/r5a/bench.err | tail -1 ) > gpurun_out/r5a/bench.log 2>&1
cat gpurun_out/r5a/tests.log; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r5a/bench.log").read().split("\n")[0])
    print(d["value"], d["ms_per_step"], d["config"]["fingerprint"], d.get("value_exact", {}).get("fingerprint"))
except Exception as e:
    print("bench:", e); print(open("gpurun_out/r5a/bench.log").read()[-2000:]); print(open("gpurun_out/r5a/bench.err").read()[-2000:])
PY
