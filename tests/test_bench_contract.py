"""bench.py's contract on the GPU box at toy sizes: one JSON line with the driver's keys plus `roofline` and (N=1)
`cpu_baseline`, for every workload; the N>1 launch line of the driver (torch.distributed.run, one rank per GPU) exercised
with two ranks sharing the one GPU of the box through the test hook (gloo + the Python slab harness: RCCL refuses two ranks
on one device; the C++ RCCL driver itself is covered by tests/test_comm_driver.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
        "data", "config", "roofline"}


def last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("args", [["--workload", "mri", "--nx", "32", "--ny", "48", "--nz", "32"],
                                  ["--workload", "implode3d", "--size", "48"],
                                  ["--workload", "orszag-tang", "--size", "64"]], ids=["mri", "implode3d", "orszag-tang"])
def test_bench_line_single_gpu(args, gpu_lib):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + args,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = last_json(res.stdout)
    assert KEYS <= set(d) and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["value"] > 0 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # the headline is the exact library; the contracted-arithmetic variant is reported beside it, never instead of it
    assert d["config"]["arithmetic"] == "exact" and "bit-identical" in d["config"]["parity"]
    c = d["contracted_arithmetic"]
    assert c["value"] > 0 and "librgpu_fast.so" in c["library"] and "not bit-identical" in c["parity"]


def test_bench_line_contracted_on_request(gpu_lib):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--arith", "contracted",
                          "--workload", "mri", "--nx", "32", "--ny", "48", "--nz", "32"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    d = last_json(res.stdout)
    assert d["config"]["arithmetic"] == "contracted" and "not bit-identical" in d["config"]["parity"] and "contracted_arithmetic" not in d


def test_bench_two_ranks_through_the_launch_line(gpu_lib):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--nx", "32", "--ny", "48", "--nz", "32"]
    env = dict(os.environ, RGPU_BENCH_ONE_DEVICE="1", RGPU_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = last_json(res.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and "cpu_baseline" not in d
