"""N>1 path on CPU: world_size 2 (and 3) z-slab runs over gloo, halo planes exchanged in place by
ramsesgpu_amd.slab.SlabRun, stepping through the TEST-ONLY emulation library; result == single-domain oracle,
bit for bit (plain path: ghosts of the input; rotating path: ghosts of the output; Dirichlet ends; dt all-reduce)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

CASES = [
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=12", 3, 2),                       # plain MHD, periodic ring
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MHD.omega0=0.02", 4, 2),          # rotating + shearing box
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=8;hydro.riemannSolver=hllc", 4, 2),     # hydro, Dirichlet ends
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=12", 3, 3),                            # three slabs
]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("base,ov,nsteps,world", CASES, ids=["%s-x%d" % (c[0], c[3]) for c in CASES])
def test_slabs_match_single_domain(base, ov, nsteps, world, emu_lib, oracle, tmp_path):
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "slab_worker.py"), base, ov, str(nsteps), out]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().strip() == "OK", open(out).read()
