"""N>1 path on CPU: world_size 2 (and 3) z-slab runs over gloo, halo planes exchanged in place by
tests/slab_harness.py SlabRun, stepping through the TEST-ONLY emulation library; result == single-domain oracle,
bit for bit (plain path: ghosts of the input; rotating path: ghosts of the output; Dirichlet ends; dt all-reduce).
Both schedules of SlabRun are covered: the overlapped one (boundary planes first, exchange in flight during the
inner planes, 1/dt scanned plane range by plane range) and the serial one (exchange between the step pieces)."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

OPEN_BC = ";mesh.boundary_xmin=2;mesh.boundary_xmax=2;mesh.boundary_ymin=1;mesh.boundary_ymax=2;mesh.boundary_zmin=2;mesh.boundary_zmax=1"
# (ini, overrides, steps, world, overlap)
CASES = [
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=12", 3, 2, 1),                      # plain MHD, periodic ring, no inner planes
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 4, 2, 1),                      # ... with inner planes [6,12)
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=24", 3, 2, 0),                      # serial schedule
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=20" + OPEN_BC, 4, 2, 1),            # plain MHD, open / reflecting faces: dt sees unfilled ghosts
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MHD.omega0=0.02", 4, 2, 1),        # rotating + shearing box, no inner planes
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=20;MHD.omega0=0.02", 4, 2, 1),        # ... inner planes [6,10)
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MHD.omega0=0.02", 3, 2, 0),        # serial schedule
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16;MHD.omega0=0.05", 3, 2, 1),      # rotating frame without shearing box
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=8;hydro.riemannSolver=hllc", 4, 2, 1),  # hydro, Dirichlet ends
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16", 4, 2, 1),                          # hydro, inner planes [4,8)
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=12", 3, 3, 1),                         # three slabs (nz=4 each < 2 gw)
    ("mhd_mri_3d", "mesh.nx=6;mesh.ny=8;mesh.nz=27", 3, 3, 1),                         # three slabs, inner planes [6,9)
    ("kelvin_helmholtz_gpu_3d", "mesh.nx=8;mesh.ny=4;mesh.nz=16", 3, 2, 1),           # libc rand() stream continued across slabs
    ("mhd_fieldloop3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16", 3, 2, 1),                    # drand48 noise of the vector potential
    ("rayleigh_taylor_gpu_3d_mhd", "mesh.nx=6;mesh.ny=6;mesh.nz=16;rayleigh-taylor.randomEnabled=yes", 3, 2, 1),   # gravity + rand() over ghosts too
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16;implode.amplitude=0.02;hydro.unsplitVersion=2", 3, 2, 1),        # direction-wise update order
    ("orszag-tang3d", "mesh.nx=8;mesh.ny=8;mesh.nz=16;hydro.nu=0.005;MHD.eta=0.01", 3, 2, 1),                      # dissipative stage: second exchange
    ("mhd_mri_3d", "mesh.nx=8;mesh.ny=12;mesh.nz=12;MRI.amp=0.2;hydro.nu=1e-6;MHD.eta=2e-6", 3, 2, 1),             # ... on the rotating path
    ("implode3d", "mesh.nx=8;mesh.ny=8;mesh.nz=12;hydro.nu=0.002", 3, 3, 1),                                       # hydro viscosity, three slabs
    ("orszag-tang3d", "mesh.nx=6;mesh.ny=6;mesh.nz=18;hydro.nu=0.005;MHD.eta=0.01", 3, 3, 1),                      # middle slab: internal interfaces on both sides
    ("mhd_BrioWu", "mesh.nx=8;mesh.ny=6;mesh.nz=16;BrioWu.direction=2;MHD.implementationVersion=4;MHD.eta=0.02;mesh.boundary_zmin=2;mesh.boundary_zmax=2", 3, 2, 1),  # open z ends
    ("Keplerian_disk2d", "mesh.nx=10;mesh.ny=10;mesh.nz=12;hydro.riemannSolver=hll", 3, 2, 1),    # per-cell gravity field, slab by slab
    ("mhd_mri_3d_stratified", "mesh.nx=6;mesh.ny=8;mesh.nz=24;hydro.slope_type=2.0;MRI.amp=0.3", 3, 2, 1),       # stratified box: g_z(z) planes per slab, z-stratified end faces
    ("mhd_mri_3d_stratified", "mesh.nx=6;mesh.ny=8;mesh.nz=24;hydro.slope_type=2.0;MRI.amp=0.3", 3, 3, 0),       # ... three slabs, serial schedule
    ("turbulence_hydro", "mesh.nx=8;mesh.ny=8;mesh.nz=12", 3, 2, 1),                                              # random forcing: all-reduced normalisation sums (round-off agreement)
    ("turbulence_mhd", "mesh.nx=6;mesh.ny=6;mesh.nz=18;hydro.slope_type=2.0", 3, 3, 1),
    ("turbulence_hydro_ou", "mesh.nx=8;mesh.ny=8;mesh.nz=12;turbulence-Ornstein-Uhlenbeck.initialDensityPerturbationAmplitude=0.1", 3, 2, 1),   # Ornstein-Uhlenbeck forcing: same process on every rank
    ("turbulence_mhd_ou", "mesh.nx=6;mesh.ny=6;mesh.nz=18", 3, 3, 1),
]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("base,ov,nsteps,world,overlap", CASES,
                         ids=["%s-%d-x%d-%s" % (c[0], n, c[3], "overlap" if c[4] else "serial") for n, c in enumerate(CASES)])
def test_slabs_match_single_domain(base, ov, nsteps, world, overlap, emu_lib, oracle, tmp_path):
    out = str(tmp_path / "result.txt")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "slab_worker.py"), base, ov, str(nsteps), out]
    env = dict(os.environ, OMP_NUM_THREADS="1", SLAB_OVERLAP=str(overlap))
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=300)
    assert res.returncode == 0, res.stdout[-3000:]
    assert open(out).read().strip() == "OK", open(out).read()
