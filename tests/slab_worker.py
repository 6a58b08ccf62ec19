"""Worker of tests/test_slab_gloo.py: one rank of a world_size-N z-slab run on CPU (gloo + TEST-ONLY emulation
library).  Each rank steps its slab with halo exchange through tests/slab_harness.py SlabRun, then the slabs are
gathered on rank 0 and compared, bit for bit, with the single-domain oracle run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle_api import Oracle  # noqa: E402
from slab_harness import SlabRun  # noqa: E402
from ramsesgpu_amd.solver import Library, interior  # noqa: E402


def main():
    base, ov, nsteps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # default: the TEST-ONLY emulation library on CPU; SLAB_DEVICE=cuda:0 runs the product library (HIP) instead, all
    # ranks sharing that GPU (gloo moves the ghost planes: RCCL refuses two ranks on one device)
    device = os.environ.get("SLAB_DEVICE", "cpu")
    if device == "cpu":
        lib = Library(os.path.join(ROOT, "tests", "_build", "librgpu_emu.so"))
    else:
        from ramsesgpu_amd.solver import load_library
        lib = load_library()
    ini = os.path.join(ROOT, "configs", base + ".ini")
    run = SlabRun(ini, ov, library=lib, device=device, overlap=os.environ.get("SLAB_OVERLAP", "1") != "0")
    run.init_simulation()
    dts = [run.oneStepIntegration() for _ in range(nsteps)]
    hist = run.history_mri() if run.p.mhdEnabled else None
    local = run.local_interior().contiguous().cpu()
    parts = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
    dist.gather(local, parts, dst=0)
    ok = True
    if rank == 0:
        got = torch.cat(parts, dim=1).numpy()
        oracle = Oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
        p = lib.params_from_ini(ini, ov)
        U0 = lib.init_condition(ini, ov, p)
        oracle.set_gravity_field(lib.init_gravity(ini, ov, p))
        oracle.set_forcing_field(lib.init_forcing(ini, ov, p))
        ref_full, dts_ref, _ = oracle.run(p, U0, nsteps)
        ref = interior(ref_full, p)
        nbad = int((got != ref).sum())
        ok = nbad == 0 and np.array_equal(np.array(dts), dts_ref)
        if p.randomForcingEnabled or (p.ouForcingEnabled and device != "cpu"):   # (OU on a GPU: the device's cos())
            # the forcing normalisation is a global sum: the reference adds sequentially, the slabs in a fixed parallel
            # order + all-reduce -> agreement to round-off (stated tolerance: relative L2 < 1e-12), not bit for bit
            rel = float(np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum()))
            ok = rel < 1e-12 and np.allclose(np.array(dts), dts_ref, rtol=1e-12, atol=0)
            nbad = 0 if ok else nbad
        # slab-wise history sums (all-reduced) against the oracle's single-domain loops.  Exact comparison needs the ghost
        # faces in the state the reference's history sees: true on the rotating path, with the serial schedule, and for
        # periodic / shearing faces (see SlabRun.history_mri)
        pg = lib.params_from_ini(ini, ov)
        exact = run._rotating or not run.overlap or all(b in (3, 4) for b in pg.bc)
        if hist is not None and exact:
            href = oracle.history_mri(p, ref_full)
            scale = max(abs(v) for v in href) + 1e-30
            for k, v in zip(("mass", "maxwell", "reynolds", "magp", "mean_Bx", "mean_By", "mean_Bz", "divB"), href):
                tol = 1e-11 * max(abs(v), scale if k != "divB" else scale / min(p.dx, p.dy, p.dz))
                if not abs(hist[k] - v) <= tol:
                    ok = False
                    sys.stderr.write("history mismatch %s: slabs %r oracle %r tol %g\n" % (k, hist[k], v, tol))
        with open(out, "w") as f:
            f.write("OK\n" if ok else "MISMATCH %d doubles, dt equal=%s\n" % (nbad, np.array_equal(np.array(dts), dts_ref)))
    dist.barrier()
    run.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
