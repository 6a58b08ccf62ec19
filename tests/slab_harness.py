"""TEST HARNESS (not part of the product package): the z-slab schedule of csrc/comm/rgpu_comm.cpp written a second time, in
Python over torch.distributed, so that the plane-ranged entry points of include/rgpu.h can be driven with gloo on CPU and
with two ranks on one GPU.  The product's slab driver is the C++ one behind include/rgpu_comm.h.

z-slab decomposition of the 3D step across the GPUs of one node (one process per GPU).

Replaces the reference's MPI cartesian decomposition + host-staged MPI_Sendrecv (HydroRunBaseMpi.cpp:3529-3661,
HydroMpiParameters.cpp:44-80) for the one layout the scope contract needs: mx = my = 1, mz = world_size.
k is the slowest spatial index, so the `ghostWidth` planes to exchange are, per variable, ONE contiguous chunk of
isize*jsize*ghostWidth doubles: no pack / unpack kernels and no host staging -- the chunks are sent straight from
and received straight into the state array with torch.distributed point-to-point calls (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" on CPU for the tests).  The shearing-box remaps are per-(j,k) operations along y at
fixed x borders, hence local to a slab.  The only collective is the MAX all-reduce of 1/dt
(HydroRunBaseMpi.cpp:509-513, 696-700 use MIN on dt; max on 1/dt is the same number).

Order of the ghost fill, identical to the reference's once z-neighbours are other ranks:
   plain    : X, Y, then Z (exchange)                       HydroRunBase.cpp:2333-2342
   shearing : Y, shear remap of x ghosts, Z (exchange), Y   MHDRunGodunov.cpp:3779-3793

Overlap (default): the exchange is hidden behind the update of the planes no neighbour needs.  With all ghosts of
the input valid at entry, a step is

   update planes [0,2gw) and [nz,ksize)       the planes the neighbours (and the physical z faces) read
   in-plane ghost fill of the planes to send   X,Y  |  Y, shear, Y  -- they act within one z plane, so finishing a
                                               plane before it is sent equals the reference's fill-then-copy order
   start isend / irecv of the OUTPUT state     (asynchronous; RCCL runs on its own stream)
   update planes [2gw,nz) + their in-plane fill
   wait, physical z faces

so the next step again starts with valid ghosts.  On the plain path the reference evaluates the CFL condition on
the output BEFORE its ghosts are refilled (oneStepIntegration: compute_dt, then godunov_unsplit fills); the driver
therefore scans 1/dt plane range by plane range right after each update and before each fill (max is order
independent), which keeps the time step -- and hence everything -- bit-identical to the single-domain run.
"""
import ctypes as C

import torch
import torch.distributed as dist

from ramsesgpu_amd._capi import BC_COPY, BC_PERIODIC
from ramsesgpu_amd.solver import Solver, load_library


class SlabRun:
    def __init__(self, ini_path, overrides="", library=None, device="cuda", group=None, overlap=True):
        self.L = library or load_library()
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.ini_path, self.overrides = ini_path, overrides
        self.p = self.L.params_from_ini(ini_path, overrides, slab=(self.rank, self.world))
        if not self.p.three_d:
            raise ValueError("2D problems do not shard: run replicas")
        self.device = torch.device(device)
        # state arrays live in torch tensors so that ghost planes can be sent / received in place
        self.U = [torch.zeros(self.p.shape, dtype=torch.float64, device=self.device) for _ in range(2)]
        stream = 0
        if self.device.type == "cuda":
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.solver = Solver(self.p, self.L, external_state=(self.U[0].data_ptr(), self.U[1].data_ptr()), stream=stream)
        self._invdt = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.nStep, self.totalTime, self.dt = 0, 0.0, 0.0
        self.overlap = overlap
        self._ops = {}
        self._primed = None      # parity of the state whose ghosts are all valid
        self._scanned = None     # parity of the state whose 1/dt sits in the solver's device slot
        gw, nz = self.p.ghostWidth, self.p.nz
        self._lo_ghost, self._lo_int = slice(0, gw), slice(gw, 2 * gw)
        self._hi_int, self._hi_ghost = slice(nz, nz + gw), slice(nz + gw, nz + 2 * gw)

    # ---- initial condition -----------------------------------------------------------------------------------
    def init_simulation(self):
        """each rank builds its own slab of the initial condition (no scatter from rank 0)"""
        import numpy as np
        hU = self.L.init_condition(self.ini_path, self.overrides, self.p)
        self.U[0].copy_(torch.from_numpy(np.ascontiguousarray(hU)))
        G = self.L.init_gravity(self.ini_path, self.overrides, self.p)   # this slab's planes of h_gravity, if the problem has one
        if G is not None:
            self.solver.set_gravity_field(G)
        F = self.L.init_forcing(self.ini_path, self.overrides, self.p)   # ... and of h_randomForcing ("turbulence")
        if F is not None:
            self.solver.set_forcing_field(F)
        self.make_all_boundaries(0, 0.0, 0.0)
        self.U[1].copy_(self.U[0])
        self.nStep, self.totalTime = 0, 0.0
        self._scanned = None

    # ---- halo exchange -----------------------------------------------------------------------------------------
    def _p2p_ops(self, parity):
        """send / recv descriptors of one state array (built once per parity: the tensors never move)"""
        if parity in self._ops:
            return self._ops[parity]
        U = self.U[parity]
        prev, nxt = (self.rank - 1) % self.world, (self.rank + 1) % self.world
        has_prev = self.p.bc[4] == BC_COPY
        has_next = self.p.bc[5] == BC_COPY
        ops = []
        nv = U.shape[0]
        # per variable: one contiguous chunk per face.  Tags keep the two directions apart (gloo); with NCCL the
        # posting order below is mirrored on the peer, which is what its grouped send/recv matching needs.
        for v in range(nv):
            if has_prev:
                ops.append(dist.P2POp(dist.isend, U[v, self._lo_int], prev, self.group, tag=2 * v))
            if has_next:
                ops.append(dist.P2POp(dist.isend, U[v, self._hi_int], nxt, self.group, tag=2 * v + 1))
        for v in range(nv):
            if has_next:
                ops.append(dist.P2POp(dist.irecv, U[v, self._hi_ghost], nxt, self.group, tag=2 * v))
            if has_prev:
                ops.append(dist.P2POp(dist.irecv, U[v, self._lo_ghost], prev, self.group, tag=2 * v + 1))
        self._ops[parity] = ops
        return ops

    def _exchange_start(self, parity):
        """post the exchange of the z ghost planes that belong to a neighbour slab (faces with bc == BC_COPY)"""
        if self.world == 1:
            return []
        ops = self._p2p_ops(parity)
        if ops and self.device.type == "cuda" and dist.get_backend(self.group) != "nccl":
            # RCCL orders its transfers after the kernels already queued on the current stream.  The gloo backend (used
            # by the tests that put two ranks on one GPU) reads the device buffers from host threads as soon as it is
            # called, so the queued kernels have to be finished first.
            torch.cuda.current_stream(self.device).synchronize()
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def _exchange_wait(pending):
        for w in pending:
            w.wait()

    def exchange_z(self, parity):
        self._exchange_wait(self._exchange_start(parity))

    def make_all_boundaries(self, parity, totalTime, dt):
        s = self.solver
        if self.p.shearingBoxEnabled:
            s.make_boundaries(parity, 2)
            s._chk(s.lib.rgpu_make_boundaries_shear(s.ctx, parity, totalTime, dt), "make_boundaries_shear")
            s.make_boundaries(parity, 3)   # physical z faces only; BC_COPY faces are left to the exchange
            self.exchange_z(parity)
            s.make_boundaries(parity, 2)
        else:
            s.make_boundaries(parity, 1)
            s.make_boundaries(parity, 2)
            s.make_boundaries(parity, 3)
            self.exchange_z(parity)
        self._primed = parity

    # ---- time step --------------------------------------------------------------------------------------------
    def _all_reduce_inv_dt(self, inv):
        if self.world > 1:
            self._invdt[0] = inv
            dist.all_reduce(self._invdt, op=dist.ReduceOp.MAX, group=self.group)
            inv = float(self._invdt.item())
        return inv

    def compute_dt(self, useU):
        if self._scanned == useU:      # 1/dt of this state was accumulated plane by plane during the last step
            inv = self.solver.inv_dt_result()
        else:
            self._scanned = None       # the full scan reuses (and resets) the device slot
            inv = self.solver.compute_inv_dt(useU)
        return self.p.cfl / self._all_reduce_inv_dt(inv)

    @property
    def _rotating(self):
        return bool(self.p.mhdEnabled and self.p.Omega0 > 0)

    def godunov_unsplit_serial(self, nStep, dt):
        """exchange between the step pieces, nothing overlapped (the layout of INTEGRATION.md's first recipe)"""
        s, t = self.solver, self.totalTime
        s.step_pre(nStep, dt, t)
        if not self._rotating:
            self.exchange_z(nStep % 2)          # plain path: ghosts of the INPUT
        s.step_core(nStep, dt, t)
        if self._dissipative:
            # viscosity / resistivity work on the updated state with ALL its ghosts: one more fill + exchange
            # (make_all_boundaries(h_UNew) of the reference, mhd_godunov_unsplit_cpu_v3.cpp:662-668)
            self.make_all_boundaries((nStep + 1) % 2, t, dt)
            s.step_dissipative(nStep, dt, t)
        if self.p.randomForcingEnabled:
            self._random_forcing(nStep, dt)
        if self.p.ouForcingEnabled:      # Ornstein-Uhlenbeck forcing: the same process on every rank, no communication
            s._chk(s.lib.rgpu_step_ou_forcing(s.ctx, (nStep + 1) % 2, dt), "step_ou_forcing")
        s.step_post_a(nStep, dt, t)
        if self._rotating:
            self.exchange_z((nStep + 1) % 2)    # rotating path: ghosts of the OUTPUT
        s.step_post_b(nStep, dt, t)
        self._primed = (nStep + 1) % 2 if self._rotating else None
        self._scanned = None

    def _plane_ranges(self):
        """(boundary update ranges, planes to finish and send, inner update range) in array plane indices"""
        gw, nz = self.p.ghostWidth, self.p.nz
        ks = nz + 2 * gw
        if nz <= 2 * gw:
            return [(0, ks)], [(gw, nz + gw)], None
        return [(0, 2 * gw), (nz, ks)], [(gw, 2 * gw), (nz, nz + gw)], (2 * gw, nz)

    @property
    def _dissipative(self):
        return bool(self.p.nu > 0 or (self.p.mhdEnabled and self.p.eta > 0))

    def _random_forcing(self, nStep, dt):
        """random forcing of the "turbulence" problem on the updated state: the two normalisation sums of every slab are
        added (SUM all-reduce), then every slab applies the same factor (compute_random_forcing_normalization +
        add_random_forcing, HydroRunBase.cpp:1201-1428)"""
        import math
        s, p = self.solver, self.p
        pout = (nStep + 1) % 2
        s0, s1 = s.forcing_sums(pout)
        if self.world > 1:
            t = torch.tensor([s0, s1], dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            s0, s1 = (float(v) for v in t.cpu())
        if p.randomForcingEdot == 0:
            norm = 0.0
        else:
            nb = p.nx * p.ny * p.nz_global
            norm = (math.sqrt(s0 * s0 + s1 * dt * p.randomForcingEdot * 2 * nb) - s0) / s1
        s.add_forcing(pout, norm)

    def godunov_unsplit(self, nStep, dt):
        # the dissipative stage needs a second exchange inside the step; the random forcing changes the whole updated
        # state after it (and needs a global sum first): both use the serial schedule
        if not self.overlap or self._dissipative or self.p.randomForcingEnabled or self.p.ouForcingEnabled:
            return self.godunov_unsplit_serial(nStep, dt)
        s, t = self.solver, self.totalTime
        pin, pout = nStep % 2, (nStep + 1) % 2
        rot = self._rotating
        if self._primed != pin and not rot:
            # ghosts of the input not known to be valid (first step): fill them like the reference does at entry
            s.step_pre(nStep, dt, t)
            self.exchange_z(pin)
        boundary, send, inner = self._plane_ranges()
        scan = not rot          # rotating path: the reference's compute_dt sees the refilled ghosts -> full scan later
        for a, b in boundary:
            s.step_core_planes(nStep, dt, t, a, b)
        if scan:
            for n, (a, b) in enumerate(boundary):
                s.inv_dt_accumulate(pout, a, b, reset=(n == 0))
        for a, b in send:
            s.step_fill_planes(nStep, dt, t, a, b)
        pending = self._exchange_start(pout)
        if inner is not None:
            a, b = inner
            s.step_core_planes(nStep, dt, t, a, b)
            if scan:
                s.inv_dt_accumulate(pout, a, b)
            s.step_fill_planes(nStep, dt, t, a, b)
        self._exchange_wait(pending)
        s.make_boundaries(pout, 3)      # physical z faces (+ 3D jet); BC_COPY faces were filled by the exchange
        self._primed = pout
        self._scanned = pout if scan else None

    def oneStepIntegration(self):
        self.dt = self.compute_dt(self.nStep % 2)
        self.godunov_unsplit(self.nStep, self.dt)
        self.nStep += 1
        self.totalTime += self.dt
        return self.dt

    # ---- history diagnostics -----------------------------------------------------------------------------------
    def history_mri(self):
        """MHDRunBase::history_mri over the whole box: per-slab column sums on the device, SUM all-reduce of the
        isize-long columns (the y-z means need the global sums before the Reynolds stress can be formed).
        The magnetic terms read the faces of the first high ghost layer.  The reference's history sees them as the
        update left them on the plain path (ghosts refilled at the START of the next step) and refilled on the rotating
        path; the overlapped schedule refills them at the end of every step, which is the same numbers for periodic /
        shearing / slab-interface faces (the MRI box) but not for open or reflecting physical faces on the plain path:
        use overlap=False when those boundary-face terms matter."""
        import numpy as np
        s, p = self.solver, self.p
        parity = self.nStep % 2
        gw = p.ghostWidth

        def reduce(a):
            if self.world == 1:
                return a
            t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t.cpu().numpy()

        cols = reduce(s.history_columns(parity))
        dTau = p.dx * p.dy * p.dz / (p.xMax - p.xMin) / (p.yMax - p.yMin) / (p.zMax - p.zMin)
        nyz = p.ny * p.nz_global
        rcol = reduce(s.history_reynolds(parity, cols[1] / nyz, cols[2] / nyz, dTau))
        sums = cols[:, gw:-gw].sum(axis=1)
        return {"mass": sums[0] * dTau, "maxwell": sums[4] * dTau, "reynolds": rcol[gw:-gw].sum(), "magp": sums[3] * dTau / 2.0,
                "mean_Bx": sums[5] * dTau, "mean_By": sums[6] * dTau, "mean_Bz": sums[7] * dTau, "divB": sums[8]}

    def local_interior(self):
        gw = self.p.ghostWidth
        return self.U[self.nStep % 2][:, gw:-gw, gw:-gw, gw:-gw]

    def close(self):
        self.solver.close()
