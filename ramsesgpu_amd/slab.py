"""z-slab decomposition of the 3D step across the GPUs of one node (one process per GPU).

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
"""
import ctypes as C

import torch
import torch.distributed as dist

from ._capi import BC_COPY, BC_PERIODIC
from .solver import Solver, load_library


class SlabRun:
    def __init__(self, ini_path, overrides="", library=None, device="cuda", group=None):
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
        gw, nz = self.p.ghostWidth, self.p.nz
        self._lo_ghost, self._lo_int = slice(0, gw), slice(gw, 2 * gw)
        self._hi_int, self._hi_ghost = slice(nz, nz + gw), slice(nz + gw, nz + 2 * gw)

    # ---- initial condition -----------------------------------------------------------------------------------
    def init_simulation(self):
        """each rank builds its own slab of the initial condition (no scatter from rank 0)"""
        import numpy as np
        hU = self.L.init_condition(self.ini_path, self.overrides, self.p)
        self.U[0].copy_(torch.from_numpy(np.ascontiguousarray(hU)))
        self.make_all_boundaries(0, 0.0, 0.0)
        self.U[1].copy_(self.U[0])
        self.nStep, self.totalTime = 0, 0.0

    # ---- halo exchange -----------------------------------------------------------------------------------------
    def exchange_z(self, parity):
        """fill the z ghost planes that belong to a neighbour slab (faces with bc == BC_COPY)"""
        if self.world == 1:
            return
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
        if not ops:
            return
        for w in dist.batch_isend_irecv(ops):
            w.wait()

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

    # ---- time step --------------------------------------------------------------------------------------------
    def compute_dt(self, useU):
        inv = self.solver.compute_inv_dt(useU)
        if self.world > 1:
            self._invdt[0] = inv
            dist.all_reduce(self._invdt, op=dist.ReduceOp.MAX, group=self.group)
            inv = float(self._invdt.item())
        return self.p.cfl / inv

    def godunov_unsplit(self, nStep, dt):
        s, t = self.solver, self.totalTime
        s.step_pre(nStep, dt, t)
        if not (self.p.mhdEnabled and self.p.Omega0 > 0):
            self.exchange_z(nStep % 2)          # plain path: ghosts of the INPUT
        s.step_core(nStep, dt, t)
        s.step_post_a(nStep, dt, t)
        if self.p.mhdEnabled and self.p.Omega0 > 0:
            self.exchange_z((nStep + 1) % 2)    # rotating path: ghosts of the OUTPUT
        s.step_post_b(nStep, dt, t)

    def oneStepIntegration(self):
        self.dt = self.compute_dt(self.nStep % 2)
        self.godunov_unsplit(self.nStep, self.dt)
        self.nStep += 1
        self.totalTime += self.dt
        return self.dt

    def local_interior(self):
        gw = self.p.ghostWidth
        return self.U[self.nStep % 2][:, gw:-gw, gw:-gw, gw:-gw]

    def close(self):
        self.solver.close()
