"""ctypes binding of the C++ z-slab driver (include/rgpu_comm.h, librgpu_comm.so): one process per GPU, RCCL halo
exchange on a side stream, 1/dt all-reduced in the context's device slot.  CommRun mirrors the method names of the
reference's Mpi run classes (init_simulation, make_all_boundaries, compute_dt, godunov_unsplit, oneStepIntegration) (tests/slab_harness.py
is the same schedule over torch.distributed, a test harness); here Python only launches -- the schedule, the
exchange and the reduction are C++."""
import ctypes as C
import os

import numpy as np

from . import _capi
from .solver import RgpuError, Solver, load_library

ID_BYTES = 128


def comm_lib_path(arithmetic="exact"):
    """librgpu_comm.so drives librgpu.so, librgpu_comm_fast.so the contracted-arithmetic librgpu_fast.so (one variant per
    process: both export the same symbols)"""
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "librgpu_comm.so" if arithmetic == "exact" else "librgpu_comm_fast.so")


def load_comm_library(path=None):
    # RTLD_LOCAL: DT_NEEDED + rpath ($ORIGIN) bind the driver to ITS librgpu*.so; a global load would promote every rgpu_*
    # symbol and let another library of the same ABI in the process (the contracted variant, a test build) interpose them
    lib = C.CDLL(path or comm_lib_path())
    lib._rgpu_path = os.path.abspath(path or comm_lib_path())
    cm = C.c_void_p
    lib.rgpu_comm_unique_id.restype = C.c_int
    lib.rgpu_comm_unique_id.argtypes = [C.c_char_p]
    lib.rgpu_comm_create.restype = C.c_int
    lib.rgpu_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(cm)]
    lib.rgpu_comm_destroy.restype = None
    lib.rgpu_comm_destroy.argtypes = [cm]
    lib.rgpu_comm_last_error.restype = C.c_char_p
    lib.rgpu_comm_last_error.argtypes = [cm]
    lib.rgpu_comm_transport_name.restype = C.c_char_p
    lib.rgpu_comm_set_device.restype = C.c_int
    lib.rgpu_comm_set_device.argtypes = [C.c_int]
    lib.rgpu_comm_rccl_version.restype = C.c_int
    lib.rgpu_comm_rccl_version.argtypes = [cm]
    lib.rgpu_comm_info.restype = C.c_int
    lib.rgpu_comm_info.argtypes = [cm, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    lib.rgpu_comm_last_exchange_ms.restype = C.c_double
    lib.rgpu_comm_last_exchange_ms.argtypes = [cm]
    lib.rgpu_comm_halo_bytes.restype = C.c_longlong
    lib.rgpu_comm_halo_bytes.argtypes = [cm]
    lib.rgpu_comm_set_overlap.restype = C.c_int
    lib.rgpu_comm_set_overlap.argtypes = [cm, C.c_int]
    lib.rgpu_comm_schedule.restype = C.c_int
    lib.rgpu_comm_schedule.argtypes = [cm]
    for name in ("rgpu_comm_exchange_z_wait",):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [cm]
    lib.rgpu_comm_exchange_z_start.restype = C.c_int
    lib.rgpu_comm_exchange_z_start.argtypes = [cm, C.c_int]
    lib.rgpu_comm_make_all_boundaries.restype = C.c_int
    lib.rgpu_comm_make_all_boundaries.argtypes = [cm, C.c_int, C.c_double, C.c_double]
    lib.rgpu_comm_compute_dt.restype = C.c_int
    lib.rgpu_comm_compute_dt.argtypes = [cm, C.c_int, C.POINTER(C.c_double)]
    lib.rgpu_comm_godunov_unsplit.restype = C.c_int
    lib.rgpu_comm_godunov_unsplit.argtypes = [cm, C.c_int, C.c_double, C.c_double]
    lib.rgpu_comm_history_mri.restype = C.c_int
    lib.rgpu_comm_history_mri.argtypes = [cm, C.c_int, C.POINTER(C.c_double)]
    lib.rgpu_comm_history_turbulence.restype = C.c_int
    lib.rgpu_comm_history_turbulence.argtypes = [cm, C.c_int, C.POINTER(C.c_double)]
    lib.rgpu_comm_one_step_integration.restype = C.c_int
    lib.rgpu_comm_one_step_integration.argtypes = [cm, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rgpu_comm_clocked_steps.restype = C.c_longlong
    lib.rgpu_comm_clocked_steps.argtypes = [cm]
    lib.rgpu_comm_run_steps.restype = C.c_int
    lib.rgpu_comm_run_steps.argtypes = [cm, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.rgpuh_run_slabs.restype = C.c_int
    lib.rgpuh_run_slabs.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_double), C.c_char_p, C.c_int]
    return lib


DECLARED_SYMBOLS = [
    "rgpu_comm_unique_id", "rgpu_comm_create", "rgpu_comm_destroy", "rgpu_comm_last_error", "rgpu_comm_exchange_z_start",
    "rgpu_comm_exchange_z_wait", "rgpu_comm_make_all_boundaries", "rgpu_comm_compute_dt", "rgpu_comm_godunov_unsplit",
    "rgpu_comm_one_step_integration", "rgpu_comm_run_steps", "rgpu_comm_clocked_steps", "rgpu_comm_history_mri", "rgpu_comm_history_turbulence", "rgpu_comm_set_overlap", "rgpu_comm_schedule", "rgpu_comm_halo_bytes", "rgpu_comm_last_exchange_ms", "rgpu_comm_set_device", "rgpu_comm_info", "rgpu_comm_rccl_version", "rgpu_comm_transport_name", "rgpuh_run_slabs",
]


def unique_id(comm_lib):
    buf = C.create_string_buffer(ID_BYTES)
    if comm_lib.rgpu_comm_unique_id(buf) != 0:
        raise RgpuError("rgpu_comm_unique_id failed")
    return buf.raw


class CommRun:
    """rank `rank` of `world` z-slabs.  comm_id: the 128 bytes of rgpu_comm_unique_id from rank 0 (any side channel)."""

    def __init__(self, ini_path, overrides, rank, world, comm_id, library=None, comm_library=None, overlap=None, self_ring=False):
        """self_ring (world = 1, periodic z only; measurement / test): the slab is its own z neighbour -- its periodic z faces
        become slab interfaces ([run] slabSelfRing) and the halo planes really go through the transport, to itself"""
        self.L = library or load_library()
        # the driver library must be the one linked against self.L (librgpu_comm.so <-> librgpu.so, _fast <-> _fast):
        # a context created by one library must not be stepped by the other
        self.CL = comm_library or load_comm_library(comm_lib_path(self.L.arithmetic))
        here = os.path.dirname(os.path.abspath(__file__))
        cpath = getattr(self.CL, "_rgpu_path", "")
        if os.path.dirname(cpath) == here and os.path.dirname(os.path.abspath(self.L.path)) == here:
            # (librgpu_comm_measure.so: the measurement build of scripts/slab_probe.py, linked against librgpu_fast.so by build.py --measure)
            ok_names = [os.path.basename(comm_lib_path(self.L.arithmetic))] + (["librgpu_comm_measure.so"] if self.L.arithmetic == "contracted" else [])
            if os.path.basename(cpath) not in ok_names:
                raise RgpuError("CommRun: %s does not drive %s (arithmetic %s)" % (os.path.basename(cpath), os.path.basename(self.L.path), self.L.arithmetic))
        self.rank, self.world = rank, world
        self.ini_path, self.overrides = ini_path, overrides
        if self_ring:
            if world != 1:
                raise ValueError("self_ring is a ring of ONE rank")
            overrides = (overrides + ";" if overrides else "") + "run.slabSelfRing=yes"
            self.overrides = overrides
        self.p = self.L.params_from_ini(ini_path, overrides, slab=(rank, world))
        if not self.p.three_d:
            raise ValueError("2D problems do not shard: run replicas")
        self.solver = Solver(self.p, self.L)
        self.cm = C.c_void_p()
        rc = self.CL.rgpu_comm_create(self.solver.ctx, rank, world, comm_id, C.byref(self.cm))
        if rc != 0:
            msg = self.CL.rgpu_comm_last_error(self.cm).decode() if self.cm else "allocation"
            raise RgpuError("rgpu_comm_create: %s (%d)" % (msg, rc))
        # overlap: None = the driver's choice (-1), False / True = serial / overlapped (0 / 1), 2 = boundary-first (include/rgpu_comm.h)
        mode = -1 if overlap is None else (int(overlap) if not isinstance(overlap, bool) else (1 if overlap else 0))
        self._chk(self.CL.rgpu_comm_set_overlap(self.cm, mode), "set_overlap")
        self.nStep, self.totalTime, self.dt = 0, 0.0, 0.0

    def _chk(self, rc, what):
        if rc != 0:
            raise RgpuError("%s: %s (%d)" % (what, self.CL.rgpu_comm_last_error(self.cm).decode(), rc))

    def schedule(self):
        """the step schedule in force (0 serial, 1 overlapped, 2 boundary-first): rgpu_comm_schedule"""
        return int(self.CL.rgpu_comm_schedule(self.cm))

    def halo_bytes(self):
        """bytes this rank sends per halo exchange (0: nothing is exchanged)"""
        return int(self.CL.rgpu_comm_halo_bytes(self.cm))

    def last_exchange_ms(self):
        """duration of the last halo exchange on the halo stream [ms] (< 0: none)"""
        return float(self.CL.rgpu_comm_last_exchange_ms(self.cm))

    def info(self):
        """what the transport (RCCL) reports: {"ranks", "rank", "device", "pci_bus_id", "transport"}"""
        n, r, d = C.c_int(0), C.c_int(-1), C.c_int(-1)
        pci = C.create_string_buffer(64)
        self._chk(self.CL.rgpu_comm_info(self.cm, C.byref(n), C.byref(r), C.byref(d), pci, 64), "comm_info")
        return {"ranks": n.value, "rank": r.value, "device": d.value, "pci_bus_id": pci.value.decode(),
                "transport": self.CL.rgpu_comm_transport_name().decode(), "rccl_version": int(self.CL.rgpu_comm_rccl_version(self.cm))}

    def init_simulation(self):
        """each rank builds its own slab of the initial condition (no scatter from rank 0)"""
        hU = self.L.init_condition(self.ini_path, self.overrides, self.p)
        self.solver.upload(np.ascontiguousarray(hU), both=False)
        G = self.L.init_gravity(self.ini_path, self.overrides, self.p)
        if G is not None:
            self.solver.set_gravity_field(G)
        F = self.L.init_forcing(self.ini_path, self.overrides, self.p)
        if F is not None:
            self.solver.set_forcing_field(F)
        self.make_all_boundaries(0, 0.0, 0.0)
        self.nStep, self.totalTime = 0, 0.0

    def make_all_boundaries(self, parity, totalTime, dt):
        self._chk(self.CL.rgpu_comm_make_all_boundaries(self.cm, parity, totalTime, dt), "make_all_boundaries")

    def compute_dt(self, useU):
        dt = C.c_double(0)
        self._chk(self.CL.rgpu_comm_compute_dt(self.cm, useU, C.byref(dt)), "compute_dt")
        return dt.value

    def godunov_unsplit(self, nStep, dt):
        self._chk(self.CL.rgpu_comm_godunov_unsplit(self.cm, nStep, dt, self.totalTime), "godunov_unsplit")

    def oneStepIntegration(self):
        n, t, dt = C.c_int(self.nStep), C.c_double(self.totalTime), C.c_double(0)
        self._chk(self.CL.rgpu_comm_one_step_integration(self.cm, C.byref(n), C.byref(t), C.byref(dt)), "oneStepIntegration")
        self.nStep, self.totalTime, self.dt = n.value, t.value, dt.value
        return self.dt

    def run_steps(self, nsteps, tEnd=float("inf")):
        """up to nsteps turns of the reference's time loop (rgpu_comm_run_steps: the time step stays on the device between steps, the
        host reads the records of a batch once); returns the steps done, self.dt_log = their time steps"""
        n, t, d = C.c_int(self.nStep), C.c_double(self.totalTime), C.c_double(self.dt)
        log = (C.c_double * max(int(nsteps), 1))()
        done = self.CL.rgpu_comm_run_steps(self.cm, int(nsteps), float(tEnd), C.byref(n), C.byref(t), C.byref(d), log)
        if done < 0:
            self._chk(done, "run_steps")
        self.nStep, self.totalTime, self.dt = n.value, t.value, d.value
        self.dt_log = [log[i] for i in range(done)]
        return done

    def enable_timers(self, on=True):
        self.solver.enable_timers(on)

    def clocked_steps(self):
        """steps whose time step came from the device record"""
        return int(self.CL.rgpu_comm_clocked_steps(self.cm))

    def local_interior(self):
        """interior cells of this slab (host copy)"""
        gw = self.p.ghostWidth
        self.solver.synchronize()
        return self.solver.getDataHost(self.nStep % 2)[:, gw:-gw, gw:-gw, gw:-gw]

    def close(self):
        if self.cm:
            self.CL.rgpu_comm_destroy(self.cm)
            self.cm = C.c_void_p()
        self.solver.close()
