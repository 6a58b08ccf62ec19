"""Python mirror of the reference's run-class interface on top of the C ABI (include/rgpu.h).

`Solver` keeps the reference's method names and argument meaning for the path in scope
(HydroRunBase / MHDRunBase / *RunGodunov): make_all_boundaries, compute_dt(useU), godunov_unsplit(nStep, dt),
oneStepIntegration, copyGpuToCpu / getDataHost.  No numerics live here: every call goes to librgpu.so, which
fails loudly (RGPU_ENODEVICE) when there is no GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import RgpuParams

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path(arithmetic=None):
    """in-tree location of the product library (built by __graft_entry__.build / ramsesgpu_amd/build.py).
    arithmetic: "exact" (librgpu.so: bit-identical to the reference) or "contracted" (librgpu_fast.so: FMA contraction,
    ~1-ulp division / square root; agrees with the reference to round-off); default from $RGPU_ARITH, else exact."""
    if os.environ.get("RGPU_LIB"):   # an experiment build (scripts/)
        return os.environ["RGPU_LIB"]
    arithmetic = arithmetic or os.environ.get("RGPU_ARITH", "exact")
    if arithmetic not in ("exact", "contracted"):
        raise ValueError("arithmetic must be 'exact' or 'contracted', not %r" % (arithmetic,))
    return os.path.join(_HERE, "librgpu.so" if arithmetic == "exact" else "librgpu_fast.so")


class RgpuError(RuntimeError):
    pass


def _init_torch_hip_first():
    """The PyTorch-ROCm wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's).  A process must end up with
    ONE HIP runtime, or torch sees "No HIP GPUs" and torch-owned device pointers (slab driver) are foreign to our
    kernels.  Whoever loads first wins, so let torch initialise HIP before librgpu.so is dlopen'ed; librgpu.so's
    DT_NEEDED libamdhip64.so.7 then resolves to the already loaded runtime.  No-op without torch / without a GPU."""
    if os.environ.get("RGPU_NO_TORCH_PRELOAD"):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


class Library:
    """A loaded C-ABI library (product: librgpu.so)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RgpuError("%s not found: build it first (python -c 'import __graft_entry__ as g; g.build()')" % path)
        self.path = path
        _init_torch_hip_first()
        self.lib = C.CDLL(path)
        _capi.declare_host_api(self.lib)
        _capi.declare_device_api(self.lib)

    @property
    def backend(self):
        return self.lib.rgpu_backend_name().decode()

    @property
    def arithmetic(self):
        """"exact" or "contracted" (see lib_path)"""
        return self.lib.rgpu_arithmetic().decode()

    def set_option(self, name, value):
        """diagnostic option of the library (include/rgpu.h, "Environment and options"); returns the previous value"""
        old = self.lib.rgpu_set_option(name.encode(), int(value))
        if old < 0 and self.lib.rgpu_get_option(name.encode()) != int(value):
            raise RgpuError("unknown option %r" % name)
        return old

    def get_option(self, name):
        return self.lib.rgpu_get_option(name.encode())

    # ---- host side: parameter file and initial condition ------------------------------------------------------
    def params_from_ini(self, ini_path, overrides="", slab=None):
        p = RgpuParams()
        err = C.create_string_buffer(512)
        ov = overrides or ""
        if slab is not None:
            ov = (ov + ";" if ov else "") + "slab.rank=%d;slab.count=%d" % slab
        rc = self.lib.rgpuh_params_from_ini(ini_path.encode(), ov.encode(), C.byref(p), err, 512)
        if rc:
            raise RgpuError("params_from_ini(%s): %s" % (ini_path, err.value.decode()))
        return p

    def run_settings(self, ini_path, overrides=""):
        n, t, o = C.c_int(), C.c_double(), C.c_int()
        err = C.create_string_buffer(512)
        rc = self.lib.rgpuh_run_settings(ini_path.encode(), (overrides or "").encode(), C.byref(n), C.byref(t), C.byref(o), err, 512)
        if rc:
            raise RgpuError(err.value.decode())
        return {"nStepmax": n.value, "tEnd": t.value, "nOutput": o.value}

    def init_condition(self, ini_path, overrides, params):
        U = np.zeros(params.shape, dtype=np.float64)
        err = C.create_string_buffer(512)
        rc = self.lib.rgpuh_init_condition(ini_path.encode(), (overrides or "").encode(), C.byref(params), U.ctypes.data, err, 512)
        if rc:
            raise RgpuError("init_condition: %s" % err.value.decode())
        return U

    def init_forcing(self, ini_path, overrides, params):
        """static driving field [3][ksize][jsize][isize] of the "turbulence" problem (params.randomForcingEnabled), else None"""
        if not int(params.randomForcingEnabled):
            return None
        F = np.zeros((3,) + tuple(params.shape[1:]), dtype=np.float64)
        err = C.create_string_buffer(512)
        rc = self.lib.rgpuh_init_forcing(ini_path.encode(), (overrides or "").encode(), C.byref(params), F.ctypes.data, err, 512)
        if rc < 0:
            raise RgpuError("init_forcing: %s" % err.value.decode())
        return F if rc == 1 else None

    def init_gravity(self, ini_path, overrides, params):
        """static gravity field [3][ksize][jsize][isize] of the problems that define one (params.gravityEnabled == 2),
        else None"""
        if int(params.gravityEnabled) != 2:
            return None
        G = np.zeros((3,) + tuple(params.shape[1:]), dtype=np.float64)
        err = C.create_string_buffer(512)
        rc = self.lib.rgpuh_init_gravity(ini_path.encode(), (overrides or "").encode(), C.byref(params), G.ctypes.data, err, 512)
        if rc < 0:
            raise RgpuError("init_gravity: %s" % err.value.decode())
        return G if rc == 1 else None


_default = None


def load_library(path=None):
    global _default
    if path is None:
        if _default is None:
            _default = Library(lib_path())
        return _default
    return Library(path)


class Solver:
    """One run object == one rgpu_ctx (one GPU / one z-slab)."""

    def __init__(self, params, library=None, external_state=None, stream=0):
        self.L = library or load_library()
        self.lib = self.L.lib
        self.p = params
        self.ctx = C.c_void_p()
        if external_state is None:
            rc = self.lib.rgpu_create(C.byref(params), C.byref(self.ctx))
        else:
            dU, dU2 = external_state
            rc = self.lib.rgpu_create_external(C.byref(params), C.c_void_p(dU), C.c_void_p(dU2), C.c_void_p(stream), C.byref(self.ctx))
        if rc:
            msg = self.lib.rgpu_last_error(self.ctx).decode() if self.ctx else "?"
            self.close()
            raise RgpuError("rgpu_create failed (%d): %s" % (rc, msg))
        self.nStep = 0
        self.totalTime = 0.0
        self.dt = 0.0

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.rgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc:
            raise RgpuError("%s failed (%d): %s" % (what, rc, self.lib.rgpu_last_error(self.ctx).decode()))

    # ---- reference interface ----------------------------------------------------------------------------------
    def upload(self, hU, both=True):
        hU = np.ascontiguousarray(hU, dtype=np.float64)
        assert hU.shape == tuple(self.p.shape), (hU.shape, self.p.shape)
        self._chk(self.lib.rgpu_upload(self.ctx, hU.ctypes.data, int(both)), "upload")

    def getDataHost(self, nStep=None):
        """copyGpuToCpu(nStep) + getDataHost(nStep)"""
        parity = (self.nStep if nStep is None else nStep) % 2
        out = np.empty(self.p.shape, dtype=np.float64)
        self._chk(self.lib.rgpu_download(self.ctx, out.ctypes.data, parity), "download")
        return out

    def make_boundaries(self, parity, idim):
        self._chk(self.lib.rgpu_make_boundaries(self.ctx, parity, idim), "make_boundaries")

    def make_all_boundaries(self, parity=0, totalTime=0.0, dt=0.0):
        self._chk(self.lib.rgpu_make_all_boundaries(self.ctx, parity, totalTime, dt), "make_all_boundaries")

    def compute_inv_dt(self, useU=0):
        v = C.c_double()
        self._chk(self.lib.rgpu_compute_inv_dt(self.ctx, useU, C.byref(v)), "compute_inv_dt")
        return v.value

    def compute_dt(self, useU=0):
        d = self.lib.rgpu_compute_dt(self.ctx, useU)
        if d != d:
            raise RgpuError("compute_dt: " + self.lib.rgpu_last_error(self.ctx).decode())
        return d

    HISTORY_NAMES = ("mass", "maxwell", "reynolds", "magp", "mean_Bx", "mean_By", "mean_Bz", "divB")

    def set_gravity_field(self, G):
        """upload h_gravity (gravityEnabled == 2): [3][ksize][jsize][isize] doubles"""
        G = np.ascontiguousarray(G, dtype=np.float64)
        assert G.shape == (3,) + tuple(self.p.shape[1:]), G.shape
        self._chk(self.lib.rgpu_set_gravity_field(self.ctx, G.ctypes.data), "set_gravity_field")

    def set_forcing_field(self, F):
        """upload h_randomForcing (randomForcingEnabled): [3][ksize][jsize][isize] doubles"""
        F = np.ascontiguousarray(F, dtype=np.float64)
        assert F.shape == (3,) + tuple(self.p.shape[1:]), F.shape
        self._chk(self.lib.rgpu_set_forcing_field(self.ctx, F.ctypes.data), "set_forcing_field")

    def forcing_sums(self, parity):
        out = (C.c_double * 2)()
        self._chk(self.lib.rgpu_forcing_sums(self.ctx, parity, out), "forcing_sums")
        return float(out[0]), float(out[1])

    def add_forcing(self, parity, norm):
        self._chk(self.lib.rgpu_add_forcing(self.ctx, parity, float(norm)), "add_forcing")

    def read_cell(self, parity, i, j, k=0):
        """U(i, j, k, :) of one cell, ghost-inclusive local indices (the probe of history_inertial_wave)"""
        out = np.zeros(int(self.p.nbVar))
        self._chk(self.lib.rgpu_read_cell(self.ctx, parity, int(i), int(j), int(k), out.ctypes.data_as(_capi.c_double_p)), "read_cell")
        return out

    def history_mri(self, nStep=None):
        """history_mri / history_default (MHDRunBase.cpp:3311-3619) reduced on the device"""
        parity = (self.nStep if nStep is None else nStep) % 2
        out = (C.c_double * 8)()
        self._chk(self.lib.rgpu_history_mri(self.ctx, parity, out), "history_mri")
        return dict(zip(self.HISTORY_NAMES, [float(v) for v in out]))

    def state_checksum(self, parity):
        """sum mod 2^64 of the bit patterns of all interior doubles of U[parity] (rgpu_state_checksum): equal for every slab count"""
        out = C.c_ulonglong(0)
        self._chk(self.lib.rgpu_state_checksum(self.ctx, parity, C.byref(out)), "state_checksum")
        return int(out.value)

    def history_turbulence(self, nStep=None):
        """history_turbulence (MHDRunBase.cpp:3626-3810) reduced on the device: the 18 columns after totalTime and dt"""
        parity = (self.nStep if nStep is None else nStep) % 2
        out = (C.c_double * 18)()
        self._chk(self.lib.rgpu_history_turbulence(self.ctx, parity, out), "history_turbulence")
        return [float(v) for v in out]

    def history_columns(self, parity):
        isz = self.p.nx + 2 * self.p.ghostWidth
        cols = np.zeros((9, isz), dtype=np.float64)
        self._chk(self.lib.rgpu_history_columns(self.ctx, parity, cols.ctypes.data_as(C.POINTER(C.c_double))), "history_columns")
        return cols

    def history_reynolds(self, parity, mean_vx, mean_vy, dTau):
        isz = self.p.nx + 2 * self.p.ghostWidth
        mvx, mvy = np.ascontiguousarray(mean_vx, dtype=np.float64), np.ascontiguousarray(mean_vy, dtype=np.float64)
        col = np.zeros(isz, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        self._chk(self.lib.rgpu_history_reynolds(self.ctx, parity, mvx.ctypes.data_as(dp), mvy.ctypes.data_as(dp), dTau,
                                                 col.ctypes.data_as(dp)), "history_reynolds")
        return col

    def godunov_unsplit(self, nStep, dt, totalTime=None):
        t = self.totalTime if totalTime is None else totalTime
        self._chk(self.lib.rgpu_godunov_unsplit(self.ctx, nStep, dt, t), "godunov_unsplit")

    def oneStepIntegration(self):
        n, t, d = C.c_int(self.nStep), C.c_double(self.totalTime), C.c_double(self.dt)
        self._chk(self.lib.rgpu_one_step_integration(self.ctx, C.byref(n), C.byref(t), C.byref(d)), "oneStepIntegration")
        self.nStep, self.totalTime, self.dt = n.value, t.value, d.value
        return self.dt

    def run_steps(self, nsteps, tEnd=float("inf")):
        """up to nsteps turns of the reference's time loop (rgpu_run_steps: where a step is one fused kernel the time step stays on
        the device and the batch is queued without a host round trip); returns the number of steps done"""
        n, t, d = C.c_int(self.nStep), C.c_double(self.totalTime), C.c_double(self.dt)
        log = (C.c_double * max(int(nsteps), 1))()
        done = self.lib.rgpu_run_steps_log(self.ctx, int(nsteps), float(tEnd), C.byref(n), C.byref(t), C.byref(d), log)
        if done < 0:
            self._chk(done, "run_steps")
        self.nStep, self.totalTime, self.dt = n.value, t.value, d.value
        self.dt_log = [log[i] for i in range(done)]     # the time steps of this call, in order
        return done

    def start(self, hU, nStepmax, tEnd=float("inf")):
        """init part + time loop of start() (MHDRunGodunov.cpp:3801-3989) without outputs; returns the dt list"""
        self.upload(hU, both=False)
        self.nStep, self.totalTime = 0, 0.0
        self.make_all_boundaries(0, 0.0, 0.0)
        # h_U.copyTo(h_U2)
        self.upload(self.getDataHost(0), both=True)
        dts = []
        while self.totalTime < tEnd and self.nStep < nStepmax:
            dts.append(self.oneStepIntegration())
        return dts

    # ---- pieces used by the slab driver -------------------------------------------------------------------------
    def step_pre(self, nStep, dt, t):
        self._chk(self.lib.rgpu_step_pre(self.ctx, nStep, dt, t), "step_pre")

    def step_core(self, nStep, dt, t):
        self._chk(self.lib.rgpu_step_core(self.ctx, nStep, dt, t), "step_core")

    def step_core_planes(self, nStep, dt, t, k_lo, k_hi):
        self._chk(self.lib.rgpu_step_core_planes(self.ctx, nStep, dt, t, k_lo, k_hi), "step_core_planes")

    def step_core_planes_split(self, nStep, dt, t, k_lo, k_hi, what):
        """what: 1 = RGPU_CORE_FLUXES, 2 = RGPU_CORE_UPDATE (include/rgpu.h)"""
        self._chk(self.lib.rgpu_step_core_planes_split(self.ctx, nStep, dt, t, k_lo, k_hi, what), "step_core_planes_split")

    def step_fill_planes(self, nStep, dt, t, k_lo, k_hi):
        self._chk(self.lib.rgpu_step_fill_planes(self.ctx, nStep, dt, t, k_lo, k_hi), "step_fill_planes")

    def step_core_planes_pair(self, nStep, dt, t, r1, r2, what):
        """two disjoint plane ranges in one call (rgpu_step_core_planes_pair): what as step_core_planes_split"""
        self._chk(self.lib.rgpu_step_core_planes_pair(self.ctx, nStep, dt, t, r1[0], r1[1], r2[0], r2[1], what), "step_core_planes_pair")

    def step_fill_planes_pair(self, nStep, dt, t, r1, r2):
        self._chk(self.lib.rgpu_step_fill_planes_pair(self.ctx, nStep, dt, t, r1[0], r1[1], r2[0], r2[1]), "step_fill_planes_pair")

    def inv_dt_accumulate(self, parity, k_lo, k_hi, reset=False):
        self._chk(self.lib.rgpu_inv_dt_accumulate(self.ctx, parity, k_lo, k_hi, int(reset)), "inv_dt_accumulate")

    def inv_dt_result(self):
        v = C.c_double(0.0)
        self._chk(self.lib.rgpu_inv_dt_result(self.ctx, C.byref(v)), "inv_dt_result")
        return v.value

    def step_dissipative(self, nStep, dt, t):
        self._chk(self.lib.rgpu_step_dissipative(self.ctx, nStep, dt, t), "step_dissipative")

    def step_post_a(self, nStep, dt, t):
        self._chk(self.lib.rgpu_step_post_a(self.ctx, nStep, dt, t), "step_post_a")

    def step_post_b(self, nStep, dt, t):
        self._chk(self.lib.rgpu_step_post_b(self.ctx, nStep, dt, t), "step_post_b")

    def synchronize(self):
        self._chk(self.lib.rgpu_synchronize(self.ctx), "synchronize")

    # ---- instrumentation ----------------------------------------------------------------------------------------
    def enable_timers(self, on=True):
        self._chk(self.lib.rgpu_enable_timers(self.ctx, int(on)), "enable_timers")

    def reset_timers(self):
        self._chk(self.lib.rgpu_reset_timers(self.ctx), "reset_timers")

    def timers(self):
        a = (C.c_double * len(_capi.T_NAMES))()
        self._chk(self.lib.rgpu_get_timers(self.ctx, a, len(_capi.T_NAMES)), "get_timers")
        return dict(zip(_capi.T_NAMES, list(a)))

    def dominant_kernel(self):
        name = C.create_string_buffer(64)
        ms, n = C.c_double(), C.c_long()
        self._chk(self.lib.rgpu_dominant_kernel(self.ctx, name, 64, C.byref(ms), C.byref(n)), "dominant_kernel")
        return name.value.decode(), ms.value, n.value


def interior(U, p):
    """strip the ghost cells of a [nvar][k][j][i] array"""
    gw = p.ghostWidth
    if p.three_d:
        return U[:, gw:-gw, gw:-gw, gw:-gw]
    return U[:, :, gw:-gw, gw:-gw]
