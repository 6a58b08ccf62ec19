"""ctypes view of include/rgpu.h (the C ABI) -- plain structs and prototypes, no compute here."""
import ctypes as C

RGPU_ABI_VERSION = 5

ID, IP, IU, IV, IW, IA, IB, IC = range(8)
BC_UNDEFINED, BC_DIRICHLET, BC_NEUMANN, BC_PERIODIC, BC_SHEARINGBOX, BC_COPY, BC_Z_STRATIFIED = range(7)
RS_APPROX, RS_HLL, RS_HLLC, RS_HLLD, RS_LLF = range(5)
XDIR, YDIR, ZDIR = 1, 2, 3
T_NAMES = ["boundaries", "prim", "elec", "trace", "flux", "emf", "update", "shear", "dt", "dissipative", "sweep"]


class RgpuParams(C.Structure):
    """struct rgpu_params (include/rgpu.h) -- field order and types must match exactly."""
    _fields_ = [
        ("abi_version", C.c_int32),
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
        ("ghostWidth", C.c_int32),
        ("nbVar", C.c_int32),
        ("mhdEnabled", C.c_int32),
        ("bc", C.c_int32 * 6),
        ("xMin", C.c_double), ("xMax", C.c_double), ("yMin", C.c_double), ("yMax", C.c_double),
        ("zMin", C.c_double), ("zMax", C.c_double),
        ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
        ("cfl", C.c_double),
        ("gamma0", C.c_double), ("cIso", C.c_double), ("smallr", C.c_double), ("smallc", C.c_double),
        ("smalle", C.c_double), ("smallp", C.c_double), ("smallpp", C.c_double), ("gamma6", C.c_double),
        ("Omega0", C.c_double),
        ("slope_type", C.c_double),
        ("niter_riemann", C.c_int32), ("iorder", C.c_int32),
        ("riemannSolver", C.c_int32),
        ("magRiemannSolver", C.c_int32),
        ("implementationVersion", C.c_int32),
        ("unsplitVersion", C.c_int32),
        ("shearingBoxEnabled", C.c_int32),
        ("enableJet", C.c_int32), ("ijet", C.c_int32), ("offsetJet", C.c_int32),
        ("djet", C.c_double), ("ujet", C.c_double), ("pjet", C.c_double), ("cjet", C.c_double),
        ("slab_rank", C.c_int32), ("slab_count", C.c_int32),
        ("nz_global", C.c_int32),
        ("gravityEnabled", C.c_int32),
        ("gravity_x", C.c_double), ("gravity_y", C.c_double), ("gravity_z", C.c_double),
        ("nu", C.c_double), ("eta", C.c_double),
        ("zStratifiedFloor", C.c_int32), ("randomForcingEnabled", C.c_int32), ("randomForcingEdot", C.c_double),
        ("ouForcingEnabled", C.c_int32), ("ouInitRandom", C.c_int32),
        ("ouTimeScaleTurb", C.c_double), ("ouAmplitudeTurb", C.c_double), ("ouKsi", C.c_double),
    ]

    @property
    def three_d(self):
        return self.nz_global != 1

    @property
    def shape(self):
        """(nbVar, ksize, jsize, isize) of a ghost-inclusive state array."""
        gw = self.ghostWidth
        ks = self.nz + 2 * gw if self.three_d else 1
        return (self.nbVar, ks, self.ny + 2 * gw, self.nx + 2 * gw)

    def copy(self):
        q = RgpuParams()
        C.memmove(C.byref(q), C.byref(self), C.sizeof(RgpuParams))
        return q


c_double_p = C.POINTER(C.c_double)


def declare_host_api(lib):
    """prototypes of the rgpuh_* entry points + rgpu_state_elems"""
    lib.rgpu_state_elems.restype = C.c_size_t
    lib.rgpu_state_elems.argtypes = [C.POINTER(RgpuParams)]
    lib.rgpuh_params_from_ini.restype = C.c_int
    lib.rgpuh_params_from_ini.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(RgpuParams), C.c_char_p, C.c_int]
    lib.rgpuh_run_settings.restype = C.c_int
    lib.rgpuh_run_settings.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), c_double_p, C.POINTER(C.c_int),
                                       C.c_char_p, C.c_int]
    lib.rgpuh_init_condition.restype = C.c_int
    lib.rgpuh_init_condition.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(RgpuParams), C.c_void_p, C.c_char_p, C.c_int]
    lib.rgpuh_init_forcing.restype = C.c_int
    lib.rgpuh_init_forcing.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(RgpuParams), C.c_void_p, C.c_char_p, C.c_int]
    lib.rgpuh_init_gravity.restype = C.c_int
    lib.rgpuh_init_gravity.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(RgpuParams), C.c_void_p, C.c_char_p, C.c_int]
    return lib


def declare_device_api(lib):
    """prototypes of the rgpu_* entry points (include/rgpu.h)"""
    P = C.POINTER(RgpuParams)
    ctx = C.c_void_p
    lib.rgpu_create.restype = C.c_int
    lib.rgpu_create.argtypes = [P, C.POINTER(ctx)]
    lib.rgpu_create_external.restype = C.c_int
    lib.rgpu_create_external.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ctx)]
    lib.rgpu_destroy.restype = None
    lib.rgpu_destroy.argtypes = [ctx]
    lib.rgpu_device_bytes.restype = C.c_size_t
    lib.rgpu_device_bytes.argtypes = [P]
    lib.rgpu_last_error.restype = C.c_char_p
    lib.rgpu_last_error.argtypes = [ctx]
    lib.rgpu_upload.restype = C.c_int
    lib.rgpu_upload.argtypes = [ctx, C.c_void_p, C.c_int]
    lib.rgpu_download.restype = C.c_int
    lib.rgpu_download.argtypes = [ctx, C.c_void_p, C.c_int]
    lib.rgpu_device_state.restype = C.c_void_p
    lib.rgpu_device_state.argtypes = [ctx, C.c_int]
    lib.rgpu_state_checksum.restype = C.c_int
    lib.rgpu_state_checksum.argtypes = [ctx, C.c_int, C.POINTER(C.c_ulonglong)]
    lib.rgpu_history_turbulence.restype = C.c_int
    lib.rgpu_history_turbulence.argtypes = [ctx, C.c_int, c_double_p]
    lib.rgpu_step_ou_forcing.restype = C.c_int
    lib.rgpu_step_ou_forcing.argtypes = [ctx, C.c_int, C.c_double]
    lib.rgpu_ou_forcing_get_state.restype = C.c_int
    lib.rgpu_ou_forcing_get_state.argtypes = [ctx, c_double_p]
    lib.rgpu_ou_forcing_set_state.restype = C.c_int
    lib.rgpu_ou_forcing_set_state.argtypes = [ctx, c_double_p]
    lib.rgpu_ou_forcing_state.restype = C.c_int
    lib.rgpu_ou_forcing_state.argtypes = [ctx, c_double_p, c_double_p]
    lib.rgpu_get_params.restype = C.c_int
    lib.rgpu_get_params.argtypes = [ctx, P]
    lib.rgpu_stream_handle.restype = C.c_void_p
    lib.rgpu_stream_handle.argtypes = [ctx]
    lib.rgpu_inv_dt_device_slot.restype = C.c_void_p
    lib.rgpu_inv_dt_device_slot.argtypes = [ctx]
    lib.rgpu_make_boundaries.restype = C.c_int
    lib.rgpu_make_boundaries.argtypes = [ctx, C.c_int, C.c_int]
    lib.rgpu_make_boundaries_shear.restype = C.c_int
    lib.rgpu_make_boundaries_shear.argtypes = [ctx, C.c_int, C.c_double, C.c_double]
    lib.rgpu_make_all_boundaries.restype = C.c_int
    lib.rgpu_make_all_boundaries.argtypes = [ctx, C.c_int, C.c_double, C.c_double]
    lib.rgpu_compute_inv_dt.restype = C.c_int
    lib.rgpu_compute_inv_dt.argtypes = [ctx, C.c_int, c_double_p]
    lib.rgpu_history_columns.restype = C.c_int
    lib.rgpu_history_columns.argtypes = [ctx, C.c_int, c_double_p]
    lib.rgpu_history_reynolds.restype = C.c_int
    lib.rgpu_history_reynolds.argtypes = [ctx, C.c_int, c_double_p, c_double_p, C.c_double, c_double_p]
    lib.rgpu_set_forcing_field.restype = C.c_int
    lib.rgpu_set_forcing_field.argtypes = [ctx, C.c_void_p]
    lib.rgpu_forcing_sums.restype = C.c_int
    lib.rgpu_forcing_sums.argtypes = [ctx, C.c_int, c_double_p]
    lib.rgpu_add_forcing.restype = C.c_int
    lib.rgpu_add_forcing.argtypes = [ctx, C.c_int, C.c_double]
    lib.rgpu_set_gravity_field.restype = C.c_int
    lib.rgpu_set_gravity_field.argtypes = [ctx, C.c_void_p]
    lib.rgpu_history_mri.restype = C.c_int
    lib.rgpu_history_mri.argtypes = [ctx, C.c_int, c_double_p]
    lib.rgpu_read_cell.restype = C.c_int
    lib.rgpu_read_cell.argtypes = [ctx, C.c_int, C.c_int, C.c_int, C.c_int, c_double_p]
    lib.rgpu_invalidate_dt.restype = C.c_int
    lib.rgpu_invalidate_dt.argtypes = [ctx]
    lib.rgpu_compute_dt.restype = C.c_double
    lib.rgpu_compute_dt.argtypes = [ctx, C.c_int]
    lib.rgpu_inv_dt_fused_active.restype = C.c_int
    lib.rgpu_inv_dt_fused_active.argtypes = [C.c_void_p, C.c_int]
    lib.rgpu_inv_dt_fused_commit.restype = C.c_int
    lib.rgpu_inv_dt_fused_commit.argtypes = [C.c_void_p, C.c_int]
    lib.rgpu_step_core_planes_split.restype = C.c_int
    lib.rgpu_step_core_planes_split.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    lib.rgpu_step_core_planes_pair.restype = C.c_int
    lib.rgpu_step_core_planes_pair.argtypes = [ctx, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rgpu_step_fill_planes_pair.restype = C.c_int
    lib.rgpu_step_fill_planes_pair.argtypes = [ctx, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int]
    for name in ("rgpu_step_core_planes", "rgpu_step_fill_planes"):
        f = getattr(lib, name)
        f.restype = C.c_int
        f.argtypes = [ctx, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
    lib.rgpu_inv_dt_accumulate.restype = C.c_int
    lib.rgpu_inv_dt_accumulate.argtypes = [ctx, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rgpu_inv_dt_result.restype = C.c_int
    lib.rgpu_inv_dt_result.argtypes = [ctx, c_double_p]
    for name in ("rgpu_godunov_unsplit", "rgpu_step_pre", "rgpu_step_core", "rgpu_step_dissipative", "rgpu_step_post_a", "rgpu_step_post_b"):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [ctx, C.c_int, C.c_double, C.c_double]
    lib.rgpu_one_step_integration.restype = C.c_int
    lib.rgpu_one_step_integration.argtypes = [ctx, C.POINTER(C.c_int), c_double_p, c_double_p]
    lib.rgpu_device_time_step_ready.restype = C.c_int
    lib.rgpu_device_time_step_ready.argtypes = [ctx, C.c_int]
    lib.rgpu_run_steps_log.restype = C.c_int
    lib.rgpu_run_steps_log.argtypes = [ctx, C.c_int, C.c_double, C.POINTER(C.c_int), c_double_p, c_double_p, c_double_p]
    lib.rgpu_run_steps.restype = C.c_int
    lib.rgpu_run_steps.argtypes = [ctx, C.c_int, C.c_double, C.POINTER(C.c_int), c_double_p, c_double_p]
    lib.rgpu_synchronize.restype = C.c_int
    lib.rgpu_synchronize.argtypes = [ctx]
    lib.rgpu_enable_timers.restype = C.c_int
    lib.rgpu_enable_timers.argtypes = [ctx, C.c_int]
    lib.rgpu_get_timers.restype = C.c_int
    lib.rgpu_get_timers.argtypes = [ctx, c_double_p, C.c_int]
    lib.rgpu_reset_timers.restype = C.c_int
    lib.rgpu_reset_timers.argtypes = [ctx]
    lib.rgpu_timer_name.restype = C.c_char_p
    lib.rgpu_timer_name.argtypes = [C.c_int]
    lib.rgpu_dominant_kernel.restype = C.c_int
    lib.rgpu_dominant_kernel.argtypes = [ctx, C.c_char_p, C.c_int, c_double_p, C.POINTER(C.c_long)]
    lib.rgpu_backend_name.restype = C.c_char_p
    lib.rgpu_backend_name.argtypes = []
    lib.rgpu_arithmetic.restype = C.c_char_p
    lib.rgpu_arithmetic.argtypes = []
    lib.rgpu_set_option.restype = C.c_int
    lib.rgpu_set_option.argtypes = [C.c_char_p, C.c_int]
    lib.rgpu_get_option.restype = C.c_int
    lib.rgpu_get_option.argtypes = [C.c_char_p]
    lib.rgpu_clock_check.restype = C.c_int
    lib.rgpu_clock_check.argtypes = [ctx]
    return lib


# every symbol include/rgpu.h declares (checked by tests/test_abi.py against the built library)
DECLARED_SYMBOLS = [
    "rgpu_create", "rgpu_create_external", "rgpu_destroy", "rgpu_state_elems", "rgpu_device_bytes", "rgpu_last_error",
    "rgpu_upload", "rgpu_download", "rgpu_device_state", "rgpu_get_params", "rgpu_stream_handle", "rgpu_inv_dt_device_slot", "rgpu_make_boundaries", "rgpu_make_boundaries_shear",
    "rgpu_make_all_boundaries", "rgpu_history_columns", "rgpu_history_reynolds", "rgpu_history_mri", "rgpu_history_turbulence", "rgpu_history_turbulence_sums", "rgpu_state_checksum", "rgpu_read_cell", "rgpu_compute_inv_dt", "rgpu_invalidate_dt", "rgpu_compute_dt", "rgpu_godunov_unsplit", "rgpu_step_pre",
    "rgpu_step_core", "rgpu_step_dissipative", "rgpu_step_core_planes", "rgpu_step_core_planes_split", "rgpu_inv_dt_fused_commit", "rgpu_inv_dt_fused_active", "rgpu_inv_dt_fusable", "rgpu_step_fill_planes", "rgpu_step_core_planes_pair", "rgpu_step_fill_planes_pair", "rgpu_inv_dt_accumulate", "rgpu_inv_dt_result",
    "rgpu_step_post_a", "rgpu_step_post_b", "rgpu_one_step_integration", "rgpu_run_steps", "rgpu_run_steps_log", "rgpu_device_time_step_ready", "rgpu_clock_capable", "rgpu_clock_open", "rgpu_clock_tick", "rgpu_clock_close", "rgpu_clock_stopped", "rgpu_clock_check", "rgpu_set_option", "rgpu_get_option", "rgpu_synchronize",
    "rgpu_enable_timers", "rgpu_get_timers", "rgpu_reset_timers", "rgpu_timer_name", "rgpu_dominant_kernel",
    "rgpu_backend_name", "rgpu_arithmetic", "rgpu_selftest_arith", "rgpu_selftest_alfven", "rgpu_step_ou_forcing", "rgpu_ou_forcing_state", "rgpu_ou_forcing_get_state", "rgpu_ou_forcing_set_state", "rgpuh_params_from_ini", "rgpuh_run_settings", "rgpuh_init_condition", "rgpuh_init_gravity", "rgpu_set_gravity_field", "rgpuh_init_forcing", "rgpu_set_forcing_field", "rgpu_forcing_sums", "rgpu_add_forcing", "rgpuh_run", "rgpuh_run_hooked",
]
