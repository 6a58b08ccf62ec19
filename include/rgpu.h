/*
 * rgpu.h -- C ABI of the MI355X-native Godunov / MUSCL-Hancock unsplit time step
 *           (hydro HLLC/approx/HLL, MHD HLLD + constrained transport, shearing box).
 *
 * This is the drop-in boundary for ONE hot path of pkestene/ramsesGPU: the body of
 *   HydroRunBase::oneStepIntegration(int& nStep, real_t& t, real_t& dt)   (HydroRunBase.h:433)
 *     = compute_dt[_mhd](nStep % 2) + godunov_unsplit(nStep, dt)          (MHDRunGodunov.cpp:4077-4089,
 *                                                                           HydroRunGodunov.cpp:4082-4126)
 * plus the ghost fill it calls (make_all_boundaries / make_all_boundaries_shear).
 * The reference has no FFI; its seam is that C++ virtual interface with state in the protected members
 * h_U,h_U2,d_U,d_U2 (HydroRunBase.h:556-566).  A maintainer binds these entry points from the run classes
 * (see INTEGRATION.md).  Everything is plain C: pointers, sizes, doubles.  No torch / HIP types.
 *
 * Array layout (contract shared with the reference's HostArray/DeviceArray, Arrays.h:95-98,236-239):
 *   U[ i + isize*( j + jsize*( k + ksize*ivar ) ) ],  fp64, ghost cells included,
 *   isize = nx+2*ghostWidth, jsize = ny+2*ghostWidth, ksize = nz+2*ghostWidth (1 in 2D),
 *   component order ID,IP,IU,IV,IW,IA,IB,IC (constants.h:59-71); hydro uses the first 4 (2D) / 5 (3D).
 *   Sizes are 64-bit here (the reference's uint sizes overflow at 518^3 x 8).
 *
 * Error model: every int entry point returns 0 on success or a negative RGPU_E* code; the message is kept in
 * the context (rgpu_last_error).  The reference exit()s on CUDA failure (cutil_inline_runtime.h:167-174); a
 * library must not.  Not thread-safe per context (the reference is single-threaded per run object).
 * All entry points return with their results visible to the next call (internally asynchronous on one stream).
 */
#ifndef RGPU_H_
#define RGPU_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RGPU_ABI_VERSION 5   /* layout of rgpu_params; new entry points do not change it */

/* component indexes -- constants.h:59-71 */
enum { RGPU_ID = 0, RGPU_IP = 1, RGPU_IU = 2, RGPU_IV = 3, RGPU_IW = 4, RGPU_IA = 5, RGPU_IB = 6, RGPU_IC = 7 };

/* BoundaryConditionType -- constants.h:209-217 */
enum {
  RGPU_BC_UNDEFINED = 0, RGPU_BC_DIRICHLET = 1, RGPU_BC_NEUMANN = 2, RGPU_BC_PERIODIC = 3,
  RGPU_BC_SHEARINGBOX = 4,
  RGPU_BC_COPY = 5 /* ghost planes are supplied by a neighbour slab (or any external z driver).  The planes written into the ghost region of
                    * such a face must be COMPLETE planes of the neighbour's state -- its x / y ghost cells and corners included, i.e. taken
                    * after the neighbour's own x / y fill: the library fills x / y ghosts on the interior planes only and does not re-run
                    * the X / Y passes over received planes (librgpu_comm.so sends such planes) */,
  RGPU_BC_Z_STRATIFIED = 6 /* z faces of the vertically stratified MRI box (3D MHD, isothermal, Omega0 > 0):
                            * make_boundary2_z_stratified, make_boundary_base.h:1356-1647 */
};

/* RiemannSolverType -- constants.h:140-146 */
enum { RGPU_RS_APPROX = 0, RGPU_RS_HLL = 1, RGPU_RS_HLLC = 2, RGPU_RS_HLLD = 3, RGPU_RS_LLF = 4 };
/* MagneticRiemannSolverType -- constants.h:149-156 */
enum { RGPU_MAG_HLLD = 0, RGPU_MAG_HLLF = 1, RGPU_MAG_HLLA = 2, RGPU_MAG_ROE = 3, RGPU_MAG_LLF = 4, RGPU_MAG_UPWIND = 5 };

/* direction ids used by rgpu_make_boundaries -- constants.h:220 (XDIR=1,YDIR=2,ZDIR=3) */
enum { RGPU_XDIR = 1, RGPU_YDIR = 2, RGPU_ZDIR = 3 };

/* error codes */
enum {
  RGPU_OK = 0, RGPU_EINVAL = -1, RGPU_ENODEVICE = -2, RGPU_ENOMEM = -3, RGPU_EHIP = -4, RGPU_EUNSUPPORTED = -5
};

/*
 * Scalar knobs of one run: 1:1 with GlobalConstants (constants.h:277-317) and the HydroParameters members the
 * path reads (HydroParameters.h:73-136), with the derivations of HydroParameters.h:196-325 ALREADY APPLIED by
 * the host (float-parsed values widened to double, smallp/smallpp/gamma6 derived, MHD => ghostWidth 3, nbVar 8).
 */
typedef struct rgpu_params {
  int32_t abi_version;          /* must be RGPU_ABI_VERSION */
  int32_t nx, ny, nz;           /* interior cells of THIS domain (a z-slab when slab_count>1); nz=1 => 2D */
  int32_t ghostWidth;           /* 2 hydro, 3 MHD (HydroParameters.h:260-271) */
  int32_t nbVar;                /* 4 hydro 2D, 5 hydro 3D, 8 MHD (HydroParameters.h:204-235) */
  int32_t mhdEnabled;
  int32_t bc[6];                /* xmin,xmax,ymin,ymax,zmin,zmax (HydroParameters.h:253-258) */
  double  xMin, xMax, yMin, yMax, zMin, zMax;   /* GLOBAL box (HydroParameters.h:238-243) */
  double  dx, dy, dz;           /* HydroParameters.h:245-247 (dz from the GLOBAL nz) */
  double  cfl;                  /* HydroParameters.h:274-282 */
  double  gamma0, cIso, smallr, smallc, smalle, smallp, smallpp, gamma6;   /* HydroParameters.h:292-312 */
  double  Omega0;               /* [MHD] omega0 (HydroParameters.h:313) */
  double  slope_type;           /* HydroParameters.h:319-321 */
  int32_t niter_riemann, iorder;
  int32_t riemannSolver;        /* RGPU_RS_*  (HydroParameters.h:353-381) */
  int32_t magRiemannSolver;     /* RGPU_MAG_* (HydroParameters.h:388-417) */
  int32_t implementationVersion;/* [MHD] implementationVersion: 2D 0 or 1; 3D 3/4 (Omega0>0 forces 1/4, MHDRunGodunov.cpp:119-126) */
  int32_t unsplitVersion;       /* [hydro] unsplitVersion: 1 or 2 (HydroRunGodunov.cpp:1852-1866; 2 = direction-wise sweeps) */
  int32_t shearingBoxEnabled;   /* bc xmin==xmax==4 and Omega0>0 (MHDRunGodunov.cpp:97-103) */
  int32_t enableJet, ijet, offsetJet;  /* HydroParameters.h:435-444 */
  double  djet, ujet, pjet, cjet;
  /* z-slab decomposition (replaces the reference's MPI cartesian topology, HydroMpiParameters.cpp:44-80) */
  int32_t slab_rank, slab_count;/* 0,1 for a single device */
  int32_t nz_global;            /* == nz when slab_count==1 */
  int32_t gravityEnabled;       /* [gravity] static=yes, or forced by problem=Rayleigh-Taylor (HydroRunBase.cpp:250-260):
                                 * 1 = the uniform vector below, 2 = a per-cell field given with rgpu_set_gravity_field */
  /* uniform static gravity field ([gravity] static_field_x/y/z, HydroParameters.h:322-324): the reference keeps it in a
   * per-cell array h_gravity that its problems fill with exactly this vector (HydroRunBase.cpp:6336-6337, 6403-6405) */
  double  gravity_x, gravity_y, gravity_z;
  /* dissipative stage after the Godunov update (HydroParameters.h:327-328): kinematic viscosity [hydro] nu and
   * resistivity [MHD] eta; 0 = off */
  double  nu, eta;
  int32_t zStratifiedFloor;     /* [MRI] floor (HydroRunBase.cpp:2206): RGPU_BC_Z_STRATIFIED copies the density instead of
                                 * extrapolating the hydrostatic profile */
  int32_t randomForcingEnabled; /* problem "turbulence" (HydroRunBase.cpp:213-227): the static solenoidal driving field of
                                 * rgpu_set_forcing_field is added to the momenta at the end of every step, scaled so that
                                 * the kinetic energy input rate is randomForcingEdot (3D, no rotating frame) */
  double  randomForcingEdot;    /* [turbulence] edot, or the Mac Low (1999) estimate when negative (HydroRunBase.cpp:7175-7194) */
  /* problem "turbulence-Ornstein-Uhlenbeck" (HydroRunBase.cpp:230-247): 31 Fourier modes of a forcing field driven by an
   * Ornstein-Uhlenbeck process (ForcingOrnsteinUhlenbeck, Forcing_OrnsteinUhlenbeck.cpp) are advanced on the host every
   * step and their real-space sum accelerates the gas at the end of the step (3D, no rotating frame).  The context owns
   * the process (mode signs, projection tensor, the four-digit base-4096 seed of RandomGen); every z-slab runs the same one. */
  int32_t ouForcingEnabled;
  int32_t ouInitRandom;         /* [turbulence-Ornstein-Uhlenbeck] init_random (seed of the Gaussian generator) */
  double  ouTimeScaleTurb, ouAmplitudeTurb, ouKsi;   /* timeScaleTurb, amplitudeTurb, ksi (1 solenoidal .. 0 compressive) */
} rgpu_params;

typedef struct rgpu_ctx rgpu_ctx;

/* ---- life cycle ------------------------------------------------------------------------------------------ */

/* Allocates U, U2 and all scratch on the current HIP device.  Replaces the allocations of the HydroRunBase /
 * MHDRunGodunov constructors (HydroRunBase.cpp:92-300, MHDRunGodunov.cpp:128-418).  Fails with RGPU_ENODEVICE when
 * no GPU is present: there is NO CPU fallback.
 * Size limit: the kernels address a cell of one context with a 32-bit flat index (byte offsets and component strides are
 * 64-bit), so (nx + 2 gw)(ny + 2 gw)(nz + 2 gw) must stay below 2^32 - 1 cells PER CONTEXT -- 1619^3 with gw = 3; at the 34
 * doubles per cell of the 3D MHD step that is 1.17 TB, four times the 288 GB of one MI355X, so the memory is exhausted long
 * before the index.  rgpu_create checks it and fails with RGPU_EUNSUPPORTED ("more than 2^32 cells per device"); larger boxes
 * run as z-slabs (rgpu_comm.h), every slab its own context.  (The reference's 32-bit BYTE offsets overflow at 518^3 x 8.) */
int rgpu_create(const rgpu_params* p, rgpu_ctx** out);

/* Same, but U and U2 are device buffers owned by the caller (e.g. torch tensors) of rgpu_state_elems() doubles
 * each, and work is issued on the caller's hipStream_t (pass NULL for the default stream). */
int rgpu_create_external(const rgpu_params* p, double* dU, double* dU2, void* hip_stream, rgpu_ctx** out);

void rgpu_destroy(rgpu_ctx* c);

/* number of doubles in one state array: isize*jsize*ksize*nbVar */
size_t rgpu_state_elems(const rgpu_params* p);

/* bytes of device memory rgpu_create will allocate (U, U2 and scratch) */
size_t rgpu_device_bytes(const rgpu_params* p);

const char* rgpu_last_error(rgpu_ctx* c);

/* ---- host <-> device (init, output, history only; never inside the step) ---------------------------------- */

/* == d_U.copyFromHost(h_U) [+ d_U2] (MHDRunBase.cpp:1346-1351) */
int rgpu_upload(rgpu_ctx* c, const double* hU, int both);
/* == copyGpuToCpu(nStep) + getDataHost(nStep) (HydroRunBase.cpp:7217-7229, 2442-2447); parity = nStep%2 */
int rgpu_download(rgpu_ctx* c, double* hU, int parity);
/* Static gravity field of a context created with gravityEnabled = 2: hG[3][ksize][jsize][isize] (ghost cells included,
 * the z component ignored in 2D) = the reference's h_gravity / d_gravity (HydroRunBase.h, filled by the initial
 * conditions of Keplerian-disk, HydroRunBase.cpp:6489-6500, and of the stratified MRI box, MHDRunBase.cpp:3163-3211).
 * Until it is called the field is zero, like the reference's freshly allocated array. */
int rgpu_set_gravity_field(rgpu_ctx* c, const double* hG);
/* Driving of the "turbulence" problem (randomForcingEnabled): hF[3][ksize][jsize][isize] = the reference's
 * h_randomForcing / d_randomForcing (turbulenceInit.cpp, copied to the device at HydroRunBase.cpp:7206-7209).
 * rgpu_godunov_unsplit applies it after the dissipative stage like the reference (HydroRunGodunov.cpp:2930-2938,
 * mhd_godunov_unsplit_cpu_v3.cpp:696-704).  A z-slab driver does the same in pieces: rgpu_forcing_sums returns
 * out[2] = { sum rho v.f , sum rho f.f } over the interior of this domain (compute_random_forcing_normalization,
 * HydroRunBase.cpp:1201-1312; the other seven sums of the reference are debug output), the caller adds the slabs'
 * values, computes norm = (sqrt(s0^2 + s1 dt edot 2 nbCells) - s0) / s1 and calls rgpu_add_forcing
 * (add_random_forcing, HydroRunBase.cpp:1397-1428).  The sums are accumulated in a fixed order that is not the
 * reference's loop order: results agree with the reference to round-off, not bit for bit. */
int rgpu_set_forcing_field(rgpu_ctx* c, const double* hF);
/* ForcingOrnsteinUhlenbeck::add_forcing_field (Forcing_OrnsteinUhlenbeck.cpp:498-549, 597-686) on U[parity]: one
 * Ornstein-Uhlenbeck update of the 31 modes with time step dt (host; RandomGen::gaussDev), then momenta and total energy
 * of every interior cell get the real-space sum of the modes.  rgpu_godunov_unsplit calls it itself after the dissipative
 * stage like the reference (HydroRunGodunov.cpp:2940-2945, mhd_godunov_unsplit_cpu_v3.cpp:706-710); a z-slab driver calls
 * it on every slab (same process on every rank, no communication).  The device evaluates cos() with its own libm: states
 * agree with the reference to round-off (relative L2 ~1e-16 per step), not bit for bit.
 * rgpu_ou_forcing_state copies out mode[3][31], forcingField[3][31] (what output_forcing writes, :392-444). */
int rgpu_step_ou_forcing(rgpu_ctx* c, int parity, double dt);
int rgpu_ou_forcing_state(rgpu_ctx* c, double* mode93, double* forcingField93);
/* The whole process (modes, amplitudes, projection tensor, generator) as RGPU_OU_STATE_DOUBLES doubles, for restart files:
 * get before writing one, set after rgpu_create when resuming (the reference keeps a *_forcing_NNNNNNN.npz for the same
 * purpose, output_forcing / init_forcing(restart), Forcing_OrnsteinUhlenbeck.cpp:236-352, 392-444). */
#define RGPU_OU_STATE_DOUBLES 471
int rgpu_ou_forcing_get_state(rgpu_ctx* c, double* state);
int rgpu_ou_forcing_set_state(rgpu_ctx* c, const double* state);
int rgpu_forcing_sums(rgpu_ctx* c, int parity, double* out);
int rgpu_add_forcing(rgpu_ctx* c, int parity, double norm);
/* raw device pointers of U (parity 0) / U2 (parity 1), for zero-copy halo exchange */
double* rgpu_device_state(rgpu_ctx* c, int parity);
/* What a communication layer (include/rgpu_comm.h) needs to work on a context without host round trips: the parameters
 * the context was created with, the HIP stream its kernels are queued on (hipStream_t as void*), and the 8-byte device
 * slot the CFL scan leaves its maximum in (a non-negative double; rgpu_inv_dt_result reads it back) -- the operand
 * of the MAX all-reduce that replaces the reference's MPI allReduce of dt (HydroRunBaseMpi.cpp:509-513). */
int rgpu_get_params(rgpu_ctx* c, rgpu_params* out);
void* rgpu_stream_handle(rgpu_ctx* c);
double* rgpu_inv_dt_device_slot(rgpu_ctx* c);
/* The slot is the first of RGPU_DT_SLOTS doubles (zero at create): the update kernels that carry the CFL scan of the new state
 * spread their maxima over all of them (rgpu_inv_dt_fused_commit).  A communication layer all-reduces ALL RGPU_DT_SLOTS values,
 * whatever the step left in them -- a fixed count, so that ranks in different states (first step, a failed step piece) can never
 * post all-reduces of different sizes; rgpu_inv_dt_result reads the ones that are valid. */
#define RGPU_DT_SLOTS 1024
#define RGPU_CLOCK_BATCH 256   /* steps per batch of the device-side time step (rgpu_clock_open .. rgpu_clock_close) */

/* ---- the path ------------------------------------------------------------------------------------------- */

/* Ghost fill of one direction: make_boundaries(U, idim) (HydroRunBase.cpp:2276-2316), incl. make_jet.
 * Faces whose bc is RGPU_BC_COPY or RGPU_BC_SHEARINGBOX are left untouched. */
int rgpu_make_boundaries(rgpu_ctx* c, int parity, int idim);

/* Shearing-box remap of the x ghosts: MHDRunGodunov::make_boundaries_shear (MHDRunGodunov.cpp:3539-3759).
 * totalTime is the time at the START of the step that produced U[parity], dt its time step
 * (the remap uses totalTime+dt, :3554). */
int rgpu_make_boundaries_shear(rgpu_ctx* c, int parity, double totalTime, double dt);

/* make_all_boundaries (X,Y,Z; HydroRunBase.cpp:2333-2342) or, when shearingBoxEnabled and 3D,
 * make_all_boundaries_shear (Y, shear, Z, Y; MHDRunGodunov.cpp:3779-3793). */
int rgpu_make_all_boundaries(rgpu_ctx* c, int parity, double totalTime, double dt);

/* max over the interior of the inverse time step: the invDt of compute_dt (HydroRunBase.cpp:372-426) /
 * compute_dt_mhd (MHDRunBase.cpp:140-250) BEFORE "cfl / invDt"; with slabs the caller max-reduces it. */
int rgpu_compute_inv_dt(rgpu_ctx* c, int parity, double* invDt);

/* Tell the context that the caller changed U[0] / U[1] behind its back (arrays adopted with rgpu_create_external and
 * written by the host program; the reference has no counterpart: its compute_dt always rescans).  Where nothing modifies the
 * new state between the step and the next compute_dt, the step's last kernel carries the CFL scan and rgpu_compute_dt only
 * reads the result back; after this call the next rgpu_compute_dt scans the arrays again.  The ghost-fill entry points above
 * do it themselves when they can change what the scan reads (any face that is not periodic / copy / shearing, the jet). */
int rgpu_invalidate_dt(rgpu_ctx* c);

/* == compute_dt[_mhd](useU): cfl / invDt.  Returns NaN on error (see rgpu_last_error). */
double rgpu_compute_dt(rgpu_ctx* c, int useU);

/* == godunov_unsplit(nStep, dt) (MHDRunGodunov.cpp:572-617, HydroRunGodunov.cpp:419-441): reads U[nStep%2],
 * writes U[(nStep+1)%2].  Plain path fills the ghosts of the INPUT at entry; the rotating path (Omega0>0) fills
 * the ghosts of the OUTPUT at exit (MHDRunGodunov.cpp:2031-3440).  totalTime is needed by the shear remaps
 * (:3213, :3554).  With RGPU_BC_COPY z faces use the three calls below instead. */
int rgpu_godunov_unsplit(rgpu_ctx* c, int nStep, double dt, double totalTime);

/* The same step cut at the points where a z-slab driver exchanges ghost planes:
 *   plain   : pre (X,Y fill of input) -> [exchange z ghosts of input]  -> core -> post (nothing)
 *   rotating: pre (nothing)           -> core -> post_a (Y fill, shear remap of output)
 *                                     -> [exchange z ghosts of output] -> post_b (Y fill of output) */
int rgpu_step_pre   (rgpu_ctx* c, int nStep, double dt, double totalTime);
int rgpu_step_core  (rgpu_ctx* c, int nStep, double dt, double totalTime);
int rgpu_step_post_a(rgpu_ctx* c, int nStep, double dt, double totalTime);
int rgpu_step_post_b(rgpu_ctx* c, int nStep, double dt, double totalTime);

/* Plane-ranged pieces (3D contexts) for a slab driver that hides the halo exchange behind the update of the planes
 * nobody else needs ("boundary planes first"): with every ghost of the INPUT valid at entry,
 *   core_planes [0,2gw) and [nz,ksize)  -> fill_planes [gw,2gw) and [nz,nz+gw) -> start exchange of the OUTPUT
 *   core_planes [2gw,nz)                -> fill_planes [2gw,nz)                -> wait -> physical z faces
 * k_lo/k_hi are array plane indices (ghosts included), half open.
 * rgpu_step_core_planes completes the update of planes [k_lo,k_hi) of U[(nStep+1)%2] (every intermediate stage is
 * run on exactly the planes that update needs); rgpu_step_core == planes [0,ksize).
 * rgpu_step_fill_planes applies the in-plane part of the ghost fill to planes [k_lo,k_hi) of the OUTPUT state:
 * X,Y faces (HydroRunBase.cpp:2333-2342) or, shearing box, Y + shear remap + Y (MHDRunGodunov.cpp:3779-3793 with
 * the z copy commuted out: all three act within one z plane). */
int rgpu_step_core_planes(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi);
/* The same piece in two halves, for solvers whose update is a kernel of its own (3D MHD): RGPU_CORE_FLUXES computes the
 * face fluxes and edge EMFs the update of planes [k_lo,k_hi) needs (one z-marching launch -- over the whole slab it costs one
 * pipeline fill instead of one per plane range), RGPU_CORE_UPDATE then completes planes [k_lo,k_hi) -- any sub-ranges of a
 * FLUXES range, in any order.  For every other solver FLUXES does nothing and UPDATE is rgpu_step_core_planes, so the
 * schedule  FLUXES [0,ksize) ; UPDATE boundary ranges ; exchange || UPDATE inner range  is valid for all of them. */
enum { RGPU_CORE_FLUXES = 1, RGPU_CORE_UPDATE = 2, RGPU_CORE_SCAN = 4 };
/* | RGPU_CORE_SCAN: the CFL scan of the new state rides in the update kernels of the pieces (no pass over U for the next
 * compute_dt): FLUXES | SCAN resets the context's RGPU_DT_SLOTS device slots, every UPDATE | SCAN accumulates the maxima of the
 * cells it updates, rgpu_inv_dt_fused_commit(ctx, parity of the new state) closes the accumulation and returns the number of
 * slots to all-reduce (rgpu_inv_dt_device_slot) before rgpu_inv_dt_result.  rgpu_inv_dt_fused_active tells right after the
 * FLUXES call whether the step can carry the scan; if not (dissipative stage, forcing, open faces on the rotating path, flat
 * kernels) scan with rgpu_inv_dt_accumulate as before. */
/* 1 when THIS context's configuration lets its update pieces carry the scan (depends on its boundary types: the end slabs of
 * a run may differ from the inner ones).  All ranks must pass the same flag combination and all-reduce the same number of
 * slots: the slab driver takes the minimum over the ranks once and drops RGPU_CORE_SCAN everywhere if any rank says 0. */
int rgpu_inv_dt_fusable(rgpu_ctx* c);
int rgpu_inv_dt_fused_active(rgpu_ctx* c, int parity);   /* after FLUXES | SCAN: 1 when this step's pieces carry the scan */
int rgpu_inv_dt_fused_commit(rgpu_ctx* c, int parity);
int rgpu_step_core_planes_split(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int what);
/* The dissipative stage of the step ([hydro] nu / [MHD] eta > 0; no-op otherwise) on U[(nStep+1)%2], WITHOUT the ghost
 * fill that precedes it in rgpu_godunov_unsplit: a slab driver calls it between rgpu_step_core and rgpu_step_post_a after
 * it has filled the ghosts of the output itself (rgpu_make_boundaries / _shear + its z exchange), as the reference's MPI
 * classes do with make_all_boundaries(h_UNew) (mhd_godunov_unsplit_cpu_v3.cpp:662-668). */
int rgpu_step_dissipative(rgpu_ctx* c, int nStep, double dt, double totalTime);
int rgpu_step_fill_planes(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi);
/* The same two pieces for TWO disjoint plane ranges at once -- the two boundary ranges of a slab, [0,2gw) + [nz,ksize) and
 * [gw,2gw) + [nz,nz+gw) -- so that what lies between the end of the flux sweep and the start of the halo exchange is a handful of
 * launches: the 3D MHD update kernel marches both ranges in one launch; the ghost fill of both ranges is one launch whenever the
 * x / y faces are mirror / copy / periodic or the shearing box with periodic y (one thread per ghost cell, every value a function
 * of interior cells of its plane: X, Y -- or Y, shear remap, Y -- need no ordering).  Same doubles as the one-range calls.
 * Either range may be empty.  (3D hydro: the sweep is the whole step -- both ranges in one launch of it.) */
int rgpu_step_core_planes_pair(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int k_lo2, int k_hi2, int what);
int rgpu_step_fill_planes_pair(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int k_lo2, int k_hi2);

/* rgpu_compute_inv_dt in pieces: accumulate the max over the cells of planes [k_lo,k_hi) into the context's device
 * slot (reset != 0 starts a new maximum), asynchronously on the context stream; rgpu_inv_dt_result synchronises
 * and returns it with the seeds of compute_dt[_mhd] applied.  Lets a slab driver scan each plane of the output
 * BEFORE its ghosts are refilled, which is the state the reference's compute_dt sees on the plain path
 * (oneStepIntegration: compute_dt, then godunov_unsplit fills the ghosts; MHDRunGodunov.cpp:4077-4089). */
int rgpu_inv_dt_accumulate(rgpu_ctx* c, int parity, int k_lo, int k_hi, int reset);
int rgpu_inv_dt_result(rgpu_ctx* c, double* invDt);

/* History diagnostics of the MHD runs, reduced on the device instead of copying the state to the host
 * (MHDRunBase::history_mri, MHDRunBase.cpp:3476-3619; history_default, :3311-3407 is its mass / divB subset).
 * rgpu_history_mri: out[8] = mass, maxwell stress, reynolds stress, magnetic pressure, mean Bx, By, Bz, sum of divB,
 * normalised like the reference's history file (single domain).  Sums are accumulated in a fixed order that is not the
 * reference's loop order: they agree with it to round-off.
 * Slab runs combine the two lower-level calls: rgpu_history_columns gives cols[9][isize] = sums over the interior y,z
 * extent of this domain of {rho, mx/rho, my/rho (every i), magp terms, maxwell term, Bx, By, Bz, divB (interior i)};
 * after an all-reduce, mean_vx/vy[i] = cols[1..2][i] / (ny * nz_global) feed rgpu_history_reynolds, which returns
 * cols[isize] = sum_jk rho * dTau * (vx - mean_vx)(vy - mean_vy). */
int rgpu_history_columns(rgpu_ctx* c, int parity, double* cols);
int rgpu_history_reynolds(rgpu_ctx* c, int parity, const double* mean_vx, const double* mean_vy, double dTau, double* cols);
int rgpu_history_mri(rgpu_ctx* c, int parity, double* out);
/* MHDRunBase::history_turbulence (MHDRunBase.cpp:3626-3810; problems "turbulence" and "turbulence-Ornstein-Uhlenbeck", 3D),
 * reduced on the device: out[18] = the columns of the reference's history file after totalTime and dt --
 * mass divB eKin eMag helicity mean_rho mean_B mean_Bx mean_By mean_Bz mean_rhovx mean_rhovy mean_rhovz Ma_s Ma_alfven
 * coef_x coef_y coef_z (the three high-k DFT amplitudes of Bx it monitors).  Single-domain contexts; fixed summation order,
 * device cos / sin: agreement with the reference to round-off. */
int rgpu_history_turbulence(rgpu_ctx* c, int parity, double* out);
/* The 18 raw sums behind rgpu_history_turbulence over the interior cells of this context -- also a slab's (its own planes):
 * the z-slab driver adds them up across the ranks (rgpu_comm_history_turbulence, the reference's history_mhd_turbulence of the
 * MPI classes, HydroRunBaseMpi.cpp:11346-11530). */
int rgpu_history_turbulence_sums(rgpu_ctx* c, int parity, double* sums18);
/* One cell of the state, out[nbVar] = U(i,j,k,:) with ghost-inclusive local indices: what history_inertial_wave
 * (MHDRunBase.cpp:3414-3469) probes -- U(ghostWidth + nx/2, ghostWidth) in 2D, U(ghostWidth + nx/2, 1, ghostWidth) in 3D --
 * without copying the whole array back (the reference calls copyGpuToCpu first). */
int rgpu_read_cell(rgpu_ctx* c, int parity, int i, int j, int k, double* out);

/* Reproducibility fingerprint of U[parity]: the sum, modulo 2^64, of the 64-bit patterns of every variable of every INTERIOR cell of
 * this context (a slab: its own planes).  Integer addition is associative: the value does not depend on how the box was cut into
 * slabs or tiles, so the sum of the slabs' checksums (mod 2^64) of an N-rank run equals the single-device run's whenever the states
 * are equal bit for bit -- which the z-slab driver promises (rgpu_comm.h).  bench.py prints it for every N.  (The reference compares
 * runs through its output files; this is the same check without the 8.9 GB copy, HydroRunBase.cpp:7217-7229 copyGpuToCpu.) */
int rgpu_state_checksum(rgpu_ctx* c, int parity, unsigned long long* out);

/* == oneStepIntegration(nStep, t, dt) (MHDRunGodunov.cpp:4077-4089) for a single device */
int rgpu_one_step_integration(rgpu_ctx* c, int* nStep, double* t, double* dt);

/* The body of the reference's time loop, "while (totalTime < tEnd && nStep < nStepmax) oneStepIntegration(nStep, totalTime, dt)"
 * (MHDRunGodunov.cpp:3921-3990, HydroRunGodunov.cpp:3960), for up to nsteps steps: returns the number of steps done (< nsteps only when
 * totalTime reached tEnd; pass HUGE_VAL for "no end") or a negative RGPU_E* code; *nStep, *t, *dt advance exactly as nsteps calls of
 * rgpu_one_step_integration would advance them -- same states, same dt sequence, bit for bit.  What it adds: where a step is ONE fused
 * kernel that leaves the CFL maxima and the ghost cells of its output on the device (2D hydro in a box of periodic, reflecting or
 * outflow faces; 2D MHD in an all-periodic box -- with a Neumann face its kernel writes no ghost images and the plain loop runs; no
 * gravity, no rotating frame) the time step itself stays on the device (csrc/hip/step_clock.h: dt = cfl / max 1/dt, the
 * loop condition and t += dt evaluated by a one-workgroup kernel between two steps) and a batch of steps is queued without a host round
 * trip -- at the shipped 2D sizes that round trip costs as much as a third of the step.  Since round 5 the 3D steps do the same (hydro,
 * plain / rotating / shearing-box MHD through the z-marching sweeps: rgpu_clock_capable).  Every other configuration runs the plain loop. */
int rgpu_run_steps(rgpu_ctx* c, int nsteps, double tEnd, int* nStep, double* t, double* dt);
/* ... the same, and dt_log[n] = the time step of the n-th step done (the "dt=" column of the reference's log, MHDRunGodunov.cpp:3958);
 * dt_log holds nsteps doubles or is NULL.  If a launch fails after some steps of a batch were queued, *nStep, *t, *dt (and dt_log)
 * are advanced for the steps that did run before the error is returned. */
int rgpu_run_steps_log(rgpu_ctx* c, int nsteps, double tEnd, int* nStep, double* t, double* dt, double* dt_log);
/* 1 when the next step of rgpu_run_steps on the state U[parity] would take its time step from the device (see above), else 0:
 * lets a caller (and the tests) tell which loop runs. */
int rgpu_device_time_step_ready(rgpu_ctx* c, int parity);

/* The device-side time step as pieces, for a driver that queues whole batches of steps itself (the z-slab driver: all-reduce of the
 * 1/dt slots in place -> clock -> step pieces -> halo exchange, no host turn between steps; rgpu_run_steps is built on the same three).
 *   rgpu_clock_capable  1 when every kernel of this context's step that depends on dt or t can read them from a device record
 *                       (csrc/step_clock_rec.h): 2D fused steps; 3D hydro and MHD through the z-marching sweeps, incl. the rotating frame
 *                       and the shearing box (periodic y); no gravity, dissipative stage, forcing, phase timers.
 *   rgpu_clock_open     starts a batch at time t0; steps stop being executed once t >= tEnd (HUGE_VAL: never).
 *   rgpu_clock_tick     queues the clock kernel of the NEXT step: folds the RGPU_DT_SLOTS device slots (which must hold the CFL maxima of
 *                       the step's input state -- all-reduced across slabs by the caller), zeroes them for the step's own scan, forms
 *                       dt = cfl / max(1/dt), dt/dx.., the rotating-frame coefficients, the shearing-box offsets, t += dt with the host's
 *                       expressions (the same doubles).  Until the next tick / close every rgpu_step_* piece of this context reads that
 *                       record on the device; its dt / totalTime arguments are ignored.  A step whose record says "stop" (t >= tEnd, or
 *                       1/dt not finite) and every step after it are no-ops that leave state, ghost cells and slots untouched.
 *   rgpu_clock_close    reads the records back (the one synchronisation of the batch): *ran = steps that ran, *t += their dt in order,
 *                       *dt_last, dt_log[0..ran) (may be NULL), *stop = 0 or why the batch stopped (1: tEnd, 2: dt is NaN, 3: 1/dt not
 *                       finite).  nStep0 = step number of the first step of the batch.  At most RGPU_CLOCK_BATCH ticks per batch. */
int rgpu_clock_capable(rgpu_ctx* c);
int rgpu_clock_open(rgpu_ctx* c, double t0, double tEnd);
int rgpu_clock_tick(rgpu_ctx* c);
int rgpu_clock_close(rgpu_ctx* c, int nStep0, int* ran, double* t, double* dt_last, double* dt_log, int* stop);
/* 1 when the host already KNOWS that the record of the last tick says "stop" -- never on the device backend (the record is formed
 * asynchronously; the batch's no-op steps are simply queued), always up to date on a synchronous backend (the test-only host emulation),
 * where a driver can end its batch at once.  The answer is the same on every slab of a run. */
int rgpu_clock_stopped(rgpu_ctx* c);
/* The same answer on ANY backend, at the price of a synchronisation: waits for the record of the last tick and returns its stop flag
 * (0 = the step runs, 1 / 2 / 3 as in rgpu_clock_close; < 0: error).  The slab driver checks the first step of every batch this way:
 * a rank that failed in an unbatched step has told the others through ONE poisoned all-reduce and left the loop. */
int rgpu_clock_check(rgpu_ctx* c);

/* Self-test of the device arithmetic the parity contract rests on: for n operand pairs computes on the device
 *   quot[i]  = rg_div(num[i], rg_recip(den[i]))   the shared-reciprocal division of csrc/hip/rg_backend.h
 *   quot2[i] = num[i] / den[i]                    the compiler's IEEE division
 *   root[i]  = rg_sqrt(num[i]),  root2[i] = sqrt(num[i])
 * (host arrays).  Inside the documented operand range all four equal the correctly rounded results bit for bit. */
int rgpu_selftest_arith(int n, const double* num, const double* den, double* quot, double* quot2, double* root, double* root2);

/* Self-test of the one place where the exact library departs from the reference's instruction sequence on data-dependent grounds:
 * the Alfven speeds of the 2D HLLD edge solver (mag_riemann2d_hlld, riemann_mhd.h:727-738) are formed for the WINNER of each group
 * of four |b| / sqrt(rho) candidates, picked on the operands (csrc/dev_numerics.h: alfven_pick / alfven_duel); a lane whose ordering
 * is closer than the error bound sends its wave down the reference's sequence.  For n samples -- SoA, states36[q * n + i]: the four
 * corner states LL, RL, LR, RR of an edge (r p u v w a b c each, q = 0..31) and their electric fields ELL, ERL, ELR, ERR (q = 32..35)
 * -- the device evaluates the solver twice, e_select[i] through the selection and e_reference[i] through the reference's sequence,
 * with the knobs of *p (gamma0, cIso, smallc, ...); route[i] = 1 when sample i's wave (64 consecutive samples) took the reference's
 * sequence anyway.  e_select must equal e_reference bit for bit (tests/test_gpu_parity.py: >= 1e7 random and adversarial states).
 * The contracted library has no selection: both outputs come from the same sequence and route is 1. */
int rgpu_selftest_alfven(const rgpu_params* p, int n, const double* states36, double* e_select, double* e_reference, int* route);

/* ---- Environment and options ------------------------------------------------------------------------------------------------
 * Environment variables the libraries read (all optional; everything else is an argument of an entry point):
 *   RGPU_TILED=0              librgpu.so: the flat per-cell kernels everywhere instead of the LDS-tiled cooperative ones (a second,
 *                             independently tested implementation of every step; 3-10x slower)
 *   RGPU_COMM_SCHEDULE=1|2    librgpu_comm.so: step schedule of the slab driver where the caller leaves the choice to it (rgpu_comm_set_overlap(-1), rgpu_comm.h)
 *   RGPU_COMM_PACK=0          ... one send / recv per variable and face instead of the packed exchange
 *   RGPU_HALO_PRIO=high|low   ... priority of the halo stream (default: normal)
 *   RGPU_COMM_ONE_STREAM=1    ... halo traffic on the compute stream (no overlap): fallback should two streams on one RCCL
 *                             communicator misbehave on a given node
 *   RGPU_HDF5_LIB=<path>      host layer: the libhdf5 to load (default: the system's)
 *   RGPU_RESTART_FORMAT=...   host layer: what rgpuh_* writes for restarts (run_driver.h)
 * Diagnostic options (process-wide; tests run a configuration both ways through them -- a user needs none):
 *   "spec" (1)          kernels specialised for the solver configuration; 0: the generic instantiations
 *   "ghost_images" (1)  the fused 2D steps write the ghost images of their output; 0: the next step fills the ghost cells
 *   "step_clock" (1)    the time step stays on the device between the steps of rgpu_run_steps; 0: one host turn per step
 *   "xcd_sub" (-1)      sub-band size (cells) of the XCD-aware workgroup order of the flat kernels, read by rgpu_create; 0: linear
 *   "zseg" (0)          planes per z segment of the tiled sweeps; 0: planned per launch
 *   "chunks" (-1)       chunks of the two-stream schedule of the flat 3D MHD kernels, read by rgpu_create; 1: one stream
 * rgpu_set_option returns the previous value (-1: unknown name; the options above are never negative except "as created"). */
int rgpu_set_option(const char* name, int value);
int rgpu_get_option(const char* name);

/* name of the device backend the library was built for ("hip-gfx950") */
const char* rgpu_backend_name(void);

/* floating-point arithmetic of the kernels in this library:
 *   "exact"       librgpu.so       no FMA contraction, correctly rounded division / square root, the reference's operand
 *                                  order: results bit-identical to the reference's CPU path
 *   "contracted"  librgpu_fast.so  the same sources with FMA contraction and ~1-ulp division / square root: results agree
 *                                  with the reference to round-off (relative L2 < 1e-12 on every golden fixture), ~20 %
 *                                  faster on the 3D MHD step.  Same ABI: link one or the other. */
const char* rgpu_arithmetic(void);

/* block until all queued work of this context is complete */
int rgpu_synchronize(rgpu_ctx* c);

/* ---- instrumentation (the reference's DO_TIMING phase timers, MHDRunGodunov.h:382-430) ------------------- */

enum {
  RGPU_T_BOUNDARIES = 0, RGPU_T_PRIM, RGPU_T_ELEC, RGPU_T_TRACE, RGPU_T_FLUX, RGPU_T_EMF, RGPU_T_UPDATE,
  RGPU_T_SHEAR, RGPU_T_DT, RGPU_T_DISSIPATIVE, RGPU_T_SWEEP, RGPU_T_COUNT
};
/* RGPU_T_FLUX is the Riemann phase: face fluxes and edge EMFs are one kernel (RGPU_T_EMF stays 0).
 * RGPU_T_SWEEP is the LDS-tiled, z-marching fused kernel (hydro 3D: the whole step; 3D MHD: trace + Riemann problems).
 * enable!=0 brackets every phase with hipEvents (serialises the stream; off by default) */
int rgpu_enable_timers(rgpu_ctx* c, int enable);
/* accumulated seconds per phase since creation / last reset; n <= RGPU_T_COUNT */
int rgpu_get_timers(rgpu_ctx* c, double* secs, int n);
int rgpu_reset_timers(rgpu_ctx* c);
const char* rgpu_timer_name(int which);

/* name / average duration [ms] / launch count of the dominant kernel of the last timed steps, measured with
 * hipEvents on the context's stream (used by bench.py's roofline object) */
int rgpu_dominant_kernel(rgpu_ctx* c, char* name, int name_len, double* avg_ms, long* launches);

/* ---- host side of the reference interface (C++ lives in csrc/host; these are its C entry points) ---------- */

/* Parse an .ini file exactly like ConfigMap + HydroParameters (float-parsed knobs, case-insensitive keys,
 * ConfigMap.cpp:41-49, INIReader.cpp:94-101) with optional "section.key=value;..." overrides. */
int rgpuh_params_from_ini(const char* ini_path, const char* overrides, rgpu_params* out, char* err, int err_len);

/* the [run] section the reference's start() loop reads: nstepmax, tend, noutput (HydroRunBase.cpp:224-232) */
int rgpuh_run_settings(const char* ini_path, const char* overrides, int* nStepmax, double* tEnd, int* nOutput,
                       char* err, int err_len);

/* Fill hU (rgpu_state_elems doubles, zeroed first) with the initial condition named by [hydro] problem:
 * jet, implode (HydroRunBase.cpp:5282-5350, 5449-5536), Orszag-Tang, Brio-Wu, MRI
 * (MHDRunBase.cpp:1378-1475, 1870-2080, 2677-2758).  For a slab, only planes of this slab are produced
 * (the MRI drand48 stream is skipped ahead so that every slab draws the numbers the single-domain run would). */
int rgpuh_init_condition(const char* ini_path, const char* overrides, const rgpu_params* p, double* hU,
                         char* err, int err_len);

/* Fill hG (3 * cells doubles: x, y, z component, ghost cells included; zeroed first) with the static gravity field the
 * problem's init routine writes into h_gravity: Keplerian-disk (HydroRunBase.cpp:6489-6500, 6575-6597).  Returns 1 when the problem defines a field (params_from_ini then says gravityEnabled = 2), 0 when it does not,
 * a negative RGPU_E* on error. */
int rgpuh_init_gravity(const char* ini_path, const char* overrides, const rgpu_params* p, double* hG,
                       char* err, int err_len);

/* Same for the static driving field of the "turbulence" problem (turbulenceInit.cpp; params_from_ini then says
 * randomForcingEnabled = 1): hF = 3 * cells doubles.  Returns 1 / 0 / negative like rgpuh_init_gravity. */
int rgpuh_init_forcing(const char* ini_path, const char* overrides, const rgpu_params* p, double* hF,
                       char* err, int err_len);

/* Run [run] nstepmax / tend like MHDRunGodunov::start / HydroRunGodunov::start on one GPU; writes .vti outputs
 * when [output] outputVtk=yes.  Returns steps done (>=0) or a negative error. Mcell-updates/s is printed like
 * MHDRunGodunov.cpp:4064-4068. */
int rgpuh_run(const char* ini_path, const char* overrides, double* mcell_per_s, char* err, int err_len);

/* The run loop of rgpuh_run with the stepping delegated -- the seam of the z-slab front end (include/rgpu_comm.h fills the
 * hooks with the slab driver; euler_hip --slabs N).  The run builds the context of slab `slab_rank` of `slab_count` and its
 * initial state (initial condition, or [run] restart from the .h5 of the whole box), calls attach(user, ctx, &hooks), then
 * runs the reference's time loop through the hooks: ghost fill, dt, one step; outputs go to ONE HDF5 file per output step for
 * the whole box (the ranks take turns, hooks.barrier between them) and are identical to the single-domain files; detach(user)
 * at the end.  Returns the number of steps or a negative error code (message in err). */
typedef struct rgpuh_step_hooks {
  void* self;
  int (*make_all_boundaries)(void* self, int parity, double totalTime, double dt);
  int (*compute_dt)(void* self, int useU, double* dt);
  int (*one_step_integration)(void* self, int* nStep, double* totalTime, double* dt);
  int (*barrier)(void* self);
  const char* (*last_error)(void* self);
  int (*history_mri)(void* self, int parity, double* out8);   /* optional (may be 0): rgpu_history_mri over the whole box */
  /* optional (may be 0; then barrier is used): collective over all ranks, returns the NUMBER of ranks that passed
   * local_failed != 0 (or a negative code when the transport itself failed).  The run loop calls it where one rank alone
   * can fail -- file output -- so that every rank throws together instead of one leaving the others in a collective. */
  int (*agree)(void* self, int local_failed);
  /* optional: the 14 numbers after "totalTime dt" of the MPI classes' turbulence history row (rgpu_comm_history_turbulence) */
  int (*history_turbulence)(void* self, int parity, double* out14);
  /* optional: the loop body for a run of quiet steps (rgpu_comm_run_steps; contract of rgpu_run_steps: steps done or a negative code) */
  int (*run_steps)(void* self, int nsteps, double tEnd, int* nStep, double* totalTime, double* dt);
} rgpuh_step_hooks;
typedef int (*rgpuh_attach_fn)(void* user, rgpu_ctx* ctx, rgpuh_step_hooks* hooks);
typedef void (*rgpuh_detach_fn)(void* user);
int rgpuh_run_hooked(const char* ini_path, const char* overrides, int slab_rank, int slab_count, rgpuh_attach_fn attach,
                     rgpuh_detach_fn detach, void* user, double* mcell_per_s, char* err, int err_len);

#ifdef __cplusplus
}
#endif
#endif /* RGPU_H_ */
