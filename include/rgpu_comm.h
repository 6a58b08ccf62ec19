/*
 * rgpu_comm.h -- C ABI of the z-slab driver: one process per GPU, RCCL over xGMI (librgpu_comm.so, next to librgpu.so).
 *
 * Replaces, for the one decomposition the path needs (mx = my = 1, mz = nranks), the reference's
 * MPI layer around the step:
 *   cartesian topology + neighbour ranks        HydroMpiParameters.cpp:44-80, 196-201   -> rgpu_comm_create
 *   make_boundaries ZDIR: host-staged border buffers + MPI_Sendrecv (blocking)
 *                                               HydroRunBaseMpi.cpp:3529-3661            -> rgpu_comm_exchange_z_start / _wait
 *   make_all_boundaries[_shear] of the Mpi classes (X,Y,Z | X-int,Y,Z,shear,Z,Y)
 *                                               MHDRunGodunovMpi.cpp:4266-4307           -> rgpu_comm_make_all_boundaries
 *   compute_dt + MPI allReduce(MIN)             HydroRunBaseMpi.cpp:509-513, 696-700     -> rgpu_comm_compute_dt
 *   godunov_unsplit / oneStepIntegration of MHDRunGodunovMpi / HydroRunGodunovMpi        -> rgpu_comm_godunov_unsplit,
 *                                                                                           rgpu_comm_one_step_integration
 * k is the slowest index, so the ghostWidth planes of one variable are one contiguous chunk.  The chunks for one peer are
 * gathered into a device staging buffer by one small kernel, travel as ONE ncclSend / ncclRecv per peer (one grouped launch per
 * exchange) and are scattered by a second kernel (RGPU_COMM_PACK=0: sent from / received into the state arrays in place, one
 * operation per chunk -- which RCCL runs as eight launches), no host staging, on
 * a dedicated halo stream, ordered against the context's compute stream by events only.  The 1/dt maximum is all-reduced
 * in place in the context's device slot (ncclMax on the compute stream) and read back once per step.
 *
 * Step schedule (overlap, the default): update the planes the neighbours read -> finish their x / y ghosts -> start the
 * exchange of the output state -> update the inner planes while it is in flight -> wait (device side) -> physical z faces.
 * State and dt sequence are bit-identical to the single-domain run for every configuration whose step has no global sum
 * (tests/test_comm_driver.py, world sizes 1, 2, 3: hydro, plain and rotating MHD, shearing box, stratified box, dissipative
 * stage, Ornstein-Uhlenbeck forcing).  NOT bit-identical, by construction: the static random forcing of the "turbulence"
 * problem -- its normalisation is a sum over the whole box, formed here as an ncclSum of per-slab partial sums, i.e. in
 * another association order than the single-domain column sums, so `norm` and with it the state agree to round-off (and may
 * vary with the number of ranks; the tests hold relative L2 < 1e-12) -- and every history column, which is printed to six
 * digits anyway.  Ghost cells of the plain path: the overlapped schedule has refilled the x / y ghosts of the new state when a
 * step returns, the single-domain step leaves them for the next step's fill -- ghost-inclusive HDF5 files agree on the
 * interior and on the ghost cells the reference itself defines (rotating path: all of them).
 *
 * Bootstrap: rank 0 calls rgpu_comm_unique_id and hands the 128 bytes to the other ranks by any out-of-band channel
 * (a file, an environment variable, torch.distributed's store: the library does not care).
 * Error model of rgpu.h: 0 or a negative RGPU_E* code, message in rgpu_comm_last_error.
 */
#ifndef RGPU_COMM_H_
#define RGPU_COMM_H_

#include "rgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RGPU_COMM_ID_BYTES 128

typedef struct rgpu_comm rgpu_comm;

/* == ncclGetUniqueId; call on ONE rank */
int rgpu_comm_unique_id(char id[RGPU_COMM_ID_BYTES]);

/* Binds a context created with slab_rank = rank, slab_count = nranks (rgpuh_params_from_ini with slab.rank / slab.count,
 * z faces towards neighbour slabs = RGPU_BC_COPY) to a communicator of nranks processes.  Collective over all ranks.
 * The context must outlive the communicator.  nranks = 1 is allowed: without slab interfaces nothing is exchanged; with the
 * measurement key [run] slabSelfRing = yes (periodic z only) the rank is its own z neighbour and sends its planes to itself. */
int rgpu_comm_create(rgpu_ctx* ctx, int rank, int nranks, const char id[RGPU_COMM_ID_BYTES], rgpu_comm** out);
void rgpu_comm_destroy(rgpu_comm* cm);
const char* rgpu_comm_last_error(rgpu_comm* cm);

/* Halo exchange of the z ghost planes of U[parity] that belong to a neighbour slab.  _start queues it on the halo stream
 * behind everything queued on the context's stream so far and returns; _wait makes the context's stream wait for it
 * (device-side dependency, the host does not block). */
int rgpu_comm_exchange_z_start(rgpu_comm* cm, int parity);
int rgpu_comm_exchange_z_wait(rgpu_comm* cm);

/* Ghost fill of all faces of U[parity] across the slabs, in the reference's order (plain: X, Y, Z; shearing box:
 * Y, shear remap, Z, Y).  Used once for the initial state; the step keeps the ghosts valid afterwards. */
int rgpu_comm_make_all_boundaries(rgpu_comm* cm, int parity, double totalTime, double dt);

/* dt = cfl / max over all slabs of the inverse time step of U[useU] */
int rgpu_comm_compute_dt(rgpu_comm* cm, int useU, double* dt);

/* One unsplit step U[nStep%2] -> U[(nStep+1)%2] over all slabs, halo exchange included (all ghosts of the output valid
 * on return, in stream order). */
int rgpu_comm_godunov_unsplit(rgpu_comm* cm, int nStep, double dt, double totalTime);

/* == oneStepIntegration(nStep, t, dt) of the Mpi run classes */
int rgpu_comm_one_step_integration(rgpu_comm* cm, int* nStep, double* t, double* dt);

/* The body of the reference's time loop for up to nsteps steps (rgpu.h: rgpu_run_steps_log, same contract: returns the steps done or a
 * negative code; *nStep, *t, *dt and dt_log advance as nsteps calls of rgpu_comm_one_step_integration would advance them -- same states,
 * same dt sequence on every rank).  What it adds: the time step stays on the device between steps (rgpu.h: rgpu_clock_*) -- per step the
 * 1/dt slots are all-reduced in place, one small kernel forms dt / t / the shearing-box offsets / "t < tEnd" into a record, the step
 * pieces and the halo exchange read it there -- and the host reads the records of a whole batch once (rounds 1-4: one read-back and
 * synchronisation per step, 0.13 ms of a 5.6 ms step at 8 slabs).  Steps that cannot (the first of a run, the serial schedule, gravity,
 * dissipative stage, forcing) are plain rgpu_comm_one_step_integration calls inside the same loop.  Collective over all ranks. */
int rgpu_comm_run_steps(rgpu_comm* cm, int nsteps, double tEnd, int* nStep, double* t, double* dt, double* dt_log);

/* how many steps of this communicator took their time step from the device record (rgpu_comm_run_steps) -- the others went through
 * the host loop; a launcher reports it next to its timing (bench.py: config.time_loop) */
long long rgpu_comm_clocked_steps(rgpu_comm* cm);

/* MHDRunBase::history_mri / history_default over the whole box (MHDRunBase.cpp:3476-3619; the MPI classes reduce on rank 0):
 * out[8] as rgpu_history_mri -- mass, maxwell, reynolds, magnetic pressure, mean Bx, By, Bz, sum of divB.  Per-slab column
 * sums on the device (rgpu_history_columns), SUM all-reduce of the isize-long columns (the y-z means need the global sums
 * before the Reynolds stress can be formed, rgpu_history_reynolds), the same value on every rank. */
int rgpu_comm_history_mri(rgpu_comm* cm, int parity, double* out);

/* HydroRunBaseMpi::history_mhd_turbulence (HydroRunBaseMpi.cpp:11346-11530): out[14] = the columns after "totalTime dt" of its
 * row -- mass, divB, eKin, eMag, helicity, mean_B, mean_Bx, mean_By, mean_Bz, mean_rhovx, mean_rhovy, mean_rhovz, Ma_s, Ma_alfven.
 * Reproduced as the reference computes them: mass, eKin, eMag, mean_v2 and the mean field are summed over the ranks; mean_B is
 * the SUM of the ranks' |mean B| (not the norm of the sum); divB, helicity and mean_rhov are the values of THIS rank alone (the
 * reference reduces them and then prints rank 0's local variables).  Collective; rank 0's result is the file's row. */
int rgpu_comm_history_turbulence(rgpu_comm* cm, int parity, double* out14);

/* Bytes this rank SENDS per halo exchange (both faces, all variables; 0: no face of this slab is a slab interface).  A launcher
 * or test uses it to prove that planes really travel: nranks = 1 exchanges only when the context was made with the measurement
 * key [run] slabSelfRing (periodic z faces turned into slab interfaces of a ring of one), otherwise its z ghosts are filled locally. */
long long rgpu_comm_halo_bytes(rgpu_comm* cm);

/* Duration [ms] of the last halo exchange on the halo stream -- from the moment the compute stream released the planes to the last
 * plane received -- or a negative value if there was none; waits for that exchange.  With rgpu_comm_halo_bytes it gives the rate the
 * links delivered; next to the step time it tells how much of the exchange the schedule hid.  A diagnostic for multi-GPU runs. */
double rgpu_comm_last_exchange_ms(rgpu_comm* cm);

/* Step schedule.  0: serial (exchange between the step pieces).  1: overlapped -- fluxes of the whole slab, update of the boundary
 * planes, exchange behind the update of the inner planes.  2: boundary-first (3D MHD; other solvers: same as 1) -- fluxes and
 * update of the boundary planes first (two short launches of the z-marching sweep), exchange behind the sweep AND the update of
 * the inner planes; costs two extra pipeline fills of the sweep, hides a link time up to the whole inner step (for thin slabs on slow
 * links).  -1 (default): what RGPU_COMM_SCHEDULE=1|2 in the environment says, else by the thickness of the slab -- 2 for 3D MHD slabs
 * of up to 96 planes (N = 8 at 512^3: the inner update alone is shorter than an exchange over xGMI), 1 otherwise.  Every schedule gives
 * the same doubles.  rgpu_comm_schedule: the schedule the next step will run under (0 / 1 / 2). */
int rgpu_comm_set_overlap(rgpu_comm* cm, int overlap);
int rgpu_comm_schedule(rgpu_comm* cm);

/* hipSetDevice for launchers without a HIP binding of their own: call before rgpu_create / rgpu_comm_create */
int rgpu_comm_set_device(int device);

/* What the transport itself reports about the communicator -- RCCL: ncclCommCount, ncclCommUserRank, ncclCommCuDevice and
 * hipDeviceGetPCIBusId of that device -- as opposed to what the caller passed to rgpu_comm_create.  One rank drives one
 * device (the reference: HydroMpiParameters.cpp:196-201, cudaSetDevice(rank % deviceCount)); a launcher proves its binding
 * with these (bench.py prints them per rank, euler_hip --slabs logs them).  Any pointer may be NULL. */
int rgpu_comm_info(rgpu_comm* cm, int* transport_ranks, int* transport_rank, int* device, char* pci_bus_id, int pci_len);

/* version of the RCCL library the communicator runs on (ncclGetVersion: 10000 major + 100 minor + patch); 0 if unknown.  Logged by
 * bench.py and euler_hip --slabs with the binding above: the first thing to look at when a multi-GPU run misbehaves. */
int rgpu_comm_rccl_version(rgpu_comm* cm);

/* name of the transport the library was built with ("rccl") */
const char* rgpu_comm_transport_name(void);

/* euler_hip --slabs: run [run] nstepmax / tend of an .ini on rank `rank` of `nranks`; this process drives HIP device
 * `device` (-1: the current one).  id as above.  Returns the steps done or a negative error; *mcell_per_s = whole-box
 * cell updates per second.  This is the single-GPU run loop (rgpuh_run_hooked, rgpu.h) stepping through this driver: each rank
 * builds -- or, [run] restart, reads from the .h5 of the whole box -- its own slab; [output] outputHdf5 writes ONE file per
 * output step for the whole box (the ranks take turns and agree on the outcome of every turn: a write error on one rank ends
 * the run on all of them), equal to the single-domain file on the interior (see the top of this header), plus the .xmf index;
 * [output] outputVtk writes one .vti per rank and the .pvti index (HydroRunBaseMpi::outputVtk, HydroRunBaseMpi.cpp:4167-4790);
 * the history file is written by rank 0 from all-reduced sums in the formats of the MPI classes (history_mhd_mri,
 * history_mhd_turbulence, history_mhd_default; HydroRunBaseMpi.cpp:10667-11530).  (Xsmurf and NRRD: single-domain runs.) */
int rgpuh_run_slabs(const char* ini_path, const char* overrides, int rank, int nranks, int device,
                    const char id[RGPU_COMM_ID_BYTES], double* mcell_per_s, char* err, int err_len);

#ifdef __cplusplus
}
#endif
#endif /* RGPU_COMM_H_ */
