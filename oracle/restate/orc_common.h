// orc_common.h -- ORACLE (test infrastructure, NOT product code).
//
// CPU restatement of the reference's euler_cpu time step, used only by tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg as the checker of the HIP path.  Plain single-threaded C++; every loop keeps the
// reference's iteration order and every expression its operand order, so that (built with the same flags as
// oracle/_ref/euler_cpu: g++ -O2, no -march, no -ffast-math, no OpenMP) results are BIT-IDENTICAL to the
// reference binary.  That identity is pinned by tests/test_oracle_golden.py against fixtures produced by the
// reference itself (oracle/gen_golden.py).
//
// The only product file it sees is include/rgpu.h, for the parameter struct the checker and the checked
// share.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rgpu.h"

namespace orc {

enum { ID = 0, IP = 1, IU = 2, IV = 3, IW = 4, IA = 5, IB = 6, IC = 7 };
enum { IX = 0, IY = 1, IZ = 2 };
// EmfIndex, constants.h:191-195 (EMFZ first)
enum { I_EMFZ = 0, I_EMFY = 1, I_EMFX = 2 };

// Geometry + knobs of one run (what the reference keeps in HydroParameters members and the global gParams).
struct Ctx {
  rgpu_params p;
  int nx, ny, nz, gw, isize, jsize, ksize, nvar;
  bool three_d;
  size_t ncell;
  double dx, dy, dz;
  explicit Ctx(const rgpu_params& pp) : p(pp) {
    nx = p.nx; ny = p.ny; nz = p.nz; gw = p.ghostWidth; nvar = p.nbVar;
    three_d = (p.nz_global != 1);
    isize = nx + 2 * gw; jsize = ny + 2 * gw; ksize = three_d ? nz + 2 * gw : 1;
    ncell = (size_t)isize * jsize * ksize;
    dx = p.dx; dy = p.dy; dz = p.dz;
  }
  size_t idx(int i, int j, int k) const { return (size_t)i + (size_t)isize * (j + (size_t)jsize * k); }
  // static gravity at a cell: the uniform vector (gravityEnabled == 1) or the per-cell field h_gravity
  // (gravityEnabled == 2, given with orc_set_gravity_field; zero until then, like the reference's allocation)
  const double* G = gravity_field();
  double grav(int i, int j, int k, int e) const {
    if (p.gravityEnabled == 2) return G ? G[idx(i, j, k) + ncell * e] : 0.0;
    return e == 0 ? p.gravity_x : e == 1 ? p.gravity_y : p.gravity_z;
  }
  static const double*& gravity_field() { static const double* g = 0; return g; }
  // h_randomForcing of the "turbulence" problem (orc_set_forcing_field)
  const double* Frc = forcing_field();
  static const double*& forcing_field() { static const double* f = 0; return f; }
};

// random forcing at the end of a 3D step (HydroRunBase.cpp:1201-1312, 1397-1428)
void random_forcing(const Ctx& c, double* U, double dt);
// Ornstein-Uhlenbeck forcing (Forcing_OrnsteinUhlenbeck.cpp): ou_init at the start of a run (init_forcing), ou_forcing at
// the end of every 3D step (add_forcing_field); the process is a run-long state like the reference's pForcingOrnsteinUhlenbeck
void ou_init(const rgpu_params& p);
void ou_forget();
void ou_forcing(const Ctx& c, double* U, double dt);

// A component-major field with the reference's HostArray index map (Arrays.h:95-98).
struct Field {
  const Ctx* c;
  std::vector<double> own;
  double* d;
  int nv;
  Field() : c(0), d(0), nv(0) {}
  void alloc(const Ctx& ctx, int nvar) { c = &ctx; nv = nvar; own.assign(ctx.ncell * nvar, 0.0); d = own.data(); }
  void wrap(const Ctx& ctx, double* ptr, int nvar) { c = &ctx; nv = nvar; d = ptr; }
  double& operator()(int i, int j, int k, int v) { return d[c->idx(i, j, k) + c->ncell * v]; }
  double operator()(int i, int j, int k, int v) const { return d[c->idx(i, j, k) + c->ncell * v]; }
  double& operator()(int i, int j, int v) { return d[c->idx(i, j, 0) + c->ncell * v]; }
  double operator()(int i, int j, int v) const { return d[c->idx(i, j, 0) + c->ncell * v]; }
};

// entry points of the individual translation units
void make_boundaries(const Ctx& c, double* U, int idim);                               // orc_boundaries.cpp
void make_boundaries_shear(const Ctx& c, double* U, double totalTime, double dt);      // orc_boundaries.cpp
void make_all_boundaries(const Ctx& c, double* U, double totalTime, double dt);        // orc_boundaries.cpp
double compute_inv_dt(const Ctx& c, const double* U);                                  // orc_dt.cpp
void hydro_step(const Ctx& c, double* Uold, double* Unew, double dt);                  // orc_hydro.cpp
void mhd_step_2d(const Ctx& c, double* Uold, double* Unew, double dt);                 // orc_mhd2d.cpp
void mhd_step_3d(const Ctx& c, double* Uold, double* Unew, double dt, double totalTime);  // orc_mhd3d.cpp
void dissipative_stage(const Ctx& c, double* U, double dt, double totalTime);          // orc_dissipative.cpp
// threaded variants for the all-cores CPU baseline (z-slab threading, gather update; bit-identical to the sequential ones)
struct MtWork {   // the step's intermediate arrays, kept across the steps of a run
  double* buf; size_t doubles;
  MtWork(const Ctx& c, int nthreads);
  ~MtWork();
 private:
  MtWork(const MtWork&); MtWork& operator=(const MtWork&);
};
// thread placement of the threaded variants (orc_mhd3d.cpp): see set_thread_placement
int set_thread_placement(int mode);           // 0 unpinned, 1 pinned over all allowed CPUs (NUMA-major order), 2 pinned inside NUMA node 0; returns CPUs in the set
void pin_slab_thread(int t, int nthreads);    // called by thread t of n at its start
template <class F>
void slabs(int k0, int k1, int nthreads, F fn) {   // fn(ka, kb) on [k0, k1) cut into nthreads contiguous pieces, one std::thread each
  const int n = k1 - k0;
  if (nthreads <= 1 || n <= 1) { fn(k0, k1); return; }
  if (nthreads > n) nthreads = n;
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) {
    const int a = k0 + (int)((long long)n * t / nthreads), b = k0 + (int)((long long)n * (t + 1) / nthreads);
    th.emplace_back([=]() { pin_slab_thread(t, nthreads); fn(a, b); });
  }
  for (auto& x : th) x.join();
}
void mhd_step_3d_mt(const Ctx& c, MtWork& w, double* Uold, double* Unew, double dt, double totalTime, int nthreads);   // orc_mhd3d.cpp
double compute_inv_dt_mhd3d_mt(const Ctx& c, const double* U, int nthreads);                                // orc_mhd3d.cpp

}  // namespace orc
