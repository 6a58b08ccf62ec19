// tiled_mhd.h (HIP / gfx950 only) -- trace + Riemann problems of the 3D MHD unsplit step as ONE cooperative,
// LDS-tiled, z-marching kernel: the 38-double compact traced state T never reaches HBM.
//
// Flat pipeline (kernels_mhd3d.h): K_mhd_trace3d writes T (304 B/cell), K_mhd_flux3d reads it back with its 2x2x2
// neighbourhood -- half of the step's HBM traffic.  Here a 512-thread workgroup (8 waves, 2 per SIMD: the edge solvers
// need ~250 VGPRs) owns a tile of OX x OY = 16 x 8 cells and marches along z:
//
//   trace(kk)    threads 0..152 (the tile + one low-side halo row / column: cell m needs T of m-1 in x, y, z) compute
//                mhd_trace3d_at and put the 38 components into LDS buffer kk & 1            (2 x 46.5 KB of LDS)
//   barrier
//   Riemann(kk)  the six Riemann problems of a cell are dealt to FOUR threads: wave pair 0 solves the x-edge EMF
//                (2D HLLD), pair 1 the y-edge EMF, pair 2 the z-edge EMF, pair 3 the three face fluxes (HLLD) -- each
//                pair covers the 128 cells of the tile, all reading T(kk-1), T(kk) from LDS; F and emf go to HBM
//                with nontemporal stores
//   barrier      (buffer (kk-1) & 1 is overwritten by trace(kk+1))
//
// The task split is what lets one workgroup per CU (LDS-limited) keep all four SIMDs busy with two waves each: a
// one-thread-per-cell kernel would run two waves per CU.  Costs per cell: EMF ~1200 VALU instructions each, the three
// HLLD fluxes ~1050 together, trace ~914 (+20 % for the halo).
// (Reference idiom: the shared-memory staging of trace_v4 / flux_update_hydro_v4, godunov_unsplit_mhd.cuh:3260, 4595.)
//
// Arithmetic: the same device functions as the flat kernels (mhd_trace3d_at, mhd_flux3d_at), instantiated with an LDS
// accessor instead of the global-array one -- same expressions, same operand order, same bits.
#pragma once
#include "tiled_hydro.h"

namespace rgpu_tiled {

constexpr int MH_OX = 16, MH_OY = 8;              // cells whose Riemann problems a workgroup solves per plane
constexpr int MH_PX = MH_OX + 1, MH_PY = MH_OY + 1;
constexpr int MH_CELLS = MH_PX * MH_PY;           // 153 traced cells per plane (tile + low-side halo)
constexpr int MH_BUF = T_COUNT * MH_CELLS;        // doubles per plane buffer of T
constexpr int MH_QX = MH_OX + 3, MH_QY = MH_OY + 3;
constexpr int MH_QCELLS = MH_QX * MH_QY;          // 209 staged input cells per plane (traced cells +- 1)
constexpr int MH_NQB = 11;                        // staged per cell: 8 primitives + 3 face-field components
constexpr int MH_QBSLOT = MH_NQB * MH_QCELLS;     // doubles per plane slot of Q / B
constexpr int MH_ESLOT = 3 * MH_QCELLS;           // doubles per plane slot of E
constexpr int MH_ITEMS = MH_QBSLOT + MH_ESLOT;    // doubles staged per plane
constexpr int MH_THREADS = 512;
constexpr int MH_NSTAGE = (MH_ITEMS + MH_THREADS - 1) / MH_THREADS;   // 6 loads per thread and plane

struct TLdsRead {
  const double* base; unsigned skoff;   // skoff: from a cell of plane kk-1 to the same cell of plane kk (mod 2^32)
  RG_DEVFN double get(int slot, unsigned m) const { return base[slot * MH_CELLS + m]; }
  RG_DEVFN unsigned stride(int D) const { return (D == XD) ? 1u : (D == YD) ? (unsigned)MH_PX : skoff; }
};
struct TLdsWrite {
  double* cell;
  RG_DEVFN void put(int slot, double v) const { cell[slot * MH_CELLS] = v; }
};
struct TraceInLds {   // trace inputs staged in LDS: planes kk-1, kk, kk+1 of Q / B, planes kk, kk+1 of E
  const double* qb[3]; const double* eb[2];
  RG_DEVFN double q(int v, int dz, unsigned m) const { return qb[dz + 1][v * MH_QCELLS + m]; }
  RG_DEVFN double bf(int comp, int dz, unsigned m) const { return qb[dz + 1][(8 + comp) * MH_QCELLS + m]; }
  RG_DEVFN double e(int comp, int dz, unsigned m) const { return eb[dz][comp * MH_QCELLS + m]; }
  RG_DEVFN unsigned sj() const { return (unsigned)MH_QX; }
};

#ifdef RG_SWEEP_PROF   // experiment builds only (scripts/probe_sweep.py --prof): per-wave cycle accounting of the phases
__device__ unsigned long long rg_prof[8 * 4];
#define RG_PROF_T(x) const long long x = (long long)__builtin_readcyclecounter()
#else
#define RG_PROF_T(x)
#endif

// T accessor of ONE plane buffer: the +z neighbour of a cell is not in it.  stride(ZD) = 0 makes the one component of a
// state that reads the plane above (the face field on the + side) read this plane instead; the caller replaces that
// component when the plane above has been traced (see "carried states" below).
struct TLdsPlane {
  const double* base;
  RG_DEVFN double get(int slot, unsigned m) const { return base[slot * MH_CELLS + m]; }
  RG_DEVFN unsigned stride(int D) const { return (D == XD) ? 1u : (D == YD) ? (unsigned)MH_PX : 0u; }
};

// Riemann problems of one direction d at cell m of plane kk: the edge EMF along d and the flux through the low d face.
// Tk = traced state of plane kk.  States that belong to plane kk-1 (the two upper edge states of the x / y edges, the
// left state of the z face) were built one iteration earlier from T(kk-1) -- except their one component that lives on
// plane kk, which is filled in here -- and are carried in registers (c0, c1): T(kk-1) need not stay in LDS.
// After solving, the states plane kk contributes to iteration kk+1 are built into c0, c1.
template <int DIR>
RG_DEVFN void riemann_dir(const DevParams& g, const TLdsPlane& Tk, unsigned m, double xPos, double* __restrict__ F,
                          double* __restrict__ emf, unsigned idx, Prim8& c0, Prim8& c1, bool solve) {
  const size_t N = g.ncell;
  const unsigned sx = 1u, sj = (unsigned)MH_PX;
  if (DIR == XD) {   // edge along x: t1 = y, t2 = z.  rt = (+,+) from c-y-z, rb = (+,-) from c-y, lt = (-,+) from c-z, lb = (-,-) from c
    if (solve) {
      c0.b = Tk.get(T_CL, m - sj) + Tk.get(T_DCLY, m - sj);        // b2 = CL(m2) + s1 * dCLy(m2), s1 = +1, m2 = the cell above
      c1.b = Tk.get(T_CL, m) + (-1.0) * Tk.get(T_DCLY, m);         // s1 = -1
      const Prim8 rb = edge_state3d<0, +1, -1, false>(g, Tk, m - sj, idx), lb = edge_state3d<0, -1, -1, false>(g, Tk, m, idx);
      RG_STREAM_STORE(&emf[idx + (size_t)EMF_X * N], edge_emf<0>(g, c0, rb, c1, lb, xPos));
      Prim8 L = face_state3d<XD, +1, false>(g, Tk, m - sx, idx), R = face_state3d<XD, -1, false>(g, Tk, m, idx);
      double fl[8];
      mhd_face_flux<XD>(g, L, R, xPos, fl);
      store_flux<XD>(g, F, idx, fl);
    }
    c0 = edge_state3d<0, +1, +1, false>(g, Tk, m - sj, idx);
    c1 = edge_state3d<0, -1, +1, false>(g, Tk, m, idx);
  } else if (DIR == YD) {   // edge along y: t1 = z, t2 = x.  rt = (+,+) from c-z-x, rb = (+,-) from c-z, lt = (-,+) from c-x, lb from c
    if (solve) {
      c0.a = Tk.get(T_CL, m - sx) + Tk.get(T_DCLX, m - sx);        // b1 = CL(m1) + s2 * dCLx(m1), s2 = +1
      c1.a = Tk.get(T_CL, m) + (-1.0) * Tk.get(T_DCLX, m);
      const Prim8 lt = edge_state3d<1, -1, +1, false>(g, Tk, m - sx, idx), lb = edge_state3d<1, -1, -1, false>(g, Tk, m, idx);
      RG_STREAM_STORE(&emf[idx + (size_t)EMF_Y * N], edge_emf<1>(g, c0, c1, lt, lb, xPos));
      Prim8 L = face_state3d<YD, +1, false>(g, Tk, m - sj, idx), R = face_state3d<YD, -1, false>(g, Tk, m, idx);
      double fl[8];
      mhd_face_flux<YD>(g, L, R, xPos, fl);
      store_flux<YD>(g, F, idx, fl);
    }
    c0 = edge_state3d<1, +1, +1, false>(g, Tk, m - sx, idx);
    c1 = edge_state3d<1, +1, -1, false>(g, Tk, m, idx);
  } else {   // edge along z: all four states on plane kk; z face: left state from plane kk-1
    if (solve) {
      const Prim8 rt = edge_state3d<2, +1, +1, false>(g, Tk, m - sx - sj, idx), rb = edge_state3d<2, +1, -1, false>(g, Tk, m - sx, idx);
      const Prim8 lt = edge_state3d<2, -1, +1, false>(g, Tk, m - sj, idx), lb = edge_state3d<2, -1, -1, false>(g, Tk, m, idx);
      RG_STREAM_STORE(&emf[idx + (size_t)EMF_Z * N], edge_emf<2>(g, rt, rb, lt, lb, xPos));
      c0.a = Tk.get(T_CL, m);                                      // bn of the left state: the face it shares with cell m
      Prim8 R = face_state3d<ZD, -1, false>(g, Tk, m, idx);
      double fl[8];
      mhd_face_flux<ZD>(g, c0, R, xPos, fl);
      store_flux<ZD>(g, F, idx, fl);
    }
    c0 = face_state3d<ZD, +1, false>(g, Tk, m, idx);
  }
}

// Wave roles (8 waves; waves w and w + 4 share a SIMD):
//   waves 0,1,2 and 4,5,6   Riemann problems of direction d = w & 3 for the cells [64 * (w >> 2), +64) of the tile:
//                           edge EMF along d (2D HLLD, ~1100 VALU instructions) + flux through the low d face (~470)
//   wave 3                  trace of cells 0..63 of the (tile + halo) plane, then of cells 128..152
//   wave 7                  trace of cells 64..127
// so every SIMD carries ~3100 (Riemann) or ~2700 (trace) wave-instructions per plane, both of its waves busy.
// Per iteration kk (ONE phase, then a short one):  all threads issue the loads of the inputs of plane kk+3;
// trace waves: T(kk+1) -> buffer (kk+1) & 1 from the staged Q / B (kk .. kk+2), E (kk+1, kk+2);
// Riemann waves: problems of plane kk from T(kk) (buffer kk & 1) and the carried states;  barrier;
// the staged values go to the LDS slots plane kk (Q / B) and kk+1 (E) have just vacated;  barrier.
template <int SPEC>
__global__ void __launch_bounds__(MH_THREADS) mhd3d_sweep_kernel(DevParams g, TileGrid tg, const double* __restrict__ U,
                                                               const double* __restrict__ Q, const double* __restrict__ E,
                                                               double* __restrict__ F, double* __restrict__ emf,
                                                               double dtdx, double dtdy, double dtdz, int ra, int rb) {
  spec_assume<SPEC>(g);
  __shared__ double LT[2 * MH_BUF];          // T of planes kk (read) and kk+1 (written)   (buffer = plane & 1)
  __shared__ double LQ[3 * MH_QBSLOT];       // Q / B of planes kk .. kk+2                  (slot = plane % 3)
  __shared__ double LE[2 * MH_ESLOT];        // E of planes kk+1, kk+2                      (slot = plane & 1)

  const int b = (int)blockIdx.x;
  const int lin = (b & 7) * tg.per_xcd + (b >> 3);
  if ((b >> 3) >= tg.per_xcd || lin >= tg.nbx * tg.nby * tg.nseg) return;
  const int bx = lin % tg.nbx;
  const int by = (lin / tg.nbx) % tg.nby;
  const int seg = lin / (tg.nbx * tg.nby);
  const int span = rb - ra;
  const int sa = ra + (int)(((long long)span * seg) / tg.nseg);
  const int sb = ra + (int)(((long long)span * (seg + 1)) / tg.nseg);
  if (sb <= sa) return;

  const int gw = g.gw;
  const int i0 = gw + bx * MH_OX, j0 = gw + by * MH_OY;   // first cell of the tile
  const int t = (int)threadIdx.x;
  const size_t N = g.ncell;
  const unsigned sk = g.sk;

  // staging role: item it = r * 512 + t is component it / 209 (0-7 Q, 8-10 face field, 11-13 E) of input cell it % 209
  const double* sp[MH_NSTAGE];
#pragma unroll
  for (int r = 0; r < MH_NSTAGE; ++r) {
    const int it = r * MH_THREADS + t;
    const int comp = it / MH_QCELLS, cell = it - comp * MH_QCELLS;
    const int qy = cell / MH_QX, qx = cell - qy * MH_QX;
    const int si = i0 - 2 + qx, sj_ = j0 - 2 + qy;
    const double* base = comp < 8 ? Q + (size_t)comp * N : comp < 11 ? U + (size_t)(IA + comp - 8) * N : E + (size_t)(comp - 11) * N;
    sp[r] = (it < MH_ITEMS && si < g.isize && sj_ < g.jsize) ? base + (size_t)si + (size_t)sj_ * g.sj : nullptr;
  }
  double sv[MH_NSTAGE];
  auto stage_load = [&](int k) {
#pragma unroll
    for (int r = 0; r < MH_NSTAGE; ++r) sv[r] = sp[r] ? sp[r][(size_t)k * sk] : 0.0;
  };
  auto stage_store = [&](int k) {
    double* qd = LQ + (k % 3) * MH_QBSLOT;
    double* ed = LE + (k & 1) * MH_ESLOT;
#pragma unroll
    for (int r = 0; r < MH_NSTAGE; ++r) {
      const int it = r * MH_THREADS + t;
      if (it < MH_QBSLOT) qd[it] = sv[r];
      else if (it < MH_ITEMS) ed[it - MH_QBSLOT] = sv[r];
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const bool tracer = (wave & 3) == 3;

  // trace role
  auto trace_cell = [&](int k, int cell) {
    const int ty = cell / MH_PX, tx = cell - ty * MH_PX;
    const int ti = i0 - 1 + tx, tj = j0 - 1 + ty;
    if (cell < MH_CELLS && ti <= g.isize - gw && tj <= g.jsize - gw) {   // low bounds hold by construction
      const IJK c = {ti, tj, k};
      const TLdsWrite tw = {LT + (k & 1) * MH_BUF + cell};
      const TraceInLds in = {{LQ + ((k - 1) % 3) * MH_QBSLOT, LQ + (k % 3) * MH_QBSLOT, LQ + ((k + 1) % 3) * MH_QBSLOT},
                             {LE + (k & 1) * MH_ESLOT, LE + ((k + 1) & 1) * MH_ESLOT}};
      mhd_trace3d_at(g, in, tw, dtdx, dtdy, dtdz, c, (unsigned)((ty + 1) * MH_QX + tx + 1));
    }
  };

  // Riemann role
  const int dir = wave & 3;
  const int cl = (wave >> 2) * 64 + lane;
  const int oy = cl / MH_OX, ox = cl - oy * MH_OX;
  const int ci = i0 + ox, cj = j0 + oy;
  const bool fl_ok = !tracer && ci <= g.isize - gw && cj <= g.jsize - gw;
  const unsigned cidx2 = fl_ok ? (unsigned)ci + (unsigned)cj * g.sj : 0u;
  const unsigned cm = (unsigned)((oy + 1) * MH_PX + ox + 1);
  const double xPos = g.xMin + g.dx / 2 + (ci - gw) * g.dx;
  Prim8 c0, c1;
  c0.r = c0.p = 1.0; c0.u = c0.v = c0.w = c0.a = c0.b = c0.c = 0.0;
  c1 = c0;

#ifdef RG_SWEEP_PROF
  long long acc[4] = {0, 0, 0, 0};
#endif
  // prologue: inputs of planes sa-2, sa-1, sa (trace(sa-1) reads them)
  stage_load(sa - 2); stage_store(sa - 2);
  stage_load(sa - 1); stage_store(sa - 1);
  __syncthreads();   // E of plane sa-2 and of plane sa share a slot
  stage_load(sa); stage_store(sa);
  __syncthreads();
  // iteration kk: trace(kk+1) next to the Riemann problems of plane kk.  kk = sa-2 only traces plane sa-1, kk = sa-1 traces
  // plane sa and builds the carried states from T(sa-1); the last iteration kk = sb-1 has nothing left to trace.
  for (int kk = sa - 2; kk < sb; ++kk) {
    RG_PROF_T(tA);
    const bool more = kk + 3 <= sb;
    if (more) stage_load(kk + 3);
    if (tracer) {
      if (kk + 1 < sb) {
        if (wave == 3) { trace_cell(kk + 1, lane); trace_cell(kk + 1, 128 + lane); }
        else trace_cell(kk + 1, 64 + lane);
      }
    } else if (fl_ok && kk >= sa - 1) {
      const TLdsPlane Tk = {LT + (kk & 1) * MH_BUF};
      const unsigned idx = cidx2 + (unsigned)kk * sk;
      const bool solve = kk >= sa;
      if (dir == 0) riemann_dir<XD>(g, Tk, cm, xPos, F, emf, idx, c0, c1, solve);
      else if (dir == 1) riemann_dir<YD>(g, Tk, cm, xPos, F, emf, idx, c0, c1, solve);
      else riemann_dir<ZD>(g, Tk, cm, xPos, F, emf, idx, c0, c1, solve);
    }
    RG_PROF_T(tB);
    __syncthreads();
    RG_PROF_T(tC);
    if (more) stage_store(kk + 3);   // Q / B slot of plane kk and E slot of plane kk+1: dead since trace(kk+1)
    RG_PROF_T(tD);
    __syncthreads();
#ifdef RG_SWEEP_PROF
    const long long tE = (long long)__builtin_readcyclecounter();
    acc[0] += tB - tA; acc[1] += tC - tB; acc[2] += tD - tC; acc[3] += tE - tD;
#endif
  }
#ifdef RG_SWEEP_PROF
  if ((t & 63) == 0)
    for (int q = 0; q < 4; ++q) atomicAdd(&rg_prof[(t >> 6) * 4 + q], (unsigned long long)acc[q]);
#endif
}

template <int SPEC>
inline int launch_mhd3d_sweep(rg_stream_t s, const DevParams& g, const double* U, const double* Q, const double* E, double* F,
                              double* emf, double dtdx, double dtdy, double dtdz, int ra, int rb) {
  TileGrid tg;
  tg.nbx = (g.isize - 2 * g.gw + 1 + MH_OX - 1) / MH_OX;   // cells gw .. isize-gw
  tg.nby = (g.jsize - 2 * g.gw + 1 + MH_OY - 1) / MH_OY;
  const int span = rb - ra;
  static const int zseg_env = std::getenv("RGPU_ZSEG") ? std::atoi(std::getenv("RGPU_ZSEG")) : 0;
  int nseg;
  if (zseg_env > 0) nseg = (span + zseg_env - 1) / zseg_env;
  else {
    // one workgroup per CU is resident; a segment costs two extra iterations (pipeline fill).  Segments of ~64 planes
    // measured best at 512^3 (35.1 ms against 36.2 for one 513-plane march and 35.6 for 32-plane segments): enough
    // workgroups to even out the last round over the 256 CUs.  Small boxes: at least ~8 rounds, segments >= 8 planes.
    nseg = (span + 63) / 64;
    const int want = (2048 + tg.nbx * tg.nby - 1) / (tg.nbx * tg.nby);
    if (nseg < want) nseg = want;
    if (nseg > span / 8) nseg = span / 8;
  }
  if (nseg < 1) nseg = 1;
  if (nseg > span) nseg = span;
  tg.nseg = nseg;
  const int total = tg.nbx * tg.nby * tg.nseg;
  tg.per_xcd = (total + 7) / 8;
  hipLaunchKernelGGL((mhd3d_sweep_kernel<SPEC>), dim3(8u * (unsigned)tg.per_xcd), dim3(MH_THREADS), 0, s, g, tg, U, Q, E, F, emf,
                     dtdx, dtdy, dtdz, ra, rb);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

#ifdef RG_SWEEP_PROF
extern "C" inline void rgpu_prof_read_impl(unsigned long long* out, int reset) {
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(rg_prof), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(rg_prof), z, sizeof(z)); }
}
#endif

// configurations the fused sweep covers (everything but the per-cell gravity field, which the driver excludes itself)
inline bool mhd3d_sweep_covers(const DevParams& g) { return tiled_enabled() && g.three_d && g.mhd; }

// spec: 0 = generic, 1 = isothermal rotating box (MRI), 2 = adiabatic inertial box (the driver's pick_spec)
// Solves the Riemann problems of planes [ra, rb) (already clipped to [gw, ksize-gw]) from U, Q, E.
// Returns 0 = done, 1 = not applicable (caller runs trace + Riemann as flat kernels), < 0 = launch error.
template <int SPEC_MRI, int SPEC_PLAIN>
inline int mhd3d_sweep(rg_stream_t s, const DevParams& g, int spec, const double* U, const double* Q, const double* E, double* F,
                       double* emf, double dtdx, double dtdy, double dtdz, int ra, int rb) {
  if (!mhd3d_sweep_covers(g) || g.grav_on == 2) return 1;   // per-cell gravity field: flat kernels
  if (rb <= ra) return 0;
  if (spec == 1) return launch_mhd3d_sweep<SPEC_MRI>(s, g, U, Q, E, F, emf, dtdx, dtdy, dtdz, ra, rb);
  if (spec == 2) return launch_mhd3d_sweep<SPEC_PLAIN>(s, g, U, Q, E, F, emf, dtdx, dtdy, dtdz, ra, rb);
  return launch_mhd3d_sweep<SPEC_NONE>(s, g, U, Q, E, F, emf, dtdx, dtdy, dtdz, ra, rb);
}

}  // namespace rgpu_tiled
