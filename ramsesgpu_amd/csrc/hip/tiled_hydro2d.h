// tiled_hydro2d.h (HIP / gfx950 only) -- the whole 2D hydro unsplit step (primitives, slopes + trace, two Riemann problems per cell,
// conservative update, CFL term of the new state) as ONE LDS-tiled kernel: U -> Unew.
//
// Flat pipeline (kernels_hydro.h): K_hydro_prim, K_hydro_trace, K_hydro_flux, K_hydro_update -- four launches that pass Q (4),
// T (12) and F (8 doubles per cell) through L2 / HBM: 0.063 ms per step at 512^2 (a chain of 7-16 us kernels), 1.81 ms at
// 4096^2 (HBM-bound at ~400 B per cell).  Reference idiom: the one-kernel 2D step with a shared-memory tile and a ghost overlap,
// godunov_unsplit.cuh (kernel_godunov_unsplit_2d_v1), HydroRunGodunov.cpp:779-1005.
//
// One thread per cell of a TX x TY tile (tiles overlap by one cell per side: the inner (TX-2) x (TY-2) threads finish a cell):
//   phase 0  U of the cell -> primitives -> LDS (the one-cell ring around the tile by the first 2 TX + 2 TY threads)
//   phase 1  slopes + trace of the cell (hydro_trace_cell); the states at its HIGH x / y faces -> LDS, the LOW ones stay in registers
//   phase 2  Riemann problems at its LOW x / y faces (hydro_flux_cell) -> fluxes to LDS
//   phase 3  update (hydro_update_cell: both update orders), uniform-gravity source, CFL term of the new cell into the device slots
// Three barriers, 43 KB of LDS.  Same expressions, same operand order, same bits as the flat kernels.
#pragma once
#include "step_clock.h"

namespace rgpu_tiled {

template <int TX, int TY>
struct Hydro2dTile {
  double q[4][TY + 2][TX + 2];   // primitives: tile + ring
  double qm[2][4][TY][TX];       // state at the HIGH x / y face of each cell (grid frame, floors and gravity predictor applied)
  double f[2][4][TY][TX];        // flux through the LOW x / y face of each cell (face-normal frame)
};

// The ghost cells an interior cell is the source of, along one direction (bc_face_cell: dirichlet = mirror image with the normal
// momentum negated, neumann = copies of the first / last interior cell, periodic = the image one period away).  x = the cell's index,
// n = interior cells, gw = ghost width (n >= gw), bc = the face types (RGPU_BC_DIRICHLET 1, NEUMANN 2, PERIODIC 3).
struct ImgDim {
  int x, nlo, lo0, nhi, hi0;
  bool fliplo, fliphi;
  RG_DEVFN int count() const { return 1 + nlo + nhi; }
  RG_DEVFN int coord(int e) const { return e == 0 ? x : (e <= nlo ? lo0 + (e - 1) : hi0 + (e - 1 - nlo)); }
  RG_DEVFN bool flip(int e) const { return e == 0 ? false : (e <= nlo ? fliplo : fliphi); }
};
RG_DEVFN ImgDim images_of(int x, int n, int gw, int bc_lo, int bc_hi) {
  ImgDim d = {x, 0, 0, 0, 0, bc_lo == 1, bc_hi == 1};
  // low ghosts [0, gw)
  if (bc_lo == 1) { if (x < 2 * gw) { d.nlo = 1; d.lo0 = 2 * gw - 1 - x; } }
  else if (bc_lo == 2) { if (x == gw) { d.nlo = gw; d.lo0 = 0; } }
  else { if (x >= n) { d.nlo = 1; d.lo0 = x - n; } }
  // high ghosts [n + gw, n + 2 gw)
  if (bc_hi == 1) { if (x >= n) { d.nhi = 1; d.hi0 = 2 * n + 2 * gw - 1 - x; } }
  else if (bc_hi == 2) { if (x == n + gw - 1) { d.nhi = gw; d.hi0 = n + gw; } }
  else { if (x < 2 * gw) { d.nhi = 1; d.hi0 = x + n; } }
  return d;
}

// images != 0 (bit 12 set, the four face types in bits 2f .. 2f+1; caller: whole-domain step, every face dirichlet / neumann /
// periodic, no jet, nothing modifies the new state after this kernel): the interior cells also write the ghost cells the next
// step's ghost fill (X, then Y over the full extent: corners are images of images) would copy them into -- the same doubles -- and
// the ghost cells' own threads do not store: one writer per location, and that fill is not launched (ghost_ok_parity).
template <int TX, int TY, int SPEC>
__global__ void __launch_bounds__(TX * TY) hydro2d_step_kernel(DevParams g, int nbx, const double* __restrict__ Uin, double* __restrict__ Uout,
                                                               double dtdx, double dtdy, unsigned long long* dt_slots, int images, const StepClock* clk, ClockFold fold) {
  spec_assume<SPEC>(g);
  if (fold.out) {   // the clock of this step is part of the kernel (step_clock.h: clock_fold)
    __shared__ double Lred[TX * TY / 64];
    const StepClock r = clock_fold<TX * TY>(fold, Lred);
    if (r.stop) return;
    dtdx = rg_uniform(r.dtdx); dtdy = rg_uniform(r.dtdy);
  } else if (clk) {   // the time step lives on the device (hip/step_clock.h)
    if (clk->stop) return;
    dtdx = clk->dtdx; dtdy = clk->dtdy;
  }
  constexpr int NV = 4;
  constexpr int RING = 2 * TX + 2 * TY;
  static_assert(RING <= TX * TY, "ring cells are handled by the first RING threads");
  __shared__ Hydro2dTile<TX, TY> L;

  const int t = (int)threadIdx.x;
  const int by = (int)blockIdx.x / nbx, bx = (int)blockIdx.x - by * nbx;
  const int ti = t % TX, tj = t / TX;
  const int i = bx * (TX - 2) + ti, j = by * (TY - 2) + tj;
  const bool ina = i < g.isize && j < g.jsize;
  const size_t N = g.ncell;
  const unsigned idx2 = ina ? (unsigned)i + (unsigned)j * g.sj : 0u;
  const int gw = g.gw;
  // cells this thread writes: the inner threads of the tile, plus array row / column 0 (never inner cells)
  const bool own = ina && ((ti >= 1 && ti < TX - 1) || i == 0) && ((tj >= 1 && tj < TY - 1) || j == 0);
  const bool inner = i >= gw && i < g.isize - gw && j >= gw && j < g.jsize - gw;

  // ---- phase 0: primitives of the tile and of its ring (corners excluded: no stencil reads them) ----
  double u[NV], q[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) { u[v] = 1.0; q[v] = 1.0; }
  if (ina) {
#pragma unroll
    for (int v = 0; v < NV; ++v) u[v] = Uin[idx2 + v * N];
  }
  {
    int rti, rtj;
    if (t < TX) { rti = t; rtj = -1; }
    else if (t < 2 * TX) { rti = t - TX; rtj = TY; }
    else if (t < 2 * TX + TY) { rti = -1; rtj = t - 2 * TX; }
    else { rti = TX; rtj = t - 2 * TX - TY; }
    const int ri = bx * (TX - 2) + rti, rj = by * (TY - 2) + rtj;
    const bool ring = t < RING && ri >= 0 && ri < g.isize && rj >= 0 && rj < g.jsize;
    double ur[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) ur[v] = 1.0;
    if (ring) {
#pragma unroll
      for (int v = 0; v < NV; ++v) ur[v] = Uin[(unsigned)ri + (unsigned)rj * g.sj + v * N];
    }
    if (ina) hydro_prim<NV>(g, u, q);
#pragma unroll
    for (int v = 0; v < NV; ++v) L.q[v][tj + 1][ti + 1] = q[v];
    if (t < RING) {
      double rq[NV] = {1.0, 1.0, 1.0, 1.0};
      if (ring) hydro_prim<NV>(g, ur, rq);
#pragma unroll
      for (int v = 0; v < NV; ++v) L.q[v][rtj + 1][rti + 1] = rq[v];
    }
  }
  __syncthreads();

  // ---- phase 1: slopes and trace of the cell (hydro_trace_cell, ND = 2) ----
  double qpx[NV], qpy[NV];
  {
    const double st = g.slope_type;
    const double gamma = g.gamma0;
    double h[2][NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double nb[2][2] = {{L.q[v][tj + 1][ti], L.q[v][tj + 1][ti + 2]}, {L.q[v][tj][ti + 1], L.q[v][tj + 2][ti + 1]}};
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        double s;
        if (st == 0) s = 0.0;
        else s = tvd_half_slope(st, nb[d][0], q[v], nb[d][1]);
        h[d][v] = s;
      }
    }
    const double r = q[ID], p = q[IP], uu = q[IU], vv = q[IV];
    const double drx = h[0][ID], dpx = h[0][IP], dux = h[0][IU], dvx = h[0][IV];
    const double dry = h[1][ID], dpy = h[1][IP], duy = h[1][IU], dvy = h[1][IV];
    const rg_recip_t inv_r = rg_recip(r);
    const double sr0 = (-uu * drx - dux * r) * dtdx + (-vv * dry - dvy * r) * dtdy;
    const double su0 = (-uu * dux - rg_div(dpx, inv_r)) * dtdx + (-vv * duy) * dtdy;
    const double sv0 = (-uu * dvx) * dtdx + (-vv * dvy - rg_div(dpy, inv_r)) * dtdy;
    const double sp0 = (-uu * dpx - dux * gamma * p) * dtdx + (-vv * dpy - dvy * gamma * p) * dtdy;
    double tq[NV];
    tq[ID] = r + sr0; tq[IU] = uu + su0; tq[IV] = vv + sv0; tq[IP] = p + sp0;
    // face states with the floors of trace.h:388-389 (hydro_face_state), grid frame
    double qmx[NV], qmy[NV];
#pragma unroll
    for (int n = 0; n < NV; ++n) {
      qmx[n] = tq[n] + h[0][n]; qpx[n] = tq[n] - h[0][n];
      qmy[n] = tq[n] + h[1][n]; qpy[n] = tq[n] - h[1][n];
    }
#define RG_FLOOR(a) a[ID] = fmax(g.smallr, a[ID]); a[IP] = fmax(g.smallp * a[ID], a[IP])
    RG_FLOOR(qmx); RG_FLOOR(qpx); RG_FLOOR(qmy); RG_FLOOR(qpy);
#undef RG_FLOOR
    if (g.grav_on) {   // uniform static gravity: predictor on the traced states, after the floors (hydro_face_state)
#define RG_GRAV(a) a[IU] += g.hgx; a[IV] += g.hgy
      RG_GRAV(qmx); RG_GRAV(qpx); RG_GRAV(qmy); RG_GRAV(qpy);
#undef RG_GRAV
    }
#pragma unroll
    for (int n = 0; n < NV; ++n) { L.qm[0][n][tj][ti] = qmx[n]; L.qm[1][n][tj][ti] = qmy[n]; }
  }
  __syncthreads();

  // ---- phase 2: Riemann problems at the two low faces of the cell (hydro_flux_cell) ----
  double fx[NV], fy[NV];
  {
    const int tim = ti > 0 ? ti - 1 : 0, tjm = tj > 0 ? tj - 1 : 0;
    double ql[NV], qr[NV];
    // x: normal frame = grid frame
#pragma unroll
    for (int n = 0; n < NV; ++n) { ql[n] = L.qm[0][n][tj][tim]; qr[n] = qpx[n]; fx[n] = 0.0; }
    hydro_riemann<NV>(g, ql, qr, fx);
    // y: IU <-> IV
    ql[ID] = L.qm[1][ID][tjm][ti]; ql[IP] = L.qm[1][IP][tjm][ti]; ql[IU] = L.qm[1][IV][tjm][ti]; ql[IV] = L.qm[1][IU][tjm][ti];
    qr[ID] = qpy[ID]; qr[IP] = qpy[IP]; qr[IU] = qpy[IV]; qr[IV] = qpy[IU];
#pragma unroll
    for (int n = 0; n < NV; ++n) fy[n] = 0.0;
    hydro_riemann<NV>(g, ql, qr, fy);
#pragma unroll
    for (int n = 0; n < NV; ++n) { L.f[0][n][tj][ti] = fx[n]; L.f[1][n][tj][ti] = fy[n]; }
  }
  __syncthreads();

  // ---- phase 3: update (hydro_update_cell), CFL term of the new state ----
  double inv = 0.0;
  if (own) {
    const double rho_old = u[ID];
    if (inner) {
      const int tip = ti + 1 < TX ? ti + 1 : ti, tjp = tj + 1 < TY ? tj + 1 : tj;
#define RG_LOW_X u[ID] += fx[ID] * dtdx; u[IP] += fx[IP] * dtdx; u[IU] += fx[IU] * dtdx; u[IV] += fx[IV] * dtdx
#define RG_LOW_Y u[ID] += fy[ID] * dtdy; u[IP] += fy[IP] * dtdy; u[IU] += fy[IV] * dtdy; u[IV] += fy[IU] * dtdy
#define RG_HIGH_X u[ID] -= L.f[0][ID][tj][tip] * dtdx; u[IP] -= L.f[0][IP][tj][tip] * dtdx; u[IU] -= L.f[0][IU][tj][tip] * dtdx; u[IV] -= L.f[0][IV][tj][tip] * dtdx
#define RG_HIGH_Y u[ID] -= L.f[1][ID][tjp][ti] * dtdy; u[IP] -= L.f[1][IP][tjp][ti] * dtdy; u[IU] -= L.f[1][IV][tjp][ti] * dtdy; u[IV] -= L.f[1][IU][tjp][ti] * dtdy
      if (!g.dirwise_update) { RG_LOW_X; RG_LOW_Y; RG_HIGH_X; RG_HIGH_Y; }   // unsplitVersion 1: low faces, then high faces
      else { RG_LOW_X; RG_HIGH_X; RG_LOW_Y; RG_HIGH_Y; }                     // unsplitVersion 2: direction by direction
#undef RG_LOW_X
#undef RG_LOW_Y
#undef RG_HIGH_X
#undef RG_HIGH_Y
      if (g.grav_on) {   // momentum source with the mean of the old and new density; energy untouched
        const double rho_sum = rho_old + u[ID];
        u[IU] += g.hgx * rho_sum; u[IV] += g.hgy * rho_sum;
      }
      if (dt_slots) {
        double qn[NV];
        const double cs = hydro_prim<NV>(g, u, qn);
        inv = (cs + fabs(qn[IU])) / g.dx + (cs + fabs(qn[IV])) / g.dy;
      }
    }
    if (!images) {
#pragma unroll
      for (int v = 0; v < NV; ++v) Uout[idx2 + v * N] = u[v];
    } else if (inner && i >= 2 * gw && i < g.nx && j >= 2 * gw && j < g.ny) {   // no ghost cell is an image of this cell
#pragma unroll
      for (int v = 0; v < NV; ++v) Uout[idx2 + v * N] = u[v];
    } else if (inner) {
      const ImgDim ix = images_of(i, g.nx, gw, images & 3, (images >> 2) & 3), iy = images_of(j, g.ny, gw, (images >> 4) & 3, (images >> 6) & 3);
      const int nxi = ix.count(), nyi = iy.count();
      for (int b = 0; b < nyi; ++b)
        for (int a = 0; a < nxi; ++a) {
          double* o = Uout + (size_t)ix.coord(a) + (size_t)iy.coord(b) * g.sj;
          o[ID * N] = u[ID];
          o[IP * N] = u[IP];
          o[IU * N] = ix.flip(a) ? u[IU] * -1.0 : u[IU];
          o[IV * N] = iy.flip(b) ? u[IV] * -1.0 : u[IV];
        }
    }
  }
  if (dt_slots) rgpu::rg_slot_max_wave(dt_slots + (((unsigned)blockIdx.x * (unsigned)(TX * TY / 64) + (unsigned)(t >> 6)) & (rgpu::RG_DT_SLOTS - 1)), inv);
}

template <int TX, int TY, int SPEC>
inline int launch_hydro2d_step(rg_stream_t s, const DevParams& g, const double* in, double* out, double dtdx, double dtdy, unsigned long long* dt_slots, int images, const StepClock* clk, const ClockFold& fold) {
  const int nbx = (g.isize - 1 + (TX - 2) - 1) / (TX - 2);   // owners cover i in [1, nbx*(TX-2)] plus column 0
  const int nby = (g.jsize - 1 + (TY - 2) - 1) / (TY - 2);
  hipLaunchKernelGGL((hydro2d_step_kernel<TX, TY, SPEC>), dim3((unsigned)(nbx * nby)), dim3(TX * TY), 0, s, g, nbx, in, out, dtdx, dtdy, dt_slots, images, clk, fold);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// configurations the fused 2D hydro step covers (the per-cell gravity field runs the flat kernels' own instantiations)
inline bool hydro2d_step_covers(const DevParams& g) { return tiled_enabled() && !g.three_d && !g.mhd && g.nvar == 4 && g.grav_on != 2; }

// The whole 2D hydro step U -> Unew.  dt_slots: RG_DT_SLOTS device slots for the CFL maximum of the new state (reset by the caller),
// or 0; images: see the kernel.  Returns 0 = done, 1 = not covered (the caller runs the flat kernels), < 0 = launch error.
inline int hydro2d_step(rg_stream_t s, const DevParams& g, const double* in, double* out, double dtdx, double dtdy, unsigned long long* dt_slots, int images, const StepClock* clk = 0,
                        const ClockFold* fold_in = 0) {
  if (!hydro2d_step_covers(g)) return 1;
  ClockFold fold;
  if (fold_in) fold = *fold_in; else { fold.prev = 0; fold.out = 0; fold.in = 0; fold.zero = 0; fold.t0 = 0.0; fold.tEnd = 0.0; }
  const bool no_spec = !rgpu::options().spec;
  constexpr int TX = 16, TY = 16;
#define RG_TRY(SP) if (spec_matches(SP, g)) return launch_hydro2d_step<TX, TY, SP>(s, g, in, out, dtdx, dtdy, dt_slots, images, clk, fold);
  if (!no_spec) {
    const int SL1 = SPEC_SLOPE1 | SPEC_NO_GRAVITY, SL2 = SPEC_SLOPE2 | SPEC_NO_GRAVITY;
    RG_TRY(SPEC_HYDRO_HLLC | SL2) RG_TRY(SPEC_HYDRO_HLLC | SL1)
    RG_TRY(SPEC_HYDRO_APPROX | SL2) RG_TRY(SPEC_HYDRO_APPROX | SL1)
    RG_TRY(SPEC_HYDRO_HLL | SL2) RG_TRY(SPEC_HYDRO_HLL | SL1)
    RG_TRY(SPEC_HYDRO_APPROX | SPEC_SLOPE2) RG_TRY(SPEC_HYDRO_APPROX | SPEC_SLOPE1)   // with uniform gravity
  }
#undef RG_TRY
  return launch_hydro2d_step<TX, TY, SPEC_NONE>(s, g, in, out, dtdx, dtdy, dt_slots, images, clk, fold);
}

}  // namespace rgpu_tiled
