// tiled_hydro.h (HIP / gfx950 only) -- the 3D hydro unsplit step as ONE cooperative, LDS-tiled, z-marching kernel.
//
// The flat pipeline (kernels_hydro.h: prim -> trace -> flux -> update, one thread per cell, four launches) moves
// ~760 B per cell update through HBM / L2 for 80 B of state.  Here a workgroup owns an x-y tile of TX x TY cell
// columns and marches along z; everything between "read U" and "write Unew" lives in registers and LDS:
//
//   * a thread owns one (i,j) column.  Its z neighbourhood (primitives of planes kk-1, kk, kk+1, the conservative
//     state of planes kk, kk+1, the z face state qm_z of plane kk-1, the partially updated cell of plane kk-1) rides
//     in registers from one iteration to the next -- the z halo costs nothing;
//   * the x / y neighbourhood goes through LDS: the primitives of plane kk (tile + a one-cell ring without corners,
//     the ring recomputed from U by the first 2(TX+TY) threads), the x / y face states qm (the left states of the
//     neighbours' low faces) and the x / y fluxes (the high-face fluxes of the neighbours);
//   * per plane, two barriers: slopes + trace(kk) | barrier | 3 Riemann problems at the low faces of (i,j,kk) | finish
//     cell (i,j,kk-1) with the z flux just computed and store it | publish the x / y fluxes and the primitives of plane
//     kk+1 | barrier | gather the x / y fluxes into cell (i,j,kk) | prim(kk+2).
//
// Thread tiles overlap by one cell on each side (the outermost threads only supply face states), so a TX x TY
// workgroup updates (TX-2) x (TY-2) columns.  HBM traffic per cell update: 40 B read (+ the tile overlap, served by L2
// when neighbouring tiles run on the same XCD) + 40 B written; nothing else leaves the CU.
// (Reference idiom: the shared-memory tiles of godunov_unsplit.cuh:1829-1988, 3212-3488 -- z-marching with the
// flux gathered through shared memory; this is the gfx950 counterpart: wave64, 160 KB LDS, register z-pipeline.)
//
// Arithmetic: exactly the expressions of kernels_hydro.h (hydro_trace_cell / hydro_face_state / hydro_update_cell)
// on the same operands in the same order, hence the same bits as the flat kernels and as the reference.
#pragma once
#include <cstdlib>

#include "launchers.h"
#include "rg_options.h"

namespace rgpu_tiled {

using namespace rgpu_dev;
using rgpu::rg_stream_t;


inline bool tiled_enabled() {
  static const char* v = std::getenv("RGPU_TILED");   // RGPU_TILED=0: the flat per-cell kernels everywhere (include/rgpu.h, "Environment")
  static const bool on = !(v && std::atoi(v) == 0);
  return on;
}

// copy of whole z planes (ghost planes inside a requested plane range: the flat update kernel copies them too)
struct K_copy_cells {
  const double* src; double* dst; unsigned long long ncell; int nvar; const StepClock* clk;
  RG_DEVFN void operator()(unsigned idx) const {
    if (clk && clk->stop) return;   // (a stopped step of a device-clock batch writes nothing)
    for (int v = 0; v < nvar; ++v) dst[idx + (size_t)v * ncell] = src[idx + (size_t)v * ncell];
  }
};

template <int TX, int TY>
struct HydroTile {
  double q[5][TY + 2][TX + 2];   // primitives of plane kk: tile + ring
  double qm[2][5][TY][TX];       // qm_x, qm_y: state at the HIGH x / y face of each cell (grid frame, floors applied)
  double f[2][5][TY][TX];        // flux through the LOW x / y face of each cell (face-normal frame)
};

// map the flat block index to (tile x, tile y, z segment) so that each XCD (block b runs on XCD b % 8) owns a
// contiguous run of tiles -- x fastest, then y, then z segment: overlapping tile edges are then re-read from that
// XCD's own L2
// Work items = (tile, z segment), tile fastest, dealt to the XCDs in contiguous runs of ipx items.  All workgroups of a launch
// take the same time per plane, so an XCD works through its run in rounds of `slots` resident workgroups and a partly filled
// last round would cost a whole segment.  Therefore the items of the last, incomplete round are cut once more, each into
// `tail` sub-segments: up to `slots` short workgroups that finish in 1/tail of a round.  Block b -> XCD b & 7, local index
// b >> 3: the first `full` local indices are whole items, the rest sub-segments of the remaining items.
struct TileGrid {
  int nbx, nby, nseg;      // tiles and base z segments per tile
  int ipx, full, tail;     // items per XCD; whole items per XCD; sub-segments per item of the last round (>= 1)
  int per_xcd;             // workgroups per XCD = full + (ipx - full) * tail
  // a second plane range of the SAME length in the same launch (the two boundary ranges of a slab): the kernel's [za, zb) is then
  // the concatenation, 2 L planes long; planes at or beyond zsplit = za + L belong to the second range and lie zgap planes further
  // up.  nseg is even, so no segment straddles the seam.  One range: zsplit = INT_MAX, zgap = 0.
  int zsplit, zgap;
};
struct TileItem { int bx, by, sa, sb; bool valid; };
// planes [sa, sb) of [za, zb) and the tile of block b
RG_DEVFN TileItem tile_item(const TileGrid& tg, int b, int za, int zb) {
  TileItem it;
  const int xcd = b & 7, l = b >> 3;
  int item, sub = 0, nsub = 1;
  if (l < tg.full) item = xcd * tg.ipx + l;
  else { const int q = l - tg.full; item = xcd * tg.ipx + tg.full + q / tg.tail; sub = q % tg.tail; nsub = tg.tail; }
  const int tiles = tg.nbx * tg.nby;
  it.valid = l < tg.per_xcd && item < (xcd + 1) * tg.ipx && item < tiles * tg.nseg;
  const int t = it.valid ? item % tiles : 0, seg = it.valid ? item / tiles : 0;
  it.bx = t % tg.nbx;
  it.by = t / tg.nbx;
  const int span = zb - za;
  const int a0 = za + (int)(((long long)span * seg) / tg.nseg), b0 = za + (int)(((long long)span * (seg + 1)) / tg.nseg);
  it.sa = a0 + (int)(((long long)(b0 - a0) * sub) / nsub);
  it.sb = a0 + (int)(((long long)(b0 - a0) * (sub + 1)) / nsub);
  if (it.sb <= it.sa) it.valid = false;
  if (it.sa >= tg.zsplit) { it.sa += tg.zgap; it.sb += tg.zgap; }   // second range of a two-range launch
  return it;
}
// host: base segment count and tail split minimising the modelled duration (iterations of the z march; a segment costs
// `fill` extra iterations), for `slots` resident workgroups per XCD
// pair: the launch covers two ranges of `span` planes each (TileGrid::zsplit): twice the base segments of the one-range plan
inline void tile_grid_plan(TileGrid& tg, int span, int slots, int min_planes, int fill, int zseg_env, bool pair = false) {
  tg.zsplit = 0x7fffffff; tg.zgap = 0;
  if (pair) {
    TileGrid one = tg;
    one.nby *= 2;   // as many items as two launches: the last-round split is planned for all of them
    tile_grid_plan(one, span, slots, min_planes, fill, zseg_env, false);
    tg.nseg = 2 * one.nseg;
    const int items = tg.nbx * tg.nby * tg.nseg;
    tg.ipx = (items + 7) / 8;
    tg.full = (tg.ipx / slots) * slots;
    tg.tail = (tg.ipx - tg.full) > 0 ? one.tail : 1;
    tg.per_xcd = tg.full + (tg.ipx - tg.full) * tg.tail;
    return;
  }
  const int tiles = tg.nbx * tg.nby;
  int best_n = 1, best_tail = 1;
  double best = 1e300;
  const int nmax = zseg_env > 0 ? 1 : (span / min_planes < 1 ? 1 : (span / min_planes > 64 ? 64 : span / min_planes));
  for (int n0 = 1; n0 <= nmax; ++n0) {
    const int n = zseg_env > 0 ? (span + zseg_env - 1) / zseg_env : n0;
    const int items = tiles * n, ipx = (items + 7) / 8;
    const int full = (ipx / slots) * slots, rem = ipx - full;
    const double len = (double)span / n;
    int tail = 1;
    if (rem > 0) { tail = slots / rem; const int cap = (int)(len / min_planes); if (tail > cap) tail = cap; if (tail < 1) tail = 1; }
    const double t = (full / slots) * (len + fill) + (rem > 0 ? (len / tail + fill) * ((rem * tail + slots - 1) / slots) : 0.0);
    if (t < best * 0.995) { best = t; best_n = n; best_tail = tail; }   // ties: the fewer, longer segments
  }
  if (best_n > span) best_n = span;
  if (best_n < 1) best_n = 1;
  tg.nseg = best_n;
  const int items = tiles * tg.nseg;
  tg.ipx = (items + 7) / 8;
  tg.full = (tg.ipx / slots) * slots;
  tg.tail = (tg.ipx - tg.full) > 0 ? best_tail : 1;
  tg.per_xcd = tg.full + (tg.ipx - tg.full) * tg.tail;
}

// dslot != 0: the CFL scan of the NEW state rides along -- every updated cell contributes sum_d (c + |v_d|) / delta_d
// (hydro_invdt_cell) to a 64-bit atomicMax on *dslot, so that the next compute_dt needs no pass over U (all values are
// >= 0: the bit pattern orders like an unsigned integer; max is order independent, hence the same double as the scan)
template <int TX, int TY, int SPEC, int MINW = 1>
__global__ void __launch_bounds__(TX * TY, MINW) hydro3d_sweep_kernel(DevParams g, TileGrid tg, const double* __restrict__ Uin,
                                                              double* __restrict__ Uout, double dtdx, double dtdy,
                                                              double dtdz, int za, int zb, unsigned long long* dslot, const StepClock* clk) {
  spec_assume<SPEC>(g);
  if (clk) {   // the time step lives on the device (csrc/step_clock_rec.h)
    if (clk->stop) return;
    dtdx = clk->dtdx; dtdy = clk->dtdy; dtdz = clk->dtdz;
  }
  constexpr int NV = 5;
  constexpr int NT = TX * TY;
  constexpr int RING = 2 * TX + 2 * TY;
  static_assert(RING <= NT, "ring cells are handled by the first RING threads");
  __shared__ HydroTile<TX, TY> L;

  const TileItem item = tile_item(tg, (int)blockIdx.x, za, zb);   // this workgroup's tile and its planes [sa, sb) of [za, zb)
  if (!item.valid) return;   // whole workgroup leaves: no barrier is skipped
  const int bx = item.bx, by = item.by, sa = item.sa, sb = item.sb;

  const int t = (int)threadIdx.x;
  const int ti = t % TX, tj = t / TX;
  const int i = bx * (TX - 2) + ti, j = by * (TY - 2) + tj;
  const bool ina = i < g.isize && j < g.jsize;
  const size_t N = g.ncell;
  const unsigned sk = g.sk;
  const unsigned idx2 = ina ? (unsigned)i + (unsigned)j * g.sj : 0u;
  const int gw = g.gw;
  // cells this thread writes: the inner threads of the tile, plus array row / column 0 (never inside an inner range)
  const bool own = ina && ((ti >= 1 && ti < TX - 1) || i == 0) && ((tj >= 1 && tj < TY - 1) || j == 0);
  const bool inner2d = i >= gw && i < g.isize - gw && j >= gw && j < g.jsize - gw;

  // ring cell of this thread (threads 0 .. RING-1): the one-cell frame around the tile, corners excluded
  int rti, rtj;
  if (t < TX) { rti = t; rtj = -1; }
  else if (t < 2 * TX) { rti = t - TX; rtj = TY; }
  else if (t < 2 * TX + TY) { rti = -1; rtj = t - 2 * TX; }
  else { rti = TX; rtj = t - 2 * TX - TY; }
  const int ri = bx * (TX - 2) + rti, rj = by * (TY - 2) + rtj;
  const bool ring = t < RING && ri >= 0 && ri < g.isize && rj >= 0 && rj < g.jsize;
  const unsigned ridx2 = ring ? (unsigned)ri + (unsigned)rj * g.sj : 0u;

  const double st = g.slope_type;
  const double gamma = g.gamma0;

  double qA[NV], qB[NV], qC[NV];    // primitives of planes kk-1, kk, kk+1 of this column
  double uB[NV], uC[NV], uN[NV];    // conservative state of planes kk, kk+1, kk+2
  double qmz[NV];                    // state at the high z face of cell (i,j,kk-1), grid frame
  double up[NV];                     // cell (i,j,kk-1) with every flux but the one through its high z face applied
#pragma unroll
  for (int v = 0; v < NV; ++v) { qA[v] = 1.0; qB[v] = 1.0; qC[v] = 1.0; uB[v] = 1.0; uC[v] = 1.0; uN[v] = 1.0; qmz[v] = 1.0; up[v] = 0.0; }
  double inv_dt = 0.0;
  double rho_old = 1.0;             // old density of cell (i,j,kk-1): the gravity source needs it when the cell is finished
  // ring threads: U of the ring cell one plane AHEAD.  Loaded and converted in the same iteration, the conversion would
  // wait for the load (and, the memory counter being in-order, for the prefetch of plane kk+2 issued before it) with the
  // whole workgroup queued behind it at the barrier: one exposed memory latency per plane.
  double urn[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) urn[v] = 1.0;

  // prologue: planes sa-2, sa-1, sa
  {
    const int k0 = sa - 1;
    if (ring) {
#pragma unroll
      for (int v = 0; v < NV; ++v) urn[v] = Uin[ridx2 + (size_t)k0 * sk + v * N];
    }
    if (ina) {
      double ua[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        ua[v] = Uin[idx2 + (size_t)(k0 - 1) * sk + v * N];
        uB[v] = Uin[idx2 + (size_t)k0 * sk + v * N];
        uC[v] = Uin[idx2 + (size_t)(k0 + 1) * sk + v * N];
      }
      hydro_prim<NV>(g, ua, qA);
      hydro_prim<NV>(g, uB, qB);
      hydro_prim<NV>(g, uC, qC);
    }
    // primitives of plane sa-1 into LDS (every later plane is put there by the iteration before it)
    if (ring) {
      double rq[NV];
      hydro_prim<NV>(g, urn, rq);
#pragma unroll
      for (int v = 0; v < NV; ++v) L.q[v][rtj + 1][rti + 1] = rq[v];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) L.q[v][tj + 1][ti + 1] = qB[v];
    __syncthreads();
  }

  for (int kk = sa - 1; kk <= sb; ++kk) {
    // ---- A: issue the loads of plane kk+2 (consumed at the bottom of the iteration) and of the ring of plane kk+1
    // (consumed in E).  The primitives of plane kk are in LDS since the last barrier. ----
    const bool more = (kk + 2 <= sb + 1) && ina;
    if (more) {
#pragma unroll
      for (int v = 0; v < NV; ++v) uN[v] = Uin[idx2 + (size_t)(kk + 2) * sk + v * N];
    }
    if (ring && kk < sb) {
#pragma unroll
      for (int v = 0; v < NV; ++v) urn[v] = Uin[ridx2 + (size_t)(kk + 1) * sk + v * N];
    }

    // ---- B: slopes and trace of cell (i,j,kk)  (hydro_trace_cell) ----
    double q[NV], h[3][NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      q[v] = qB[v];
      const double nb[3][2] = {{L.q[v][tj + 1][ti], L.q[v][tj + 1][ti + 2]}, {L.q[v][tj][ti + 1], L.q[v][tj + 2][ti + 1]}, {qA[v], qC[v]}};
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double s;
        if (st == 0) s = 0.0;
        else if (st == 1) s = minmod_half_slope(nb[d][0], q[v], nb[d][1]);
        else s = tvd_half_slope(st, nb[d][0], q[v], nb[d][1]);
        h[d][v] = s;
      }
    }
    double qpx[NV], qpy[NV], qpz[NV], qmz_new[NV];
    {
      double r = q[ID], p = q[IP], u = q[IU], v = q[IV], w = q[IW];
      const double drx = h[0][ID], dpx = h[0][IP], dux = h[0][IU], dvx = h[0][IV], dwx = h[0][IW];
      const double dry = h[1][ID], dpy = h[1][IP], duy = h[1][IU], dvy = h[1][IV], dwy = h[1][IW];
      const double drz = h[2][ID], dpz = h[2][IP], duz = h[2][IU], dvz = h[2][IV], dwz = h[2][IW];
      const rg_recip_t inv_r = rg_recip(r);
      const double sr0 = (-u * drx - dux * r) * dtdx + (-v * dry - dvy * r) * dtdy + (-w * drz - dwz * r) * dtdz;
      const double su0 = (-u * dux - rg_div(dpx, inv_r)) * dtdx + (-v * duy) * dtdy + (-w * duz) * dtdz;
      const double sv0 = (-u * dvx) * dtdx + (-v * dvy - rg_div(dpy, inv_r)) * dtdy + (-w * dvz) * dtdz;
      const double sw0 = (-u * dwx) * dtdx + (-v * dwy) * dtdy + (-w * dwz - rg_div(dpz, inv_r)) * dtdz;
      const double sp0 = (-u * dpx - dux * gamma * p) * dtdx + (-v * dpy - dvy * gamma * p) * dtdy + (-w * dpz - dwz * gamma * p) * dtdz;
      double tq[NV];
      tq[ID] = r + sr0; tq[IU] = u + su0; tq[IV] = v + sv0; tq[IW] = w + sw0; tq[IP] = p + sp0;
      // face states with the floors of trace.h:388-389 (hydro_face_state), grid frame
      double qmx[NV], qmy[NV];
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        qmx[n] = tq[n] + h[0][n]; qpx[n] = tq[n] - h[0][n];
        qmy[n] = tq[n] + h[1][n]; qpy[n] = tq[n] - h[1][n];
        qmz_new[n] = tq[n] + h[2][n]; qpz[n] = tq[n] - h[2][n];
      }
#define RG_FLOOR(a) a[ID] = fmax(g.smallr, a[ID]); a[IP] = fmax(g.smallp * a[ID], a[IP])
      RG_FLOOR(qmx); RG_FLOOR(qpx); RG_FLOOR(qmy); RG_FLOOR(qpy); RG_FLOOR(qmz_new); RG_FLOOR(qpz);
#undef RG_FLOOR
      if (g.grav_on) {   // uniform static gravity: predictor on the traced states, after the floors (hydro_face_state)
#define RG_GRAV(a) a[IU] += g.hgx; a[IV] += g.hgy; a[IW] += g.hgz
        RG_GRAV(qmx); RG_GRAV(qpx); RG_GRAV(qmy); RG_GRAV(qpy); RG_GRAV(qmz_new); RG_GRAV(qpz);
#undef RG_GRAV
      }
#pragma unroll
      for (int n = 0; n < NV; ++n) { L.qm[0][n][tj][ti] = qmx[n]; L.qm[1][n][tj][ti] = qmy[n]; }
    }
    __syncthreads();

    // ---- C: Riemann problems at the three low faces of cell (i,j,kk)  (hydro_flux_cell) ----
    double fx[NV], fy[NV], fz[NV];
    {
      const int tim = ti > 0 ? ti - 1 : 0, tjm = tj > 0 ? tj - 1 : 0;
      double ql[NV], qr[NV];
      // x: normal frame = grid frame
#pragma unroll
      for (int n = 0; n < NV; ++n) { ql[n] = L.qm[0][n][tj][tim]; qr[n] = qpx[n]; fx[n] = 0.0; }
      hydro_riemann<NV>(g, ql, qr, fx);
      // y: IU <-> IV
      ql[ID] = L.qm[1][ID][tjm][ti]; ql[IP] = L.qm[1][IP][tjm][ti]; ql[IU] = L.qm[1][IV][tjm][ti]; ql[IV] = L.qm[1][IU][tjm][ti]; ql[IW] = L.qm[1][IW][tjm][ti];
      qr[ID] = qpy[ID]; qr[IP] = qpy[IP]; qr[IU] = qpy[IV]; qr[IV] = qpy[IU]; qr[IW] = qpy[IW];
#pragma unroll
      for (int n = 0; n < NV; ++n) fy[n] = 0.0;
      hydro_riemann<NV>(g, ql, qr, fy);
      // z: IU <-> IW; the left state is this column's own qm_z of the previous plane
      ql[ID] = qmz[ID]; ql[IP] = qmz[IP]; ql[IU] = qmz[IW]; ql[IV] = qmz[IV]; ql[IW] = qmz[IU];
      qr[ID] = qpz[ID]; qr[IP] = qpz[IP]; qr[IU] = qpz[IW]; qr[IV] = qpz[IV]; qr[IW] = qpz[IU];
#pragma unroll
      for (int n = 0; n < NV; ++n) fz[n] = 0.0;
      hydro_riemann<NV>(g, ql, qr, fz);
#pragma unroll
      for (int n = 0; n < NV; ++n) qmz[n] = qmz_new[n];
    }

    // ---- D: cell (i,j,kk-1) is complete once the flux through its high z face is known ----
    if (own && kk - 1 >= sa) {
      if (inner2d) {
        up[ID] -= fz[ID] * dtdz; up[IP] -= fz[IP] * dtdz; up[IU] -= fz[IW] * dtdz; up[IV] -= fz[IV] * dtdz; up[IW] -= fz[IU] * dtdz;
        if (g.grav_on) {   // momentum source with the mean of the old and new density (hydro_update_cell); energy untouched
          const double rho_sum = rho_old + up[ID];
          up[IU] += g.hgx * rho_sum; up[IV] += g.hgy * rho_sum; up[IW] += g.hgz * rho_sum;
        }
        if (dslot) {
          double qn[NV];
          const double cs = hydro_prim<NV>(g, up, qn);
          inv_dt = fmax(inv_dt, (cs + fabs(qn[IU])) / g.dx + (cs + fabs(qn[IV])) / g.dy + (cs + fabs(qn[IW])) / g.dz);
        }
      }
      double* o = Uout + idx2 + (size_t)(kk - 1) * sk;
#pragma unroll
      for (int v = 0; v < NV; ++v) RG_STREAM_STORE(&o[v * N], up[v]);
    }

    // ---- E: gather the x / y fluxes of plane kk  (hydro_update_cell; the high z flux follows in D of the next iteration) ----
#pragma unroll
    for (int n = 0; n < NV; ++n) { L.f[0][n][tj][ti] = fx[n]; L.f[1][n][tj][ti] = fy[n]; }
    if (kk < sb) {   // the same barrier publishes the primitives of plane kk+1 (L.q was last read before the barrier above)
      if (ring) {
        double rq[NV];
        hydro_prim<NV>(g, urn, rq);
#pragma unroll
        for (int v = 0; v < NV; ++v) L.q[v][rtj + 1][rti + 1] = rq[v];
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) L.q[v][tj + 1][ti + 1] = qC[v];
    }
    __syncthreads();
    if (kk < sb) {
#pragma unroll
      for (int v = 0; v < NV; ++v) up[v] = uB[v];
      rho_old = uB[ID];
      if (own && inner2d) {
        const int tip = ti + 1 < TX ? ti + 1 : ti, tjp = tj + 1 < TY ? tj + 1 : tj;
#define RG_LOW_X up[ID] += fx[ID] * dtdx; up[IP] += fx[IP] * dtdx; up[IU] += fx[IU] * dtdx; up[IV] += fx[IV] * dtdx; up[IW] += fx[IW] * dtdx
#define RG_LOW_Y up[ID] += fy[ID] * dtdy; up[IP] += fy[IP] * dtdy; up[IU] += fy[IV] * dtdy; up[IV] += fy[IU] * dtdy; up[IW] += fy[IW] * dtdy
#define RG_LOW_Z up[ID] += fz[ID] * dtdz; up[IP] += fz[IP] * dtdz; up[IU] += fz[IW] * dtdz; up[IV] += fz[IV] * dtdz; up[IW] += fz[IU] * dtdz
#define RG_HIGH_X up[ID] -= L.f[0][ID][tj][tip] * dtdx; up[IP] -= L.f[0][IP][tj][tip] * dtdx; up[IU] -= L.f[0][IU][tj][tip] * dtdx; \
                  up[IV] -= L.f[0][IV][tj][tip] * dtdx; up[IW] -= L.f[0][IW][tj][tip] * dtdx
#define RG_HIGH_Y up[ID] -= L.f[1][ID][tjp][ti] * dtdy; up[IP] -= L.f[1][IP][tjp][ti] * dtdy; up[IU] -= L.f[1][IV][tjp][ti] * dtdy; \
                  up[IV] -= L.f[1][IU][tjp][ti] * dtdy; up[IW] -= L.f[1][IW][tjp][ti] * dtdy
        if (!g.dirwise_update) { RG_LOW_X; RG_LOW_Y; RG_LOW_Z; RG_HIGH_X; RG_HIGH_Y; }   // unsplitVersion 1: low faces, then high faces
        else { RG_LOW_X; RG_HIGH_X; RG_LOW_Y; RG_HIGH_Y; RG_LOW_Z; }                     // unsplitVersion 2: direction by direction
#undef RG_LOW_X
#undef RG_LOW_Y
#undef RG_LOW_Z
#undef RG_HIGH_X
#undef RG_HIGH_Y
      }
    }

    // ---- F: rotate the z pipeline ----
#pragma unroll
    for (int v = 0; v < NV; ++v) { qA[v] = qB[v]; qB[v] = qC[v]; uB[v] = uC[v]; uC[v] = uN[v]; }
    if (more) hydro_prim<NV>(g, uN, qC);
  }
  if (dslot) {   // workgroup maximum: wave64 butterfly, one LDS slot per wave (the flux buffer is free now), one atomic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) inv_dt = fmax(inv_dt, __shfl_down(inv_dt, off, 64));
    double* wmax = &L.f[0][0][0][0];
    if ((t & 63) == 0) wmax[t >> 6] = inv_dt;
    __syncthreads();
    if (t == 0) {
      double m = 0.0;
      for (int w = 0; w < NT / 64; ++w) m = fmax(m, wmax[w]);
      atomicMax(dslot, (unsigned long long)__double_as_longlong(m));
    }
  }
}

template <int TX, int TY, int SPEC, int MINW = 1>
inline int launch_hydro3d_sweep(rg_stream_t s, const DevParams& g, const double* in, double* out, double dtdx, double dtdy,
                                double dtdz, int za, int zb, unsigned long long* dslot = 0, const StepClock* clk = 0, int za2 = 0) {
  TileGrid tg;
  tg.nbx = (g.isize - 1 + (TX - 2) - 1) / (TX - 2);   // owners cover i in [1, nbx*(TX-2)] plus column 0
  tg.nby = (g.jsize - 1 + (TY - 2) - 1) / (TY - 2);
  const int span = zb - za;
  const int zseg_env = rgpu::options().zseg;
  // two workgroups are resident per CU (~200 VGPRs): 64 per XCD; a segment costs two extra iterations (pipeline fill)
  const bool pair = za2 > 0;   // a second range [za2, za2 + span) in the same launch
  tile_grid_plan(tg, span, 64, 12, 2, zseg_env, pair);
  if (pair) { tg.zsplit = za + span; tg.zgap = za2 - (za + span); }
  hipLaunchKernelGGL((hydro3d_sweep_kernel<TX, TY, SPEC, MINW>), dim3(8u * (unsigned)tg.per_xcd), dim3(TX * TY), 0, s, g, tg, in, out,
                     dtdx, dtdy, dtdz, za, pair ? zb + span : zb, dslot, clk);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// configurations the fused sweep covers (the per-cell gravity field is excluded by the caller: it knows the step's setting)
inline bool hydro3d_sweep_covers(const DevParams& g) { return tiled_enabled() && g.three_d && !g.mhd && g.nvar == 5; }

// Complete the update of planes [a,b) of a 3D hydro step.  Returns 0 = done, 1 = not applicable (the caller runs the
// flat kernels), < 0 = launch error.
// dslot: device slot for the CFL maximum of the new state (reset by the caller), or 0
inline int hydro3d_sweep(rg_stream_t s, const DevParams& g, const double* in, double* out, double dtdx, double dtdy,
                         double dtdz, int a, int b, unsigned long long* dslot = 0, const StepClock* clk = 0, int a2 = 0, int b2 = 0) {
  if (!hydro3d_sweep_covers(g) || g.grav_on == 2) return 1;   // per-cell gravity field: flat kernels
  // ghost planes inside a range: plain copy, like the flat update kernel
  const K_copy_cells kc = {in, out, g.ncell, 5, clk};
  int za[2], zb[2];
  for (int n = 0; n < 2; ++n) {
    const int lo_ = n ? a2 : a, hi_ = n ? b2 : b;
    za[n] = zb[n] = 0;
    if (hi_ <= lo_) continue;
    za[n] = lo_ < g.gw ? g.gw : lo_; zb[n] = hi_ > g.ksize - g.gw ? g.ksize - g.gw : hi_;
    if (lo_ < g.gw && rgpu::rg_launch_range<256>(s, (unsigned)lo_ * g.sk, (unsigned)((hi_ < g.gw ? hi_ : g.gw) - lo_) * g.sk, kc)) return -1;
    if (hi_ > g.ksize - g.gw) {
      const int lo = lo_ > g.ksize - g.gw ? lo_ : g.ksize - g.gw;
      if (rgpu::rg_launch_range<256>(s, (unsigned)lo * g.sk, (unsigned)(hi_ - lo) * g.sk, kc)) return -1;
    }
  }
  // thread tile: 16 x 16 measured best at 256^3 (sweep 1.18 ms; 32 x 8: 1.24, 32 x 16: 1.19-1.26, 64 x 8: 1.43, 64 x 4: 1.75;
  // a 168-VGPR build for three workgroups per CU: 1.46) -- the squarest tile recomputes the least halo (196 of 256 threads
  // update a cell)
  constexpr int TX = 16, TY = 16;
  const bool no_spec = !rgpu::options().spec;
  // planes [lo, hi) and, second > 0, a second range of the same length starting there, in one launch (TileGrid::zsplit)
  auto launch = [&](int lo, int hi, int second) -> int {
    if (!no_spec) {
      const int SL1 = SPEC_SLOPE1 | SPEC_NO_GRAVITY, SL2 = SPEC_SLOPE2 | SPEC_NO_GRAVITY;
#define RG_TRY(SP) if (spec_matches(SP, g)) return launch_hydro3d_sweep<TX, TY, SP>(s, g, in, out, dtdx, dtdy, dtdz, lo, hi, dslot, clk, second)
      RG_TRY(SPEC_HYDRO_HLLC | SL2); RG_TRY(SPEC_HYDRO_HLLC | SL1);
      RG_TRY(SPEC_HYDRO_APPROX | SL2); RG_TRY(SPEC_HYDRO_APPROX | SL1);
      RG_TRY(SPEC_HYDRO_HLL | SL2); RG_TRY(SPEC_HYDRO_HLL | SL1);
      // with uniform gravity (g.grav_on == 1)
      RG_TRY(SPEC_HYDRO_HLLC | SPEC_SLOPE2); RG_TRY(SPEC_HYDRO_HLLC | SPEC_SLOPE1);
      RG_TRY(SPEC_HYDRO_APPROX | SPEC_SLOPE2); RG_TRY(SPEC_HYDRO_APPROX | SPEC_SLOPE1);
#undef RG_TRY
    }
    return launch_hydro3d_sweep<TX, TY, SPEC_NONE>(s, g, in, out, dtdx, dtdy, dtdz, lo, hi, dslot, clk, second);
  };
  const bool one = zb[0] > za[0], two = zb[1] > za[1];
  if (one && two && zb[1] - za[1] == zb[0] - za[0] && za[1] >= zb[0]) return launch(za[0], zb[0], za[1]);   // the two boundary ranges of a slab
  if (one) { const int rc = launch(za[0], zb[0], 0); if (rc) return rc; }
  if (two) return launch(za[1], zb[1], 0);
  return 0;
}

}  // namespace rgpu_tiled
