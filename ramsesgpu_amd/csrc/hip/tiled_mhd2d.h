// tiled_mhd2d.h (HIP / gfx950 only) -- the whole 2D MHD unsplit step (primitives, slopes + CTU trace, two HLLD face problems and
// the 2D-HLLD corner problem per cell, conservative + CT update, CFL term of the new state) as ONE LDS-tiled kernel: U -> Unew.
//
// Flat pipeline (kernels_mhd2d.h): K_mhd_prim, K_mhd_trace2d, K_mhd_flux2d, K_mhd_update2d -- four launches that pass Q (8),
// T (26) and F (13 doubles per cell) through L2 / HBM; at the shipped 2D sizes (Orszag-Tang 512^2) the step is a chain of
// 10-35 us kernels.  Reference idiom: the one-kernel 2D step with a shared-memory tile and a ghost overlap,
// godunov_unsplit_mhd.cuh:274 (kernel_godunov_unsplit_mhd_2d_v1).
//
// A 256-thread workgroup (4 waves; two workgroups per CU = 2 waves per SIMD at 256 VGPRs) owns the Riemann problems of a
// 16 x 8 block of cells -- exactly two 64-lane passes per problem type -- and finishes the 15 x 7 cells whose four faces and
// four corners lie inside it:
//   phase 0  all threads   U of the 20 x 12 input cells -> registers; face field Bx, By -> LDS; primitives (19 x 11 cells) -> LDS
//   phase 1  waves 0..2    slopes + trace of the 17 x 9 traced cells -> T (26 x 153 doubles) in LDS
//   phase 2  waves 0, 1    corner EMF of 64 cells each (2D HLLD: ~970 VALU instructions)
//            waves 2, 3    x / y face flux of all 128 cells (HLLD: two passes of ~490)         -> F (13 x 128) in LDS, over Q
//   phase 3  waves 0, 1    update of the 105 cells (+ CT, + the cell's CFL term into the context's device slots)
// Four barriers, 54 KB of LDS.  Neither Q nor T nor F reaches global memory.
//
// Arithmetic: the device functions of kernels_mhd2d.h (mhd_prim, mhd_trace2d_at, mhd_flux2d_at, mhd_update2d_at,
// mhd_invdt2d_new) instantiated with LDS accessors -- same expressions, same operand order, same bits as the flat kernels.
#pragma once
#include "step_clock.h"

namespace rgpu_tiled {

constexpr int M2_FX = 16, M2_FY = 8;                 // cells whose Riemann problems a workgroup solves
constexpr int M2_OX = M2_FX - 1, M2_OY = M2_FY - 1;  // cells it finishes
constexpr int M2_TX = M2_FX + 1, M2_TY = M2_FY + 1;  // traced cells: the problem cells + one low-side layer, origin (i0-1, j0-1)
constexpr int M2_TCELLS = M2_TX * M2_TY;             // 153
constexpr int M2_IX = M2_FX + 4, M2_IY = M2_FY + 4;  // input cells (U, face field, primitives), origin (i0-2, j0-2): 20 x 12
constexpr int M2_ICELLS = M2_IX * M2_IY;             // 240
constexpr int M2_FCELLS = M2_FX * M2_FY;             // 128
constexpr int M2_THREADS = 256;
static_assert(M2_ICELLS <= M2_THREADS, "one input cell per thread");
static_assert(F2_COUNT * M2_FCELLS <= 8 * M2_ICELLS, "F reuses the LDS of Q");

struct Trace2dInLds {   // primitives and face field of the input tile, one index space (row stride M2_IX)
  const double* Q; const double* A; const double* B;
  RG_DEVFN double q(int v, unsigned m) const { return Q[v * M2_ICELLS + (int)m]; }
  RG_DEVFN double ua(unsigned m) const { return A[m]; }
  RG_DEVFN double ub(unsigned m) const { return B[m]; }
  RG_DEVFN unsigned sj() const { return (unsigned)M2_IX; }
};
struct T2LdsWrite {
  double* cell;
  RG_DEVFN void put(int slot, double v) const { cell[slot * M2_TCELLS] = v; }
};
struct T2LdsRead {
  const double* T;
  RG_DEVFN double get(int slot, unsigned m) const { return T[slot * M2_TCELLS + (int)m]; }
  RG_DEVFN unsigned sj() const { return (unsigned)M2_TX; }
};
struct F2LdsWrite {
  double* cell;
  RG_DEVFN void put(int comp, double v) const { cell[comp * M2_FCELLS] = v; }
};
struct F2LdsRead {
  const double* F;
  RG_DEVFN double get(int comp, unsigned m) const { return F[comp * M2_FCELLS + (int)m]; }
  RG_DEVFN unsigned sj() const { return (unsigned)M2_FX; }
};

template <int SPEC>
__global__ void __launch_bounds__(M2_THREADS, 2) mhd2d_step_kernel(DevParams g, RotCoef rc, int nbx, const double* __restrict__ U,
                                                                   double* __restrict__ Unew, double dt, double dtdx, double dtdy,
                                                                   unsigned long long* dt_slots, int images, const StepClock* clk) {
  spec_assume<SPEC>(g);
  if (clk) {   // the time step lives on the device (hip/step_clock.h): a batch of steps queued without a host round trip
    if (clk->stop) return;
    dt = clk->dt; dtdx = clk->dtdx; dtdy = clk->dtdy;
  }
  __shared__ double LQ[8 * M2_ICELLS];          // primitives of the input tile; from phase 2 on: the fluxes F (13 x 128)
  __shared__ double LA[M2_ICELLS], LB[M2_ICELLS];   // face field Bx, By of the input tile
  __shared__ double LT[T2_COUNT * M2_TCELLS];   // compact traced state

  const int t = (int)threadIdx.x;
  const int by = (int)blockIdx.x / nbx, bx = (int)blockIdx.x - by * nbx;
  const int gw = g.gw;
  const int i0 = gw + bx * M2_OX, j0 = gw + by * M2_OY;   // first cell of the tile
  const size_t N = g.ncell;
  const unsigned sj = g.sj;

  // ---- phase 0: input tile.  Thread t owns input cell (t % 20, t / 20) = global (i0 - 2 + ux, j0 - 2 + uy) ----
  {
    const int uy = t / M2_IX, ux = t - uy * M2_IX;
    const int gi = i0 - 2 + ux, gj = j0 - 2 + uy;
    const bool in_array = t < M2_ICELLS && gi < g.isize && gj < g.jsize;
    double u[8];
#pragma unroll
    for (int v = 0; v < 8; ++v) u[v] = 0.0;
    if (in_array) {
      const double* p = U + (size_t)gi + (size_t)gj * sj;
#pragma unroll
      for (int v = 0; v < 8; ++v) u[v] = p[(size_t)v * N];
    }
    if (t < M2_ICELLS) { LA[t] = u[IA]; LB[t] = u[IB]; }
    __syncthreads();
    // primitives where the reference computes them (mhd_prim_cell: i < isize - 1, j < jsize - 1) and the tile has the +1 faces
    if (in_array && ux < M2_IX - 1 && uy < M2_IY - 1 && gi < g.isize - 1 && gj < g.jsize - 1) {
      const Prim8 q = mhd_prim(g, u, LA[t + 1], LB[t + M2_IX], 0.0, dt);
      LQ[ID * M2_ICELLS + t] = q.r; LQ[IP * M2_ICELLS + t] = q.p; LQ[IU * M2_ICELLS + t] = q.u; LQ[IV * M2_ICELLS + t] = q.v;
      LQ[IW * M2_ICELLS + t] = q.w; LQ[IA * M2_ICELLS + t] = q.a; LQ[IB * M2_ICELLS + t] = q.b; LQ[IC * M2_ICELLS + t] = q.c;
    }
  }
  __syncthreads();

  // ---- phase 1: trace of the 17 x 9 traced cells, origin (i0 - 1, j0 - 1) ----
  if (t < M2_TCELLS) {
    const int ty = t / M2_TX, tx = t - ty * M2_TX;
    const int ti = i0 - 1 + tx, tj = j0 - 1 + ty;
    if (ti <= g.isize - gw && tj <= g.jsize - gw) {   // trace range of mhd_trace2d_cell (low bounds hold by construction)
      const Trace2dInLds in = {LQ, LA, LB};
      const T2LdsWrite tw = {LT + t};
      const double xPos = g.xMin + g.dx / 2 + (ti - gw) * g.dx;
      mhd_trace2d_at(g, in, tw, dtdx, dtdy, (unsigned)((ty + 1) * M2_IX + tx + 1), xPos);
    }
  }
  __syncthreads();

  // ---- phase 2: Riemann problems of the 16 x 8 problem cells; results into LF (the LDS of Q, dead since the trace) ----
  double* LF = LQ;
  {
    const int wave = t >> 6, lane = t & 63;
    const T2LdsRead ta = {LT};
    // waves 0, 1: corner problem of cells [64 wave, +64); waves 2, 3: x (wave 2) / y (wave 3) face problem of all 128 cells
    for (int pass = 0; pass < (wave < 2 ? 1 : 2); ++pass) {
      const int fc = (wave < 2 ? wave : pass) * 64 + lane;
      const int fy = fc / M2_FX, fx = fc - fy * M2_FX;
      const int ci = i0 + fx, cj = j0 + fy;
      const F2LdsWrite fw = {LF + fc};
      if (ci <= g.isize - gw && cj <= g.jsize - gw) {   // flux range of mhd_flux2d_cell
        const unsigned m = (unsigned)((fy + 1) * M2_TX + fx + 1);
        const unsigned gm = (unsigned)ci + (unsigned)cj * sj;
        const double xPos = g.xMin + g.dx / 2 + (ci - gw) * g.dx;
        if (wave < 2) mhd_flux2d_at<DO2_EMF, false>(g, ta, fw, m, gm, xPos);
        else if (wave == 2) mhd_flux2d_at<DO2_FX, false>(g, ta, fw, m, gm, xPos);
        else mhd_flux2d_at<DO2_FY, false>(g, ta, fw, m, gm, xPos);
      } else {
        // outside the range the flat kernels leave the (zero-initialised) flux array alone: the CT update of the first high
        // ghost layer reads such a zero
        if (wave < 2) fw.put(F2_EMF, 0.0);
        else {
          const int base = (wave == 2) ? F2_X : F2_Y;
#pragma unroll
          for (int v = 0; v < 6; ++v) fw.put(base + v, 0.0);
        }
      }
    }
  }
  __syncthreads();

  // ---- phase 3: the 15 x 7 cells of the tile: conservative + CT update, CFL term of the new state ----
  if (t < 128) {   // waves 0, 1 (whole waves: the slot maximum is formed wave-wide)
    const int oy = t / M2_OX, ox = t - oy * M2_OX;
    const int ci = i0 + ox, cj = j0 + oy;
    const bool mine = t < M2_OX * M2_OY;
    const bool in_i = ci < g.isize - gw, in_j = cj < g.jsize - gw;   // (ci >= gw, cj >= gw by construction)
    const bool ct = mine && ci <= g.isize - gw && cj <= g.jsize - gw;
    const bool interior = mine && in_i && in_j;
    double inv = 0.0;
    if (ct) {
      const unsigned gm = (unsigned)ci + (unsigned)cj * sj;
      const unsigned fm = (unsigned)(oy * M2_FX + ox);
      double u[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) u[v] = U[gm + (size_t)v * N];
      const F2LdsRead fa = {LF};
      mhd_update2d_at<false>(g, rc, fa, u, u[ID], dt, dtdx, dtdy, fm, gm, interior, true);
      if (dt_slots && interior) inv = mhd_invdt2d_new(g, fa, u, U[gm + 1 + (size_t)IA * N], U[gm + sj + (size_t)IB * N], dtdx, dtdy, fm);
      if (!images) {
#pragma unroll
        for (int v = 0; v < 8; ++v) Unew[gm + (size_t)v * N] = u[v];
      } else if (interior) {
        // all four faces periodic: the cell also writes its periodic images into the ghost cells (what the next step's ghost fill
        // would copy there: the same doubles), so that fill is not launched.  Every ghost cell -- the CT layer included, whose
        // own value the fill overwrites -- is the image of exactly one interior cell: one writer per location.
        const int nx = g.nx, ny = g.ny;
        const int xi[3] = {ci, ci + nx, ci - nx}, yj[3] = {cj, cj + ny, cj - ny};
        const bool xok[3] = {true, ci < 2 * gw, ci >= nx}, yok[3] = {true, cj < 2 * gw, cj >= ny};
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int a = 0; a < 3; ++a)
            if (xok[a] && yok[b]) {
              double* o = Unew + (size_t)xi[a] + (size_t)yj[b] * sj;
#pragma unroll
              for (int v = 0; v < 8; ++v) o[(size_t)v * N] = u[v];
            }
      }
    }
    if (dt_slots) rgpu::rg_slot_max_wave(dt_slots + (((unsigned)blockIdx.x * 2u + (unsigned)(t >> 6)) & (rgpu::RG_DT_SLOTS - 1)), inv);
  }
}

// Configurations the fused 2D step covers: no per-cell gravity field (its own instantiations) and no Dirichlet face (its ghost
// fill leaves B untouched, so the ghost cells of the output must be copies of the input's: the flat update kernel copies every
// cell it does not update, this kernel writes the cells it owns only -- with periodic / Neumann faces every ghost cell is
// rewritten by the next ghost fill before anything reads it).  Returns 0 = done, 1 = not covered, < 0 = launch error.
inline bool mhd2d_step_covers(const DevParams& g) { return tiled_enabled() && !g.three_d && g.mhd && g.grav_on != 2; }

template <int SPEC_PLAIN>
// images != 0 (caller: all four faces periodic, nx, ny >= ghost width, nothing modifies the new state after this kernel): the
// interior cells also write their periodic images, i.e. the output's ghost cells are valid on return
inline int mhd2d_step(rg_stream_t s, const DevParams& g, const RotCoef& rc, bool spec_plain, const double* U, double* Unew, double dt,
                      unsigned long long* dt_slots, int images, const StepClock* clk = 0) {
  if (!mhd2d_step_covers(g)) return 1;
  const int nbx = (g.isize - 2 * g.gw + 1 + M2_OX - 1) / M2_OX;   // cells gw .. isize-gw (the CT layer included)
  const int nby = (g.jsize - 2 * g.gw + 1 + M2_OY - 1) / M2_OY;
  const double dtdx = dt / g.dx, dtdy = dt / g.dy;
  if (spec_plain)
    hipLaunchKernelGGL((mhd2d_step_kernel<SPEC_PLAIN>), dim3((unsigned)(nbx * nby)), dim3(M2_THREADS), 0, s, g, rc, nbx, U, Unew, dt, dtdx, dtdy, dt_slots, images, clk);
  else
    hipLaunchKernelGGL((mhd2d_step_kernel<SPEC_NONE>), dim3((unsigned)(nbx * nby)), dim3(M2_THREADS), 0, s, g, rc, nbx, U, Unew, dt, dtdx, dtdy, dt_slots, images, clk);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace rgpu_tiled
