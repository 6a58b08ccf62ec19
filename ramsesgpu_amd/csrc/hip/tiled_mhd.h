// tiled_mhd.h (HIP / gfx950 only) -- primitives, edge electric field, trace and Riemann problems of the 3D MHD unsplit
// step as ONE cooperative, LDS-tiled, z-marching kernel: U -> F, emf.  Neither the primitive variables Q, nor the electric
// field E, nor the 38-double compact traced state T ever reach HBM.
//
// Flat pipeline (kernels_mhd3d.h): K_mhd_prim writes Q, K_mhd_elec writes E, K_mhd_trace3d writes T (304 B/cell),
// K_mhd_flux3d reads T back with its 2x2x2 neighbourhood -- 120 of the step's 155 GB of HBM traffic at 512^3.  Here a
// 512-thread workgroup (8 waves, 2 per SIMD: the edge solvers need ~250 VGPRs) owns a tile of 16 x 8 cells and marches
// along z.  Wave roles (waves w and w + 4 share a SIMD):
//   waves 0,1,2 and 4,5,6   Riemann problems of direction d = w & 3 for the cells [64 (w >> 2), +64) of the tile: edge
//                           EMF along d (2D HLLD, ~1100 VALU instructions) + flux through the low d face (HLLD, ~470)
//   waves 3 and 7           producers: primitives of plane kk+3 (from U, prefetched into registers), trace of plane kk+1
// Iteration kk:
//   Riemann waves  problems of plane kk from T(kk) in LDS and the states carried in registers from T(kk-1)
//   producers      issue the loads of U(kk+3); trace(kk+1) -> the other T buffer; prim(kk+3); [pair rendezvous] store
//                  Q / B (kk+3) into the LDS slot plane kk vacated; [pair rendezvous] edge electric field of plane kk+3
//                  from Q / B (kk+2, kk+3) -> the slot E(kk+1) vacated.  Q, B, E are private to the producer pair, which
//                  synchronises through an LDS counter
//   barrier        ONE workgroup barrier per plane: T(kk+1) complete, T(kk) free
// LDS: T 2 x 46.5 KB, Q / B 3 x 18.4 KB, E 2 x 5 KB = 158 KB (one workgroup per CU).
// (Reference idiom: the shared-memory staging of trace_v4 / flux_update_hydro_v4, godunov_unsplit_mhd.cuh:3260, 4595.)
//
// Arithmetic: the same device functions as the flat kernels (mhd_prim, mhd_elec_comp, mhd_trace3d_at, face_state3d,
// edge_state3d, edge_emf, mhd_face_flux), instantiated with LDS accessors instead of the global-array ones -- same
// expressions, same operand order, same bits.
#pragma once
#include "tiled_hydro.h"

namespace rgpu_tiled {

// Tile geometry: OX x OY cells whose Riemann problems a workgroup solves per plane, and what follows from it.
//   MhMain   16 x 8: 128 cells = two waves per direction -- the tiles of the sweep
//   MhLastX  2 x 32, at the LAST x face column i = isize - gw alone (round 5).  513 face columns in tiles of 16 are 33 tile columns, the
//            last with one valid column: 64 of the 2112 tile marches of the 512^3 box, 3.1 % of the sweep for 0.2 % of its faces (a
//            periodic layer is copied instead, K_copy_periodic_layer; the shearing-box x faces have no image to copy).  The same kernel
//            with this geometry marches that column in 16 tiles of 32 rows (column 1 of the tile lies outside the face range; 64 cells =
//            one wave per direction, the waves of the second half idle): a quarter of the marches, and short ones (launch_mhd3d_sweep).
template <int OX_, int OY_, bool LASTX_>
struct MhTile {
  static constexpr int OX = OX_, OY = OY_;              // cells whose Riemann problems a workgroup solves per plane
  static constexpr bool LASTX = LASTX_;                 // the tile column sits at i0 = isize - gw
  static constexpr int PX = OX + 1, PY = OY + 1;
  static constexpr int CELLS = PX * PY;                 // traced cells per plane (tile + low-side halo): 153
  // LDS layout (round 6): CELL-MAJOR records.  DS instructions take unsigned immediate offsets only (16 bits; 8 bits x 8 bytes each for
  // the two halves of a ds_read2_b64), so with the variable-major layout of rounds 2-5 every access below or far above "the" cell of a
  // thread cost a v_add_u32 for its address: 15 % of the sweep's VALU instructions were 32-bit integer work.  With records, everything a
  // thread reads lies within 2040 bytes above one of a few per-thread row bases (rg_opaque), and neighbouring slots pair up into
  // ds_read2_b64 / ds_write2_b64 without any address arithmetic.  TREC is odd x 8 bytes: lanes of consecutive cells hit distinct banks.
  static constexpr int TREC = T_COUNT + 1;              // doubles per traced-cell record of T (38 used)
  static constexpr int BUF = TREC * CELLS;              // doubles per plane buffer of T
  static constexpr int QX = OX + 3, QY = OY + 3;
  static constexpr int QCELLS = QX * QY;                // input cells per plane (traced cells +- 1), origin (i0-2, j0-2): 209
  static constexpr int NQB = 11;                        // per input cell: 8 primitives + 3 face-field components
  static constexpr int QBSLOT = NQB * QCELLS;           // doubles per plane slot of Q / B
  static constexpr int ESLOT = 3 * QCELLS;              // doubles per plane slot of E (same cell geometry as Q)
  static constexpr int SX = OX, SY = OY;                // tile pitch = tile size: every Riemann problem is solved by exactly one tile
};
typedef MhTile<16, 8, false> MhMain;
typedef MhTile<2, 32, true> MhLastX;
constexpr int MH_THREADS = 512;
// ONE structure of the kernel for both arithmetics (round 6; rounds 3-5 shipped a one-loop form with per-plane decodes and a serial
// prologue for the contracted build, whose register file was full at 256 VGPRs): per-role main loops, the loads of the three prologue
// planes in flight at once, loop-invariant decodes kept in registers.  Same-box A/B of the alternatives on the round-6 kernel (205 / 213
// VGPRs), 512^3 sweep, exact | contracted: one loop for all waves 30.4-30.7 | 24.03-24.08 against 30.05-30.10 | 23.58-23.64; serial
// prologue 30.25-30.37 | 24.03-24.08 against 30.05-30.10 | 23.13-23.26; per-plane decodes 30.4 | 24.0 against 30.05 | 24.0
// (profiles/r06_sweep_structure_ab.txt).
// a copy of an integer that the optimiser cannot fold constants into or derive from another value: the LDS addresses formed from
// it are "this register + a non-negative immediate"
RG_DEVFN unsigned rg_opaque(unsigned x) { asm("" : "+v"(x)); return x; }
// a wave-uniform value the compiler must take afresh where this stands (volatile: not hoisted out of the z loop): what is derived from
// it -- the component offsets of the flux stores, ten scalar registers per direction -- is recomputed by a few SALU instructions at the
// end of every solve instead of living in (spilled) scalar registers across it
RG_DEVFN size_t rg_fresh(size_t x) { asm volatile("" : "+s"(x)); return x; }

template <class G>
struct TLdsWrite {   // bound to the record of one traced cell
  double* cell;
  RG_DEVFN void put(int slot, double v) const { cell[slot] = v; }
};
// Trace inputs staged in LDS: planes kk-1, kk, kk+1 of Q / B (records of NQB doubles per input cell), planes kk, kk+1 of E (records
// of 3).  The cell index m the numerics pass is RELATIVE to the thread's 3 x 3 input neighbourhood: m = row * QX + col, centre =
// QX + 1 (what mhd_trace3d_at / mhd_elec_comp receive), m - 1, m + sj() ... its neighbours -- compile-time constants after inlining.
// qA[dz + 1] / qB[dz + 1]: byte offsets in LQ of the records of the neighbourhood's cells (row 0, col 0) / (row 2, col 0) of plane
// dz; eA[dz]: of the E record of cell (row 1, col 0).
template <class G>
struct TraceInLds {
  const char* lq; const char* le;
  unsigned qA[3], qB[3], eA[2];
  const int* flag; int want;   // E of plane kk+1 is complete once *flag >= want (written by the Riemann waves)
  RG_DEVFN void e_ready() const {
    if (flag) {
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  RG_DEVFN double q(int v, int dz, unsigned m) const {
    const unsigned row = m / (unsigned)G::QX, col = m - row * (unsigned)G::QX;
    const unsigned rec = (row >= 2u) ? qB[dz + 1] + (row - 2u) * (unsigned)(G::QX * G::NQB * 8) : qA[dz + 1] + row * (unsigned)(G::QX * G::NQB * 8);
    return *reinterpret_cast<const double*>(lq + rec + (col * (unsigned)G::NQB + (unsigned)v) * 8u);
  }
  RG_DEVFN double bf(int comp, int dz, unsigned m) const { return q(8 + comp, dz, m); }
  RG_DEVFN double e(int comp, int dz, unsigned m) const {   // m >= QX: rows 1 and 2 of the neighbourhood
    return *reinterpret_cast<const double*>(le + eA[dz] + ((m - (unsigned)G::QX) * 3u + (unsigned)comp) * 8u);
  }
  RG_DEVFN unsigned sj() const { return (unsigned)G::QX; }
};

// T accessor of ONE plane buffer: the +z neighbour of a cell is not in it.  stride(ZD) = 0 makes the one component of a
// state that reads the plane above (the face field on the + side) read this plane instead; the caller replaces that
// component when the plane above has been traced (see "carried states" below).
// The cell index m is RELATIVE to the 2 x 2 traced cells a Riemann thread reads: m = row * PX + col, the thread's own cell = PX + 1
// (what riemann_dir passes), m - 1, m - PX, m - PX - 1 its low-side neighbours.  row0 / row1: byte offsets in this plane's buffer of
// the records of (row 0, col 0) and (row 1, col 0).
template <class G>
struct TLdsPlane {
  const char* lt; unsigned row0, row1;
  RG_DEVFN double get(int slot, unsigned m) const {
    const unsigned row = m / (unsigned)G::PX, col = m - row * (unsigned)G::PX;
    return *reinterpret_cast<const double*>(lt + (row ? row1 + (row - 1u) * (unsigned)(G::PX * G::TREC * 8) : row0) + (col * (unsigned)G::TREC + (unsigned)slot) * 8u);
  }
  RG_DEVFN unsigned stride(int D) const { return (D == XD) ? 1u : (D == YD) ? (unsigned)G::PX : 0u; }
};

// Riemann problems of one direction d at cell m of plane kk: the edge EMF along d and the flux through the low d face.
// Tk = traced state of plane kk.  States that belong to plane kk-1 (the two upper edge states of the x / y edges, the
// left state of the z face) were built one iteration earlier from T(kk-1) -- except their one component that lives on
// plane kk, which is filled in here -- and are carried in registers (c0, c1): T(kk-1) need not stay in LDS.
// After solving, the states plane kk contributes to iteration kk+1 are built into c0, c1.
// prio_drop: the wave entered with raised priority (s_setprio 1) and gives it up after the EMF, i.e. after ~70 % of its
// work.  The two Riemann waves of a SIMD are arbitrated oldest-first: left alone the older one finishes at ~60 % of the
// phase and the younger one runs the rest alone at single-wave issue efficiency; with the younger wave favoured for the
// first 70 % of its work both finish together.
template <class G, int DIR>
RG_DEVFN void riemann_dir(const DevParams& g, const TLdsPlane<G>& Tk, unsigned m, double xPos, double* __restrict__ F,
                          double* __restrict__ emf, unsigned idx, Prim8& c0, Prim8& c1, bool solve, bool prio_drop) {
  // idx = flux_index of the cell (where its fluxes / EMFs go; the traced states take no global index here: no gravity field)
  const unsigned sx = 1u, sj = (unsigned)G::PX;
  if (DIR == XD) {   // edge along x: t1 = y, t2 = z.  rt = (+,+) from c-y-z, rb = (+,-) from c-y, lt = (-,+) from c-z, lb = (-,-) from c
    if (solve) {
      c0.b = Tk.get(T_CL, m - sj) + Tk.get(T_DCLY, m - sj);        // b2 = CL(m2) + s1 * dCLy(m2), s1 = +1, m2 = the cell above
      c1.b = Tk.get(T_CL, m) + (-1.0) * Tk.get(T_DCLY, m);         // s1 = -1
      const Prim8 rb = edge_state3d<0, +1, -1, false>(g, Tk, m - sj, idx), lb = edge_state3d<0, -1, -1, false>(g, Tk, m, idx);
      RG_STREAM_STORE(&emf[idx + (size_t)EMF_X * rg_fresh(g.fN)], edge_emf<0>(g, c0, rb, c1, lb, xPos));
      if (prio_drop) __builtin_amdgcn_s_setprio(0);
      Prim8 L = face_state3d<XD, +1, false>(g, Tk, m - sx, idx), R = face_state3d<XD, -1, false>(g, Tk, m, idx);
      double fl[8];
      mhd_face_flux<XD>(g, L, R, xPos, fl);
      store_flux<XD>(F, rg_fresh(g.fN), idx, fl);
    }
    c0 = edge_state3d<0, +1, +1, false>(g, Tk, m - sj, idx);
    c1 = edge_state3d<0, -1, +1, false>(g, Tk, m, idx);
  } else if (DIR == YD) {   // edge along y: t1 = z, t2 = x.  rt = (+,+) from c-z-x, rb = (+,-) from c-z, lt = (-,+) from c-x, lb from c
    if (solve) {
      c0.a = Tk.get(T_CL, m - sx) + Tk.get(T_DCLX, m - sx);        // b1 = CL(m1) + s2 * dCLx(m1), s2 = +1
      c1.a = Tk.get(T_CL, m) + (-1.0) * Tk.get(T_DCLX, m);
      const Prim8 lt = edge_state3d<1, -1, +1, false>(g, Tk, m - sx, idx), lb = edge_state3d<1, -1, -1, false>(g, Tk, m, idx);
      RG_STREAM_STORE(&emf[idx + (size_t)EMF_Y * rg_fresh(g.fN)], edge_emf<1>(g, c0, c1, lt, lb, xPos));
      if (prio_drop) __builtin_amdgcn_s_setprio(0);
      Prim8 L = face_state3d<YD, +1, false>(g, Tk, m - sj, idx), R = face_state3d<YD, -1, false>(g, Tk, m, idx);
      double fl[8];
      mhd_face_flux<YD>(g, L, R, xPos, fl);
      store_flux<YD>(F, rg_fresh(g.fN), idx, fl);
    }
    c0 = edge_state3d<1, +1, +1, false>(g, Tk, m - sx, idx);
    c1 = edge_state3d<1, +1, -1, false>(g, Tk, m, idx);
  } else {   // edge along z: all four states on plane kk; z face: left state from plane kk-1
    if (solve) {
      const Prim8 rt = edge_state3d<2, +1, +1, false>(g, Tk, m - sx - sj, idx), rb = edge_state3d<2, +1, -1, false>(g, Tk, m - sx, idx);
      const Prim8 lt = edge_state3d<2, -1, +1, false>(g, Tk, m - sj, idx), lb = edge_state3d<2, -1, -1, false>(g, Tk, m, idx);
      RG_STREAM_STORE(&emf[idx + (size_t)EMF_Z * rg_fresh(g.fN)], edge_emf<2>(g, rt, rb, lt, lb, xPos));
      if (prio_drop) __builtin_amdgcn_s_setprio(0);
      c0.a = Tk.get(T_CL, m);                                      // bn of the left state: the face it shares with cell m
      Prim8 R = face_state3d<ZD, -1, false>(g, Tk, m, idx);
      double fl[8];
      mhd_face_flux<ZD>(g, c0, R, xPos, fl);
      store_flux<ZD>(F, rg_fresh(g.fN), idx, fl);
    }
    c0 = face_state3d<ZD, +1, false>(g, Tk, m, idx);
  }
}

template <int SPEC, class G>
__global__ void __launch_bounds__(MH_THREADS) mhd3d_sweep_kernel(DevParams g, TileGrid tg, const double* __restrict__ U,
                                                               double* __restrict__ F, double* __restrict__ emf,
                                                               double dt, double dtdx, double dtdy, double dtdz, int ra, int rb,
                                                               const StepClock* clk) {
  spec_assume<SPEC>(g);
  constexpr int MH_OX = G::OX, MH_OY = G::OY, MH_PX = G::PX, MH_PY = G::PY, MH_CELLS = G::CELLS, MH_BUF = G::BUF, MH_QX = G::QX, MH_QCELLS = G::QCELLS,
                MH_NQB = G::NQB, MH_QBSLOT = G::QBSLOT, MH_ESLOT = G::ESLOT, MH_SX = G::SX, MH_SY = G::SY;
  if (clk) {   // the time step lives on the device (csrc/step_clock_rec.h): a batch of steps queued without a host round trip
    if (clk->stop) return;
    dt = clk->dt; dtdx = clk->dtdx; dtdy = clk->dtdy; dtdz = clk->dtdz;
  }
  __shared__ __attribute__((aligned(16))) double LT[2 * MH_BUF];          // T of planes kk (read) and kk+1 (written)   (buffer = plane & 1)
  __shared__ double LQ[3 * MH_QBSLOT];       // Q / B of planes kk .. kk+2                  (slot = plane % 3)
  __shared__ double LE[2 * MH_ESLOT];        // E of planes kk+1, kk+2                      (slot = plane & 1)
  __shared__ int Lsync;                      // arrival counter of the producer pair
  __shared__ int Lesync;                     // arrival counter of the waves that compute E: planes completed x E_NWAVES

  const TileItem item = tile_item(tg, (int)blockIdx.x, ra, rb);   // this workgroup's tile and its planes [sa, sb) of [ra, rb)
  if (!item.valid) return;
  const int bx = item.bx, by = item.by, sa = item.sa, sb = item.sb;

  const int gw = g.gw;
  const int i0 = G::LASTX ? g.isize - gw : gw + bx * MH_SX, j0 = gw + by * MH_SY;   // first cell of the tile
  const int t = (int)threadIdx.x;
  const size_t N = g.ncell;
  const unsigned sk = g.sk;
  const int wave = t >> 6, lane = t & 63;
  // the role of a wave is a function of its index alone: wave-uniform, fixed for the life of the workgroup (the per-role main
  // loops below rely on it: every wave passes the same number of workgroup barriers, each from one place in its own loop)
  static_assert(MH_THREADS == 8 * 64, "eight waves: six Riemann waves and two producers");
  const bool producer = (wave & 3) == 3;                      // waves 3 and 7: a SIMD of their own
  const int pid = wave >> 2;                                  // producer 0 traces two passes of 64 cells, producer 1 one
  const int dir = wave & 3, half = wave >> 2;                 // Riemann waves: direction, and which 64 of the tile's 128 cells

  // ---- producer role: primitives of two input cells per thread (cells pw and pw + 128 of the 19 x 11 input tile) ----
  const int pw = pid * 64 + lane;
  bool pok[2]; unsigned pidx2[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int cell = pw + 128 * r;
    const int qy = cell / MH_QX, qx = cell - qy * MH_QX;
    const int pi = i0 - 2 + qx, pj = j0 - 2 + qy;
    pok[r] = producer && cell < MH_QCELLS && pi < g.isize - 1 && pj < g.jsize - 1;   // range of mhd_prim_cell
    pidx2[r] = pok[r] ? (unsigned)pi + (unsigned)pj * g.sj : 0u;
  }
  // What a wave keeps from one iteration to the next.  Producers: pu[2][11] = the loaded U (8), Ua(i+1), Ub(j+1), Uc(k+1) and
  // then the primitives (8) + the cell's own face field (3) of their two input cells.  Riemann waves: the two carried states
  // c0, c1 (8 doubles each), declared with their loop below.
  double pu[2][11];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int v = 0; v < 11; ++v) pu[r][v] = 0.0;
  auto prim_load = [&](int k) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (pok[r]) {
        const double* u = U + pidx2[r] + (size_t)k * sk;
#pragma unroll
        for (int v = 0; v < 8; ++v) pu[r][v] = u[(size_t)v * N];
        pu[r][8] = u[(size_t)IA * N + 1];
        pu[r][9] = u[(size_t)IB * N + g.sj];
        pu[r][10] = u[(size_t)IC * N + sk];
      }
  };
  auto prim_compute = [&]() {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (pok[r]) {
        const Prim8 q = mhd_prim(g, pu[r], pu[r][8], pu[r][9], pu[r][10], dt);
        const double fa = pu[r][IA], fb = pu[r][IB], fc = pu[r][IC];
        pu[r][ID] = q.r; pu[r][IP] = q.p; pu[r][IU] = q.u; pu[r][IV] = q.v; pu[r][IW] = q.w; pu[r][IA] = q.a; pu[r][IB] = q.b; pu[r][IC] = q.c;
        pu[r][8] = fa; pu[r][9] = fb; pu[r][10] = fc;
      }
  };
  auto prim_store = [&](int k) {
    double* qd = LQ + (k % 3) * MH_QBSLOT;
#pragma unroll
    for (int r = 0; r < 2; ++r)
      if (pok[r]) {
#pragma unroll
        for (int v = 0; v < MH_NQB; ++v) qd[(pw + 128 * r) * MH_NQB + v] = pu[r][v];
      }
  };

  // ---- everybody: edge electric field of plane k from Q / B of planes k-1, k.  The trace of the 17 x 9 traced cells reads
  // Ex at (c, c+y), Ey at (c, c+x), Ez at (c, c+x, c+y): 17 x 10 + 18 x 9 + 18 x 10 = 512 values -- one per thread. ----
  // Which value, where: what does not change from plane to plane (comp < 0: this thread has no such value).
  struct ESlot { int comp; unsigned q00, store; double xPos; };
  constexpr int NEX = MH_PX * (MH_PY + 1), NEY = (MH_PX + 1) * MH_PY;
  constexpr int NE = NEX + NEY + (MH_PX + 1) * (MH_PY + 1);
  static_assert(NE <= MH_THREADS, "at most one edge value per thread and plane (16 x 8 tile: exactly 512)");
  auto elec_slot = [&](int e) {
    ESlot es = {-1, 0u, 0u, 0.0};
    if (e < 0 || e >= NE) return es;
    int comp, ex, ey;
    if (e < NEX) { comp = 0; ey = e / MH_PX; ex = e - ey * MH_PX; }
    else if (e < NEX + NEY) { comp = 1; const int c = e - NEX; ey = c / (MH_PX + 1); ex = c - ey * (MH_PX + 1); }
    else { comp = 2; const int c = e - NEX - NEY; ey = c / (MH_PX + 1); ex = c - ey * (MH_PX + 1); }
    const int ei = i0 - 1 + ex, ej = j0 - 1 + ey;
    if (!(ei < g.isize - 1 && ej < g.jsize - 1)) return es;   // range of mhd_elec_cell (low bounds hold by construction)
    es.comp = comp;
    es.q00 = (unsigned)(ey * MH_QX + ex) * (unsigned)MH_NQB;   // the cell diagonally below: (row 0, col 0) of the 2 x 2 cells read
    es.store = (unsigned)((ey + 1) * MH_QX + ex + 1) * 3u + (unsigned)comp;
    es.xPos = g.xMin + g.dx / 2 + (ei - gw) * g.dx;
    return es;
  };
  auto elec_value = [&](int k, const ESlot& es) {
    if (es.comp < 0) return;
    const unsigned qlo = (unsigned)(((k - 1) % 3) * MH_QBSLOT), qhi = (unsigned)((k % 3) * MH_QBSLOT);
    TraceInLds<G> in;
    in.lq = reinterpret_cast<const char*>(LQ); in.le = 0; in.flag = 0; in.want = 0;
    in.qA[0] = rg_opaque((qlo + es.q00) * 8u); in.qA[1] = rg_opaque((qhi + es.q00) * 8u); in.qA[2] = 0;
    in.qB[0] = in.qB[1] = in.qB[2] = 0; in.eA[0] = in.eA[1] = 0;
    constexpr unsigned MC = (unsigned)(MH_QX + 1);   // the cell itself, relative to that neighbourhood
    double v;
    if (es.comp == 0) v = mhd_elec_comp<0>(g, in, es.xPos, MC);
    else if (es.comp == 1) v = mhd_elec_comp<1>(g, in, es.xPos, MC);
    else v = mhd_elec_comp<2>(g, in, es.xPos, MC);
    (LE + (k & 1) * MH_ESLOT)[es.store] = v;
  };

  // rendezvous of the TWO producer waves (the only readers and writers of Q / B / E) through an LDS counter: lets them
  // recycle the Q / B and E slots inside the main phase, while the six Riemann waves keep solving -- the workgroup barrier
  // is then needed once per plane only (for T).  LDS operations of a wave complete in order, so a wave that has seen the
  // partner's increment also sees the stores the partner issued before it.
  int npair = 0;
  auto pair_sync = [&]() {
    npair += 2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_fetch_add(&Lsync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(&Lsync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < npair) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  };

  // ---- producer role: trace ----
  auto trace_cell = [&](int k, int cell, int e_want) {
    const int ty = cell / MH_PX, tx = cell - ty * MH_PX;
    const int ti = i0 - 1 + tx, tj = j0 - 1 + ty;
    if (cell < MH_CELLS && ti <= g.isize - gw && tj <= g.jsize - gw) {   // low bounds hold by construction
      const IJK c = {ti, tj, k};
      const TLdsWrite<G> tw = {LT + (k & 1) * MH_BUF + cell * G::TREC};
      // input cell (tx, ty) of the 19 x 11 input tile = (row 0, col 0) of this traced cell's 3 x 3 neighbourhood
      const unsigned q00 = (unsigned)(ty * MH_QX + tx) * (unsigned)MH_NQB, e10 = (unsigned)((ty + 1) * MH_QX + tx) * 3u;
      TraceInLds<G> in;
      in.lq = reinterpret_cast<const char*>(LQ); in.le = reinterpret_cast<const char*>(LE); in.flag = &Lesync; in.want = e_want;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const unsigned rec = (unsigned)(((k - 1 + p) % 3) * MH_QBSLOT) + q00;
        in.qA[p] = rg_opaque(rec * 8u);
        in.qB[p] = rg_opaque((rec + (unsigned)(2 * MH_QX * MH_NQB)) * 8u);
      }
      in.eA[0] = rg_opaque(((unsigned)((k & 1) * MH_ESLOT) + e10) * 8u);
      in.eA[1] = rg_opaque(((unsigned)(((k + 1) & 1) * MH_ESLOT) + e10) * 8u);
      mhd_trace3d_at(g, in, tw, dtdx, dtdy, dtdz, c, (unsigned)(MH_QX + 1));
    }
  };

  // ---- Riemann role ----
  // the six Riemann waves compute the edge electric field of plane kk+2 (384 threads, two trips over its 512 values)
  constexpr int E_NTHREADS = 384, E_NWAVES = E_NTHREADS / 64;
  const int ethread = producer ? -1 : (dir + 3 * half) * 64 + lane;
  // (the z march: the values ethread and ethread + 384 of every plane, decoded once)
  const ESlot es0 = elec_slot(ethread), es1 = elec_slot(ethread < 0 ? -1 : ethread + E_NTHREADS);
  const int cl = half * 64 + lane;
  const int oy = cl / MH_OX, ox = cl - oy * MH_OX;
  const int ci = i0 + ox, cj = j0 + oy;
  const bool fl_ok = !producer && cl < MH_OX * MH_OY && ci <= g.isize - gw && cj <= g.jsize - gw;
  const unsigned cidx2 = fl_ok ? flux_index(g, ci, cj, 0) : 0u;   // (F / emf have a pitch of their own)
  const unsigned cm00 = (unsigned)(oy * MH_PX + ox) * (unsigned)(G::TREC * 8);   // byte offset of the T record of the cell diagonally below
  const double xPos = g.xMin + g.dx / 2 + (ci - gw) * g.dx;

  // prologue: primitives of planes sa-2, sa-1, sa; electric field of plane sa-1
  if (t == 0) { Lsync = 0; Lesync = 0; }
  {
    // the loads of all three planes in flight at once (the start-up of a workgroup is 5 % of a 64-plane slab's march): the second and third
    // plane wait in registers of their own, dead before the main loop starts
    double pq[2][2][11];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (pok[r]) {
          const double* u = U + pidx2[r] + (size_t)(sa - 1 + q) * sk;
#pragma unroll
          for (int v = 0; v < 8; ++v) pq[q][r][v] = u[(size_t)v * N];
          pq[q][r][8] = u[(size_t)IA * N + 1];
          pq[q][r][9] = u[(size_t)IB * N + g.sj];
          pq[q][r][10] = u[(size_t)IC * N + sk];
        }
    prim_load(sa - 2);
    prim_compute();
    prim_store(sa - 2);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (pok[r]) {
#pragma unroll
          for (int v = 0; v < 11; ++v) pu[r][v] = pq[q][r][v];
        }
      prim_compute();
      prim_store(sa - 1 + q);
    }
    __syncthreads();
    elec_value(sa - 1, elec_slot(t));
  }
  __syncthreads();
  // iteration kk.  Riemann waves: electric field of plane kk+2 (from Q / B of planes kk+1, kk+2, complete since the last
  // barrier) into the slot E(kk) vacated, announce it, then the Riemann problems of plane kk.  Producers: loads of U(kk+3),
  // trace(kk+1) (waits for that announcement just before it reads E), prim(kk+3) -> the slot Q / B (kk) vacated.
  // kk = sa-2 only traces plane sa-1, kk = sa-1 traces plane sa and builds the carried states from T(sa-1); the last
  // planes have nothing left to produce.
  // One loop per role (the role of a wave never changes): the register allocator sees two independent live sets -- the producers'
  // input cells, the Riemann waves' carried states -- instead of their union in every wave.  Every wave passes the same number of
  // workgroup barriers, ONE per plane: T(kk+1) and Q / B (kk+3) complete, T(kk) free.
  // (round 3: E(kk+2) computed by the producer pair instead -- their SIMD issues fewer instructions per plane -- made the
  //  sweep SLOWER, 34.25 against 32.29 ms: the producers' chain load -> [E] -> trace -> barrier is the latency-critical one)
  if (producer) {
    int nit = 0;
    for (int kk = sa - 2; kk < sb; ++kk) {
      ++nit;
      const bool more = kk + 3 <= sb;
      const bool tracing = kk + 1 < sb;
      if (more) prim_load(kk + 3);
      if (tracing) {
        if (pid == 0) { trace_cell(kk + 1, lane, E_NWAVES * nit); trace_cell(kk + 1, 128 + lane, E_NWAVES * nit); }
        else trace_cell(kk + 1, 64 + lane, E_NWAVES * nit);
      }
      if (more) {
        prim_compute();
        pair_sync();                     // both producers are done reading Q / B (kk)
        prim_store(kk + 3);              // -> the Q / B slot of plane kk
      }
      __syncthreads();
    }
  } else {
    Prim8 c0 = {0, 0, 0, 0, 0, 0, 0, 0}, c1 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int kk = sa - 2; kk < sb; ++kk) {
      if (ethread >= 0) {
        if (kk + 1 < sb) { elec_value(kk + 2, es0); elec_value(kk + 2, es1); }   // E(kk+2) -> the slot of E(kk), dead since trace(kk)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(&Lesync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (fl_ok && kk >= sa - 1) {
        const unsigned tk0 = (unsigned)((kk & 1) * MH_BUF * 8) + cm00;
        const TLdsPlane<G> Tk = {reinterpret_cast<const char*>(LT), rg_opaque(tk0), rg_opaque(tk0 + (unsigned)(MH_PX * G::TREC * 8))};
        const unsigned idx = cidx2 + (unsigned)kk * g.fsk;
        const bool solve = kk >= sa;
        // the two Riemann waves of a SIMD are arbitrated oldest first: the younger one is favoured for its EMF (see riemann_dir)
        // (round 6, same box, contracted | exact sweep: this scheme 21.5 | 28.8-29.3 ms; no priorities 21.6; the younger wave favoured for the
        //  whole phase 21.6-22.3; the older wave favoured for the flux in return 22.3 | 30.1-30.7; roles swapped every plane 22.3 | 30.2)
        const bool raise = solve && wave >= 4;
        if (raise) __builtin_amdgcn_s_setprio(1);
        if (dir == 0) riemann_dir<G, XD>(g, Tk, (unsigned)(MH_PX + 1), xPos, F, emf, idx, c0, c1, solve, raise);
        else if (dir == 1) riemann_dir<G, YD>(g, Tk, (unsigned)(MH_PX + 1), xPos, F, emf, idx, c0, c1, solve, raise);
        else riemann_dir<G, ZD>(g, Tk, (unsigned)(MH_PX + 1), xPos, F, emf, idx, c0, c1, solve, raise);
      }
      __syncthreads();
    }
  }
}

// Periodic faces: every input of the Riemann problems at the layer j = ny + gw (i = nx + gw) is a ghost copy of what the
// layer j = gw (i = gw) sees, so their fluxes and EMFs are the same doubles.  When that layer would need a tile row (column) of
// its own -- ny (nx) a multiple of the tile edge -- the sweep leaves it out and this kernel copies it: 15 + 3 components per
// cell of one layer instead of 1/65 (1/33) of the sweep.  axis 1: y layer (all i), axis 0: x layer (all j).
// shear_save != 0 (the shearing box; axis 1 launches only): the same launch also saves the two emfY border columns of planes
// [k0, k0 + nk) for the flux / emf remap (shear_save_emf_cell, MHDRunGodunov.cpp:3203-3300) -- threads [n_copy, ...): one per (j, k).
// The row j = jsize - gw they read is the layer being copied by the other threads: they take its source row instead.
// Two plane ranges in one launch (the boundary ranges of a slab): planes k0 .. k0 + nk1 - 1, then k0b .. (nk1 < 0: one range).
struct K_copy_periodic_layer {
  DevParams g; double* F; double* emf; int axis, k0;
  double* shear_save; unsigned n_copy; int copy_on;
  int nk1, k0b;
  RG_DEVFN unsigned plane(unsigned kk) const { return (nk1 >= 0 && (int)kk >= nk1) ? (unsigned)k0b + (kk - (unsigned)nk1) : (unsigned)k0 + kk; }
  RG_DEVFN void operator()(unsigned t) const {
    if (t >= n_copy) {
      const unsigned q = t - n_copy;
      const unsigned j = q % (unsigned)g.jsize, k = plane(q / (unsigned)g.jsize);
      const unsigned js = (copy_on && (int)j == g.jsize - g.gw) ? (unsigned)g.gw : j;
      const size_t N = g.fN, P = (size_t)g.jsize * g.ksize;
      const size_t row = (size_t)g.fsj * js + (size_t)g.fsk * k + g.foff;
      const unsigned idx2 = j + (unsigned)g.jsize * k;
      shear_save[idx2] = emf[row + g.gw + (size_t)EMF_Y * N];
      shear_save[idx2 + P] = emf[row + g.nx + g.gw + (size_t)EMF_Y * N];
      return;
    }
    const unsigned n = (axis == 1) ? (unsigned)g.isize : (unsigned)g.jsize;
    const unsigned a = t % n, k = plane(t / n);
    const size_t N = g.fN;
    size_t src, dst;
    if (axis == 1) { src = flux_index(g, (int)a, g.gw, (int)k); dst = flux_index(g, (int)a, g.jsize - g.gw, (int)k); }
    else { src = flux_index(g, g.gw, (int)a, (int)k); dst = flux_index(g, g.isize - g.gw, (int)a, (int)k); }
    for (int v = 0; v < F_COUNT; ++v) F[dst + (size_t)v * N] = F[src + (size_t)v * N];
    for (int v = 0; v < 3; ++v) emf[dst + (size_t)v * N] = emf[src + (size_t)v * N];
  }
};

// reuse: bit 0 = the x layer i = nx + gw, bit 1 = the y layer j = ny + gw may be copied from the periodic image (the caller
// knows the boundary conditions)
template <int SPEC>
inline int launch_mhd3d_sweep(rg_stream_t s, const DevParams& g, const double* U, double* F,
                              double* emf, double dt, double dtdx, double dtdy, double dtdz, int ra, int rb, int reuse, const StepClock* clk, double* shear_save, int ra2) {
  TileGrid tg;
  constexpr int OX = MhMain::OX, OY = MhMain::OY;
  tg.nbx = (g.isize - 2 * g.gw + 1 + OX - 1) / OX;   // cells gw .. isize-gw
  tg.nby = (g.jsize - 2 * g.gw + 1 + OY - 1) / OY;
  const bool copy_x = (reuse & 1) && g.nx % OX == 0 && g.nx >= OX, copy_y = (reuse & 2) && g.ny % OY == 0 && g.ny >= OY;
  if (copy_x) tg.nbx -= 1;
  if (copy_y) tg.nby -= 1;
  // nx a multiple of the tile width and no periodic image to copy (the shearing box): the face column i = isize - gw would be a tile
  // column of its own with one valid column in 16 -- it goes to a second launch with the 2 x 32 geometry instead (MhLastX)
  const bool lastx = !copy_x && g.nx % OX == 0 && tg.nbx >= 2;
  if (lastx) tg.nbx -= 1;
  const int span = rb - ra;
  const int zseg_env = rgpu::options().zseg;
  // one workgroup is resident per CU: 32 per XCD; a segment costs two extra iterations (pipeline fill).  512^3 shearing box: 2048 tiles
  // of one base segment = 8 rounds of workgroups exactly, then 16 tiles of the last face column in 16 segments = one short round (round
  // 4: 2112 tiles, 8 rounds + 8 items per XCD cut into 4 sub-segments each)
  // ra2 > 0: a second range [ra2, ra2 + span) of the same length in the same launch (TileGrid::zsplit) -- the boundary ranges of a
  // slab in the boundary-first schedule: one launch, one last round of workgroups, instead of two
  const bool pair = ra2 > 0;
  const int nplanes = pair ? 2 * span : span;
  tile_grid_plan(tg, span, 32, 8, 2, zseg_env, pair);
  if (pair) { tg.zsplit = ra + span; tg.zgap = ra2 - (ra + span); }
  hipLaunchKernelGGL((mhd3d_sweep_kernel<SPEC, MhMain>), dim3(8u * (unsigned)tg.per_xcd), dim3(MH_THREADS), 0, s, g, tg, U, F, emf,
                     dt, dtdx, dtdy, dtdz, ra, ra + nplanes, clk);
  if (hipGetLastError() != hipSuccess) return -1;
  if (lastx) {
    TileGrid tl;
    tl.nbx = 1;
    tl.nby = (g.jsize - 2 * g.gw + (copy_y ? 0 : 1) + MhLastX::OY - 1) / MhLastX::OY;   // (rows past the face range are masked in the kernel)
    tile_grid_plan(tl, span, 32, 8, 2, zseg_env, pair);
    if (pair) { tl.zsplit = ra + span; tl.zgap = ra2 - (ra + span); }
    hipLaunchKernelGGL((mhd3d_sweep_kernel<SPEC, MhLastX>), dim3(8u * (unsigned)tl.per_xcd), dim3(MH_THREADS), 0, s, g, tl, U, F, emf,
                       dt, dtdx, dtdy, dtdz, ra, ra + nplanes, clk);
    if (hipGetLastError() != hipSuccess) return -1;
  }
  // x layer first (rows gw .. jsize-gw-1 hold sweep results), then the y layer over all i: the corner comes out right
  const int nk1 = pair ? span : -1;
  if (copy_x) { const unsigned n = (unsigned)g.jsize * (unsigned)nplanes; const K_copy_periodic_layer k = {g, F, emf, 0, ra, 0, n, 1, nk1, ra2}; if (rgpu::rg_launch<256>(s, n, k)) return -1; }
  if (copy_y || shear_save) {   // one launch: the y layer (all i) and, shearing box, the emfY border columns for the remap
    const unsigned nc = copy_y ? (unsigned)g.isize * (unsigned)nplanes : 0u, ns = shear_save ? (unsigned)g.jsize * (unsigned)nplanes : 0u;
    const K_copy_periodic_layer k = {g, F, emf, 1, ra, shear_save, nc, copy_y ? 1 : 0, nk1, ra2};
    if (rgpu::rg_launch<256>(s, nc + ns, k)) return -1;
  }
  return 0;
}

// configurations the fused sweep covers (everything but the per-cell gravity field, which the driver excludes itself)
inline bool mhd3d_sweep_covers(const DevParams& g) { return tiled_enabled() && g.three_d && g.mhd; }

// spec: 0 = generic, 1 = isothermal rotating box (MRI), 2 = adiabatic inertial box (the driver's pick_spec)
// Solves the Riemann problems of planes [ra, rb) (already clipped to [gw, ksize-gw]) from U alone (primitives, electric
// field and traced state are produced on the way).  Returns 0 = done, 1 = not applicable (the caller runs prim, elec,
// trace and Riemann as flat kernels), < 0 = launch error.
template <int SPEC_MRI, int SPEC_PLAIN>
inline int mhd3d_sweep(rg_stream_t s, const DevParams& g, int spec, const double* U, double* F,
                       double* emf, double dt, double dtdx, double dtdy, double dtdz, int ra, int rb, int reuse = 0, const StepClock* clk = 0,
                       double* shear_save = 0, int ra2 = 0) {
  if (!mhd3d_sweep_covers(g) || g.grav_on == 2) return 1;   // per-cell gravity field: flat kernels
  if (rb <= ra) return 0;
  if (spec == 1) return launch_mhd3d_sweep<SPEC_MRI>(s, g, U, F, emf, dt, dtdx, dtdy, dtdz, ra, rb, reuse, clk, shear_save, ra2);
  if (spec == 2) return launch_mhd3d_sweep<SPEC_PLAIN>(s, g, U, F, emf, dt, dtdx, dtdy, dtdz, ra, rb, reuse, clk, shear_save, ra2);
  return launch_mhd3d_sweep<SPEC_NONE>(s, g, U, F, emf, dt, dtdx, dtdy, dtdz, ra, rb, reuse, clk, shear_save, ra2);
}

}  // namespace rgpu_tiled
