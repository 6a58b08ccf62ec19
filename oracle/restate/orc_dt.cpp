// orc_dt.cpp -- ORACLE (test infrastructure).  CFL time step scan.
//   compute_dt      HydroRunBase.cpp:372-426        compute_dt_mhd   MHDRunBase.cpp:140-250
// Returns invDt; the caller forms cfl / invDt.
#include "orc_pointwise.h"

namespace orc {

double compute_inv_dt(const Ctx& c, const double* U) {
  const rgpu_params& p = c.p;
  const int gw = c.gw;
  const size_t N = c.ncell;
  if (!p.mhdEnabled) {
    double invDt = 0;
    if (!c.three_d) {
      for (int j = gw; j < c.jsize - gw; j++)
        for (int i = gw; i < c.isize - gw; i++) {
          const size_t o = c.idx(i, j, 0);
          const double u[4] = {U[o], U[o + N], U[o + 2 * N], U[o + 3 * N]};
          double q[4], cs;
          hydro_constoprim<4>(p, u, q, cs);
          const double vx = cs + fabs(q[IU]), vy = cs + fabs(q[IV]);
          invDt = fmax(invDt, vx / c.dx + vy / c.dy);
        }
    } else {
      for (int k = gw; k < c.ksize - gw; k++)
        for (int j = gw; j < c.jsize - gw; j++)
          for (int i = gw; i < c.isize - gw; i++) {
            const size_t o = c.idx(i, j, k);
            const double u[5] = {U[o], U[o + N], U[o + 2 * N], U[o + 3 * N], U[o + 4 * N]};
            double q[5], cs;
            hydro_constoprim<5>(p, u, q, cs);
            const double vx = cs + fabs(q[IU]), vy = cs + fabs(q[IV]), vz = cs + fabs(q[IW]);
            invDt = fmax(invDt, vx / c.dx + vy / c.dy + vz / c.dz);
          }
    }
    if (p.enableJet) invDt = fmax(invDt, (p.ujet + p.cjet) / c.dx);
    return invDt;
  }
  // MHD: the CPU path seeds with smallc / min(dx,dy) (MHDRunBase.cpp:144)
  double invDt = p.smallc / fmin(c.dx, c.dy);
  if (!c.three_d) {
    for (int j = gw; j < c.jsize - gw; j++)
      for (int i = gw; i < c.isize - gw; i++) {
        const size_t o = c.idx(i, j, 0);
        double u[8];
        for (int v = 0; v < 8; ++v) u[v] = U[o + v * N];
        const double bnb[3] = {U[o + 1 + IA * N], U[o + c.isize + IB * N], 0.0};
        double q[8], cs;
        mhd_constoprim(p, u, bnb, q, cs, 0.0);
        double s[3];
        find_speed_info<2>(p, q, s);
        if (p.enableJet) invDt = fmax(fmax(invDt, s[IX] / c.dx + s[IY] / c.dy), (p.ujet + p.cjet) / c.dx);
        else invDt = fmax(invDt, s[IX] / c.dx + s[IY] / c.dy);
      }
  } else {
    const double deltaX = p.xMax - p.xMin;
    const size_t sj = c.isize, sk = (size_t)c.isize * c.jsize;
    for (int k = gw; k < c.ksize - gw; k++)
      for (int j = gw; j < c.jsize - gw; j++)
        for (int i = gw; i < c.isize - gw; i++) {
          const size_t o = c.idx(i, j, k);
          double u[8];
          for (int v = 0; v < 8; ++v) u[v] = U[o + v * N];
          const double bnb[3] = {U[o + 1 + IA * N], U[o + sj + IB * N], U[o + sk + IC * N]};
          double q[8], cs;
          mhd_constoprim(p, u, bnb, q, cs, 0.0);
          double s[3];
          find_speed_info<3>(p, q, s);
          const double vx = s[IX];
          double vy = s[IY];
          if (p.Omega0 > 0) vy += 1.5 * p.Omega0 * deltaX / 2;
          const double vz = s[IZ];
          if (p.enableJet) invDt = fmax(fmax(invDt, vx / c.dx + vy / c.dy + vz / c.dz), (p.ujet + p.cjet) / c.dx);
          else invDt = fmax(invDt, vx / c.dx + vy / c.dy + vz / c.dz);
        }
  }
  return invDt;
}

}  // namespace orc
