#include "init_conditions.h"

#include <cmath>
#include <cstring>
#include <stdexcept>

namespace rgpu_host {

void Rand48::skip(unsigned long long n) {
  // compose the affine map x -> a x + c with itself n times (mod 2^48) by repeated squaring
  unsigned long long a = kA, c = kC, ra = 1, rc = 0;
  while (n) {
    if (n & 1ULL) {
      ra = (ra * a) & kMask;
      rc = (rc * a + c) & kMask;
    }
    c = (c * a + c) & kMask;
    a = (a * a) & kMask;
    n >>= 1;
  }
  x_ = (ra * x_ + rc) & kMask;
}

namespace {

struct Grid {
  int nx, ny, nz, gw, isize, jsize, ksize, nvar;
  int k_shift;      // global k = local k + k_shift
  int ksize_glob;   // ksize of the undecomposed domain
  int nz_glob;
  size_t ncell;
  double* U;
  bool three_d;
  double& at(int i, int j, int k, int v) const {
    return U[static_cast<size_t>(i) + static_cast<size_t>(isize) * (j + static_cast<size_t>(jsize) * k) + ncell * v];
  }
};

Grid make_grid(const rgpu_params& p, double* hU) {
  Grid g;
  g.nx = p.nx; g.ny = p.ny; g.nz = p.nz; g.gw = p.ghostWidth;
  g.three_d = (p.nz_global != 1);
  g.isize = p.nx + 2 * p.ghostWidth;
  g.jsize = p.ny + 2 * p.ghostWidth;
  g.ksize = g.three_d ? p.nz + 2 * p.ghostWidth : 1;
  g.nvar = p.nbVar;
  g.k_shift = p.slab_rank * p.nz;
  g.nz_glob = p.nz_global;
  g.ksize_glob = g.three_d ? p.nz_global + 2 * p.ghostWidth : 1;
  g.ncell = static_cast<size_t>(g.isize) * g.jsize * g.ksize;
  g.U = hU;
  return g;
}

// ---- hydro: jet (HydroRunBase.cpp:5282-5350) ---------------------------------------------------------------
void init_hydro_jet(const rgpu_params& p, const Grid& g) {
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        g.at(i, j, k, RGPU_ID) = 1.0f;
        g.at(i, j, k, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f);
        g.at(i, j, k, RGPU_IU) = 0.0f;
        g.at(i, j, k, RGPU_IV) = 0.0f;
        if (g.three_d) g.at(i, j, k, RGPU_IW) = 0.0f;
      }
}

// ---- hydro: implode (HydroRunBase.cpp:5449-5536) -------------------------------------------------------------
void init_hydro_implode(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const float amplitude = cfg.get_float("implode", "amplitude", 0.0f);
  if (amplitude != 0.0f)
    throw std::runtime_error("implode.amplitude != 0 needs the libc rand() stream: outside the implemented scope");
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        // the test is done in FLOAT on ghost-offset indexes (HydroRunBase.cpp:5466, 5501)
        bool heavy;
        if (g.three_d)
          heavy = ((float)i / g.nx + (float)j / g.ny + (float)(k + g.k_shift) / g.nz_glob) > 0.5;
        else
          heavy = ((float)i / g.nx + (float)j / g.ny) > 0.5;
        if (heavy) {
          g.at(i, j, k, RGPU_ID) = 1.0f;
          g.at(i, j, k, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f);
        } else {
          g.at(i, j, k, RGPU_ID) = 0.125f;
          g.at(i, j, k, RGPU_IP) = 0.14f / (p.gamma0 - 1.0f);
        }
        g.at(i, j, k, RGPU_IU) = 0.0f;
        g.at(i, j, k, RGPU_IV) = 0.0f;
        if (g.three_d) g.at(i, j, k, RGPU_IW) = 0.0f;
      }
}

// ---- MHD: Orszag-Tang (MHDRunBase.cpp:1378-1570; 3D only direction 0) -----------------------------------------
void init_orszag_tang(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const double TwoPi = 4.0 * std::asin(1.0);
  const double B0 = 1.0 / std::sqrt(2.0 * TwoPi);
  const double p0 = (double)(p.gamma0 / (2.0 * TwoPi));
  const double d0 = (double)(p.gamma0 * p0);
  const double v0 = 1.0;
  double kt = 0.0;
  if (g.three_d) {
    int direction = static_cast<int>(cfg.get_integer("OrszagTang", "direction", 0));
    if (direction < 0 || direction > 3) direction = 0;
    if (direction != 0) throw std::runtime_error("Orszag-Tang 3D: only direction=0 (vortex in the x-y plane) is implemented");
    kt = cfg.get_float("OrszagTang", "kt", 0.0f);
  }
  for (int k = 0; k < g.ksize; ++k) {
    const double zPos = p.zMin + p.dz / 2 + (k + g.k_shift - g.gw) * p.dz;
    for (int j = 0; j < g.jsize; ++j) {
      const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = 0; i < g.isize; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        g.at(i, j, k, RGPU_ID) = d0;
        g.at(i, j, k, RGPU_IU) = -d0 * v0 * std::sin(yPos * TwoPi);
        g.at(i, j, k, RGPU_IV) = d0 * v0 * std::sin(xPos * TwoPi);
        g.at(i, j, k, RGPU_IW) = 0.0;
        if (g.three_d) {
          g.at(i, j, k, RGPU_IA) = -B0 * std::cos(2 * TwoPi * kt * (zPos - p.zMin) / (p.zMax - p.zMin)) * std::sin(yPos * TwoPi);
          g.at(i, j, k, RGPU_IB) = B0 * std::cos(2 * TwoPi * kt * (zPos - p.zMin) / (p.zMax - p.zMin)) * std::sin(2.0 * xPos * TwoPi);
        } else {
          g.at(i, j, k, RGPU_IA) = -B0 * std::sin(yPos * TwoPi);
          g.at(i, j, k, RGPU_IB) = B0 * std::sin(2.0 * xPos * TwoPi);
        }
        g.at(i, j, k, RGPU_IC) = 0.0;
      }
    }
  }
  // total energy with the cell-centred field of the periodic box.  Only i<isize-1, j<jsize-1 is reproduced:
  // the reference's remaining branches touch ghost cells only (and in 3D write them to a wrong plane,
  // MHDRunBase.cpp:1544-1561); every ghost is overwritten by make_all_boundaries before first use.
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize - 1; ++j)
      for (int i = 0; i < g.isize - 1; ++i) {
        const double d = g.at(i, j, k, RGPU_ID), mu = g.at(i, j, k, RGPU_IU), mv = g.at(i, j, k, RGPU_IV);
        const double sa = g.at(i, j, k, RGPU_IA) + g.at(i + 1, j, k, RGPU_IA);
        const double sb = g.at(i, j, k, RGPU_IB) + g.at(i, j + 1, k, RGPU_IB);
        g.at(i, j, k, RGPU_IP) = p0 / (p.gamma0 - 1.0) + 0.5 * ((mu * mu) / d + (mv * mv) / d + 0.25 * (sa * sa) + 0.25 * (sb * sb));
      }
}

// ---- MHD: Brio-Wu (MHDRunBase.cpp:1870-2066; 2D directions 0,1,3 and 3D directions 0,1,2) --------------------
void init_brio_wu(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const double B0 = cfg.get_float("BrioWu", "B0", 1.0f);
  const double B1 = cfg.get_float("BrioWu", "B1", 0.75f);
  const double d0 = cfg.get_float("BrioWu", "d0", 1.0f);
  const double d1 = cfg.get_float("BrioWu", "d1", 0.125f);
  const double p0 = 1.0, p1 = 0.1;
  int direction = static_cast<int>(cfg.get_integer("BrioWu", "direction", 0));
  if (direction < 0 || direction > 4) direction = 0;
  if (!g.three_d) {
    if (direction != 0 && direction != 1 && direction != 3)
      throw std::runtime_error("Brio-Wu 2D: directions 0, 1 and 3 are implemented");
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        bool left;
        double e_mag, bxl, byl, bxr, byr;
        if (direction == 0) {
          left = i < g.isize / 2;
          e_mag = 0.5 * (B0 * B0 + B1 * B1);
          bxl = B1; byl = B0; bxr = B1; byr = -B0;
        } else if (direction == 1) {
          left = j < g.jsize / 2;
          e_mag = 0.5 * (B0 * B0 + B1 * B1);
          bxl = B0; byl = B1; bxr = -B0; byr = B1;
        } else {
          left = 1.0 * i / g.isize + 1.0 * j / g.jsize < 1;
          e_mag = 0.5 * ((-B0 + B1) * (-B0 + B1) / 2 + (B0 + B1) * (B0 + B1) / 2);
          bxl = -B0 / std::sqrt(2.) + B1 / std::sqrt(2.); byl = B0 / std::sqrt(2.) + B1 / std::sqrt(2.);
          bxr = B0 / std::sqrt(2.) + B1 / std::sqrt(2.);  byr = -B0 / std::sqrt(2.) + B1 / std::sqrt(2.);
        }
        g.at(i, j, 0, RGPU_ID) = left ? d0 : d1;
        g.at(i, j, 0, RGPU_IP) = (left ? p0 : p1) / (p.gamma0 - 1.0f) + e_mag;
        g.at(i, j, 0, RGPU_IA) = left ? bxl : bxr;
        g.at(i, j, 0, RGPU_IB) = left ? byl : byr;
      }
  } else {
    if (direction > 2) throw std::runtime_error("Brio-Wu 3D: directions 0, 1 and 2 are implemented");
    for (int k = g.gw; k < g.ksize - g.gw; ++k)
      for (int j = g.gw; j < g.jsize - g.gw; ++j)
        for (int i = g.gw; i < g.isize - g.gw; ++i) {
          bool left;
          double bl[3], br[3];
          if (direction == 0) {
            left = i < g.isize / 2;
            bl[0] = B1; bl[1] = B0; bl[2] = B0; br[0] = B1; br[1] = -B0; br[2] = -B0;
          } else if (direction == 1) {
            left = j < g.jsize / 2;
            bl[0] = B0; bl[1] = B1; bl[2] = B0; br[0] = -B0; br[1] = B1; br[2] = -B0;
          } else {
            left = (k + g.k_shift) < g.ksize_glob / 2;
            bl[0] = B0; bl[1] = B0; bl[2] = B1; br[0] = -B0; br[1] = -B0; br[2] = B1;
          }
          g.at(i, j, k, RGPU_ID) = left ? d0 : d1;
          g.at(i, j, k, RGPU_IP) = (left ? p0 : p1) / (p.gamma0 - 1.0f) + 0.5 * (B0 * B0 + B0 * B0 + B1 * B1);
          g.at(i, j, k, RGPU_IA) = left ? bl[0] : br[0];
          g.at(i, j, k, RGPU_IB) = left ? bl[1] : br[1];
          g.at(i, j, k, RGPU_IC) = left ? bl[2] : br[2];
        }
  }
}

// ---- MHD: MRI in the shearing box (MHDRunBase.cpp:2677-2758) -------------------------------------------------
void init_mri(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  if (!g.three_d) throw std::runtime_error("MRI is only available in 3D");
  if (!p.shearingBoxEnabled && !(p.bc[0] == RGPU_BC_SHEARINGBOX && p.bc[1] == RGPU_BC_SHEARINGBOX))
    throw std::runtime_error("MRI needs shearing box conditions along x (boundary_xmin=boundary_xmax=4)");
  const double TwoPi = 4.0 * std::asin(1.0);
  const double d0 = cfg.get_float("MRI", "density", 1.0f);
  const double beta = cfg.get_float("MRI", "beta", 400.0f);
  const double p0 = d0 * p.cIso * p.cIso;
  const double zMax = cfg.get_float("mesh", "zmax", 1.0f);
  const std::string type = cfg.get_string("MRI", "type", "noflux");
  double B0;
  if (type == "pyl")
    B0 = 3.0 / 2.0 * std::sqrt(d0 * p.Omega0 * p.Omega0 * (zMax - p.zMin) * (zMax - p.zMin) / beta);
  else
    B0 = 2.0 * std::sqrt(p0 / beta);
  const double amp = cfg.get_float("MRI", "amp", 0.01f);
  const long seed = cfg.get_integer("MRI", "seed", 0);
  const double d_amp = cfg.get_float("MRI", "density_fluctuations", 0.0f);

  Rand48 rng(seed);
  // 4 draws per cell in k,j,i order over the whole ghost-inclusive domain: skip the planes below this slab.
  // Slab ghost planes below k_shift belong to the previous slab's stream positions, which is what we want:
  // local plane k is global plane k + k_shift.
  rng.skip(4ULL * static_cast<unsigned long long>(g.k_shift) * g.jsize * g.isize);
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j)
      for (int i = 0; i < g.isize; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        g.at(i, j, k, RGPU_ID) = d0 * (1 + d_amp * 2 * (rng.next() - 0.5));
        g.at(i, j, k, RGPU_IP) = 0;
        g.at(i, j, k, RGPU_IU) = d0 * amp * (rng.next() - 0.5) * std::sqrt(p0);
        g.at(i, j, k, RGPU_IV) = d0 * amp * (rng.next() - 0.5) * std::sqrt(p0);
        g.at(i, j, k, RGPU_IW) = d0 * amp * (rng.next() - 0.5) * std::sqrt(p0);
        g.at(i, j, k, RGPU_IA) = 0.0;
        g.at(i, j, k, RGPU_IB) = 0.0;
        if (type == "noflux")
          g.at(i, j, k, RGPU_IC) = B0 * std::sin(TwoPi * xPos);
        else if (type == "pyl" || type == "fluxZ")
          g.at(i, j, k, RGPU_IC) = B0;
        else
          g.at(i, j, k, RGPU_IC) = 0.0;
      }
}

}  // namespace

void init_condition(const IniConfig& cfg, const rgpu_params& p, double* hU) {
  const Grid g = make_grid(p, hU);
  std::memset(hU, 0, sizeof(double) * g.ncell * g.nvar);
  const std::string problem = cfg.get_string("hydro", "problem", "unknown");
  if (p.mhdEnabled) {
    // dispatch of MHDRunBase::init_simulation (MHDRunBase.cpp:1286-1342)
    if (problem == "Orszag-Tang" || problem == "OrszagTang") init_orszag_tang(cfg, p, g);
    else if (problem == "Brio-Wu" || problem == "BrioWu" || problem == "brio-wu" || problem == "briowu") init_brio_wu(cfg, p, g);
    else if (problem == "MRI" || problem == "Mri" || problem == "mri") init_mri(cfg, p, g);
    else throw std::runtime_error("MHD problem '" + problem + "' is outside the implemented scope");
  } else {
    if (problem == "jet") init_hydro_jet(p, g);
    else if (problem == "implode") init_hydro_implode(cfg, p, g);
    else throw std::runtime_error("hydro problem '" + problem + "' is outside the implemented scope");
  }
}

}  // namespace rgpu_host
