#include "init_conditions.h"

#include <cmath>
#include <cstring>
#include <stdexcept>
#include <vector>

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

GlibcRand::GlibcRand(unsigned seed) {
  if (seed == 0) seed = 1;
  int word = static_cast<int>(seed);
  r_[0] = seed;
  for (int i = 1; i < 31; ++i) {   // 16807 * word mod (2^31 - 1) by Schrage's method, as srandom_r does
    const long hi = word / 127773, lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = static_cast<int>(w);
    r_[i] = static_cast<unsigned>(word);
  }
  f_ = 3; b_ = 0;
  for (int i = 0; i < 310; ++i) next();
}

int GlibcRand::next() {
  r_[f_] += r_[b_];
  const unsigned out = r_[f_] >> 1;
  f_ = (f_ + 1) % 31;
  b_ = (b_ + 1) % 31;
  return static_cast<int>(out);
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
  // density perturbation: one libc rand() draw per interior cell in k,j,i order (HydroRunBase.cpp:5455-5466)
  const double amplitude = cfg.get_float("implode", "amplitude", 0.0f);
  GlibcRand rng(static_cast<unsigned>(cfg.get_integer("implode", "seed", 1)));
  if (g.three_d) for (long n = 0; n < (long)g.k_shift * g.ny * g.nx; ++n) rng.next();   // cells of the slabs below
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
        const double pert = amplitude * (1.0 * rng.next() / GlibcRand::kRandMax - 0.5);
        if (heavy) {
          g.at(i, j, k, RGPU_ID) = 1.0f + pert;
          g.at(i, j, k, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f);
        } else {
          g.at(i, j, k, RGPU_ID) = 0.125f + pert;
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
  if (p.gravityEnabled) {
    // vertically stratified box (MHDRunBase.cpp:2763-2796): isothermal hydrostatic density with a floor, purely
    // azimuthal field within one scale height of the midplane; the momenta keep their d0-based perturbation
    const double zFloor = cfg.get_float("MRI", "zFloor", 5.0f);
    const double H = p.cIso / p.Omega0;
    for (int k = 0; k < g.ksize; ++k) {
      const double zPos = p.zMin + p.dz / 2 + (k + g.k_shift - g.gw) * p.dz;
      for (int j = 0; j < g.jsize; ++j)
        for (int i = 0; i < g.isize; ++i) {
          g.at(i, j, k, RGPU_ID) = d0 * std::fmax(std::exp(-(zPos * zPos) / 2.0 / (H * H)), std::exp(-zFloor * zFloor / 2.0));
          g.at(i, j, k, RGPU_IA) = 0.0;
          g.at(i, j, k, RGPU_IB) = 0.0;
          g.at(i, j, k, RGPU_IC) = 0.0;
          if (zPos < H && zPos > -H) g.at(i, j, k, RGPU_IB) = B0;
        }
    }
  }
}

// ---- hydro: Sod tube along x (HydroRunBase.cpp:5358-5438; the ghost-corner copies of the gw==2 case are dropped:
// the step refills every ghost before it reads one) --------------------------------------------------------------
void init_hydro_sod(const rgpu_params& p, const Grid& g) {
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        if (i < g.isize / 2) {
          g.at(i, j, k, RGPU_ID) = 1.0f;
          g.at(i, j, k, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f);
        } else {
          g.at(i, j, k, RGPU_ID) = 0.125f;
          g.at(i, j, k, RGPU_IP) = 0.1f / (p.gamma0 - 1.0f);
        }
        g.at(i, j, k, RGPU_IU) = 0.0f;
        g.at(i, j, k, RGPU_IV) = 0.0f;
        if (g.three_d) g.at(i, j, k, RGPU_IW) = 0.0f;
      }
}

// ---- hydro: spherical blast (HydroRunBase.cpp:5551-5676) -------------------------------------------------------
void init_hydro_blast(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  // every knob goes through getFloat, defaults included (float arithmetic on the defaults, HydroRunBase.cpp:5570-5577)
  double radius = cfg.get_float("blast", "radius", (float)(0.25 * (p.xMax - p.xMin)));
  const double center_x = cfg.get_float("blast", "center_x", (float)((p.xMax + p.xMin) / 2));
  const double center_y = cfg.get_float("blast", "center_y", (float)((p.yMax + p.yMin) / 2));
  const double center_z = cfg.get_float("blast", "center_z", (float)((p.zMax + p.zMin) / 2));
  const double density_in = cfg.get_float("blast", "density_in", 1.0f);
  const double density_out = cfg.get_float("blast", "density_out", 1.0f);
  const double pressure_in = cfg.get_float("blast", "pressure_in", 10.0f);
  const double pressure_out = cfg.get_float("blast", "pressure_out", 0.1f);
  radius *= radius;
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k) {
    const double zPos = p.zMin + p.dz / 2 + (k + g.k_shift - g.gw) * p.dz;
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        double d2 = (xPos - center_x) * (xPos - center_x) + (yPos - center_y) * (yPos - center_y);
        if (g.three_d) d2 = d2 + (zPos - center_z) * (zPos - center_z);
        const bool inside = d2 < radius;
        g.at(i, j, k, RGPU_ID) = inside ? density_in : density_out;
        g.at(i, j, k, RGPU_IP) = (inside ? pressure_in : pressure_out) / (p.gamma0 - 1.0f);
        g.at(i, j, k, RGPU_IU) = 0.0f;
        g.at(i, j, k, RGPU_IV) = 0.0f;
        if (g.three_d) g.at(i, j, k, RGPU_IW) = 0.0f;
      }
    }
  }
}

// ---- hydro: Kelvin-Helmholtz (HydroRunBase.cpp:5857-6252): random (libc rand(), 2 / 3 draws per interior cell in
// k,j,i order), sine, Athena-like and Robertson et al. perturbations ------------------------------------------------
void init_hydro_kelvin_helmholtz(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const char* S = "kelvin-helmholtz";
  GlibcRand rng(static_cast<unsigned>(cfg.get_integer(S, "seed", 1)));
  const double amplitude = cfg.get_float(S, "amplitude", 0.01f);
  const bool p_rand = cfg.get_bool(S, "perturbation_rand", true), p_sine = cfg.get_bool(S, "perturbation_sine", false);
  const bool p_athena = cfg.get_bool(S, "perturbation_sine_athena", false);
  const bool p_robertson = cfg.get_bool(S, "perturbation_sine_robertson", false);
  const double rho_inner = cfg.get_float(S, "rho_inner", 2.0f), rho_outer = cfg.get_float(S, "rho_outer", 1.0f);
  const double pressure = cfg.get_float(S, "pressure", 2.5f);
  const double inner_size = cfg.get_float(S, "inner_size", 0.2f), outer_size = cfg.get_float(S, "outer_size", 0.2f);
  const double vflow_in = cfg.get_float(S, "vflow_in", -0.5f), vflow_out = cfg.get_float(S, "vflow_out", 0.5f);
  const double xSize = p.xMax - p.xMin, ySize = p.yMax - p.yMin, zSize = p.zMax - p.zMin;
  const double yCenter = (p.yMin + p.yMax) * 0.5, zCenter = (p.zMin + p.zMax) * 0.5;
  const double e0 = pressure / (p.gamma0 - 1.0f);
  auto draw = [&]() { return 1.0 * rng.next() / GlibcRand::kRandMax - 0.5; };
  auto sqr = [](double x) { return x * x; };
  if (!g.three_d) {
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        double& d = g.at(i, j, 0, RGPU_ID); double& mu = g.at(i, j, 0, RGPU_IU); double& mv = g.at(i, j, 0, RGPU_IV);
        if (p_rand) {
          const bool outer = std::fabs(yPos - yCenter) > outer_size * ySize;
          d = outer ? rho_outer : rho_inner;
          mu = d * ((outer ? vflow_out : vflow_in) + amplitude * draw());
          mv = d * (0.0f + amplitude * draw());
        } else if (p_athena) {
          const double a = 0.05, sigma = 0.2, vflow = 0.5;
          d = rho_inner;
          mu = rho_inner * vflow * std::tanh(yPos / a);
          mv = rho_inner * amplitude * std::sin(2.0 * M_PI * xPos) * std::exp(-(yPos * yPos) / (sigma * sigma));
        } else if (p_robertson) {
          const int n = static_cast<int>(cfg.get_integer(S, "mode", 4));
          const double w0 = cfg.get_float(S, "w0", 0.1f), deltaY = cfg.get_float(S, "deltaY", 0.03f);
          const double y1 = p.yMin + 0.25 * ySize, y2 = p.yMin + 0.75 * ySize;
          const double ramp = 1.0 / (1.0 + std::exp(2 * (yPos - y1) / deltaY)) + 1.0 / (1.0 + std::exp(2 * (y2 - yPos) / deltaY));
          d = rho_inner + ramp * (rho_outer - rho_inner);
          mu = d * (vflow_in + ramp * (vflow_out - vflow_in));
          mv = d * w0 * std::sin(n * M_PI * xPos);
        } else if (p_sine) {
          const double perturb_vx = 0, perturb_vy = amplitude * std::sin(2.0 * M_PI * xPos / xSize);
          if (std::fabs(yPos - yCenter) > outer_size * ySize) {
            d = rho_outer; mu = rho_outer * vflow_out * (1.0 + perturb_vx); mv = rho_outer * perturb_vy;
          } else if (std::fabs(yPos - yCenter) <= inner_size * ySize) {
            d = rho_inner; mu = rho_inner * vflow_in * (1.0 + perturb_vx); mv = rho_inner * perturb_vy;
          } else {
            const double interpSize = outer_size - inner_size;
            const double rho_slope = (rho_outer - rho_inner) / (interpSize * ySize), u_slope = (vflow_out - vflow_in) / (interpSize * ySize);
            double deltaY, deltaRho, deltaU;
            if (yPos > yCenter) { deltaY = yPos - (yCenter + inner_size * ySize); deltaRho = rho_slope * deltaY; deltaU = u_slope * deltaY; }
            else { deltaY = yPos - (yCenter - inner_size * ySize); deltaRho = -rho_slope * deltaY; deltaU = -u_slope * deltaY; }
            d = rho_inner + deltaRho;
            mu = d * (vflow_in + deltaU) * (1.0 + perturb_vx);
            mv = d * perturb_vy;
          }
        } else {
          continue;   // no perturbation type selected: the reference leaves the state at zero
        }
        g.at(i, j, 0, RGPU_IP) = e0 + 0.5 * (sqr(mu) + sqr(mv)) / d;
      }
    }
    return;
  }
  if (p_rand) for (long n = 0; n < 3L * g.k_shift * g.ny * g.nx; ++n) rng.next();   // draws of the slabs below
  for (int k = g.gw; k < g.ksize - g.gw; ++k) {
    const double zPos = p.zMin + p.dz / 2 + (k + g.k_shift - g.gw) * p.dz;
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        double& d = g.at(i, j, k, RGPU_ID); double& mu = g.at(i, j, k, RGPU_IU); double& mv = g.at(i, j, k, RGPU_IV);
        double& mw = g.at(i, j, k, RGPU_IW);
        const bool outer = std::fabs(zPos - zCenter) > outer_size * zSize;
        if (p_rand) {
          d = outer ? rho_outer : rho_inner;
          mu = d * ((outer ? vflow_out : vflow_in) + amplitude * draw());
          mv = d * (0.0 + amplitude * draw());
          mw = d * (0.0 + amplitude * draw());
        } else if (p_sine) {
          const double perturb_vy = 0, perturb_vz = amplitude * std::sin(2.0 * M_PI * xPos / xSize);
          d = outer ? rho_outer : rho_inner;
          mu = d * (outer ? vflow_out : vflow_in);
          mv = d * perturb_vy;
          mw = d * perturb_vz;
        } else if (p_robertson) {
          const int n = static_cast<int>(cfg.get_integer(S, "mode", 4));
          const double w0 = cfg.get_float(S, "w0", 0.1f), deltaZ = cfg.get_float(S, "deltaZ", 0.03f);
          const double z1 = p.zMin + 0.25 * zSize, z2 = p.zMin + 0.75 * zSize;
          const double ramp = 1.0 / (1.0 + std::exp(2 * (zPos - z1) / deltaZ)) + 1.0 / (1.0 + std::exp(2 * (z2 - zPos) / deltaZ));
          d = rho_inner + ramp * (rho_outer - rho_inner);
          mu = d * (vflow_in + ramp * (vflow_out - vflow_in));
          mv = d * w0 * std::cos(n * M_PI * xPos);
          mw = d * w0 * std::sin(n * M_PI * xPos);
        } else {
          continue;
        }
        g.at(i, j, k, RGPU_IP) = e0 + 0.5 * (sqr(mu) + sqr(mv) + sqr(mw)) / d;
      }
  }
}

// ---- MHD: Kelvin-Helmholtz (MHDRunBase.cpp:2814-2984) -- with its quirks: the switches are read with getFloat, the
// pressure from a section spelled "kelvin_helmholtz", 2D positions from j / jsize (x too), rand() drawn in any case ---
void init_mhd_kelvin_helmholtz(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const char* S = "kelvin-helmholtz";
  GlibcRand rng(static_cast<unsigned>(cfg.get_integer(S, "seed", 1)));
  const double amplitude = cfg.get_float(S, "amplitude", 0.01f);
  const bool p_sine = cfg.get_float(S, "perturbation_sine", 0.0f) != 0.0f;
  const bool p_rand = cfg.get_float(S, "perturbation_rand", 1.0f) != 0.0f;
  const double rho_inner = cfg.get_float(S, "rho_inner", 2.0f), rho_outer = cfg.get_float(S, "rho_outer", 1.0f);
  const double pressure = cfg.get_float("kelvin_helmholtz", "pressure", 2.5f);
  const double v0 = cfg.get_float(S, "v0", 1.0f), b0 = cfg.get_float(S, "b0", 1.0f);
  const double yMin = p.yMin, yMax = p.yMax, xMin = p.xMin, xMax = p.xMax;
  auto draw = [&]() { return 1.0 * rng.next() / GlibcRand::kRandMax - 0.5; };
  if (g.three_d) for (long n = 0; n < 3L * g.k_shift * g.ny * g.nx; ++n) rng.next();
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const double yPos = g.three_d ? yMin + p.dy / 2 + (j - g.gw) * p.dy : yMin + (yMax - yMin) * j / g.jsize;
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = g.three_d ? xMin + p.dx / 2 + (i - g.gw) * p.dx : xMin + (xMax - xMin) * j / g.jsize;
        const bool outer = yPos < yMin + 0.25 * (yMax - yMin) || yPos > yMin + 0.75 * (yMax - yMin);
        const double d = outer ? rho_outer : rho_inner;
        g.at(i, j, k, RGPU_ID) = d;
        g.at(i, j, k, RGPU_IU) = d * ((outer ? v0 : -v0) + p_rand * amplitude * draw() + p_sine * amplitude * std::sin(2 * M_PI * xPos));
        g.at(i, j, k, RGPU_IV) = d * (p_rand * amplitude * draw() + p_sine * amplitude * std::sin(2 * M_PI * xPos));
        if (g.three_d) g.at(i, j, k, RGPU_IW) = d * (p_rand * amplitude * draw() + p_sine * amplitude * std::sin(2 * M_PI * xPos));
        else g.at(i, j, k, RGPU_IW) = 0.0;
        g.at(i, j, k, RGPU_IA) = b0; g.at(i, j, k, RGPU_IB) = 0.0; g.at(i, j, k, RGPU_IC) = 0.0;
        const double mu = g.at(i, j, k, RGPU_IU), mv = g.at(i, j, k, RGPU_IV), mw = g.at(i, j, k, RGPU_IW);
        g.at(i, j, k, RGPU_IP) = pressure / (p.gamma0 - 1.0f) + 0.5 * (mu * mu + mv * mv + mw * mw) / d + 0.5 * b0 * b0;
      }
    }
}

// ---- hydro + MHD: Rayleigh-Taylor (HydroRunBase.cpp:6262-6434, MHDRunBase.cpp:2995-3037): every cell, ghosts
// included; the energy slot receives P0 + rho * (g . x) as written there; optional libc rand() perturbation ---------
void init_rayleigh_taylor(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const char* S = "rayleigh-taylor";
  const double amplitude = cfg.get_float(S, "amplitude", 0.01f);
  const double d0 = cfg.get_float(S, "d0", 1.0f), d1 = cfg.get_float(S, "d1", 2.0f);
  const bool randomEnabled = cfg.get_bool(S, "randomEnabled", false);
  GlibcRand rng(static_cast<unsigned>(cfg.get_integer(S, "random_seed", 33)));
  const double gx = p.gravity_x, gy = p.gravity_y, gz = p.gravity_z;
  const double P0 = 1.0f / (p.gamma0 - 1.0f);
  const double Lx = p.xMax - p.xMin, Ly = p.yMax - p.yMin, Lz = p.zMax - p.zMin;
  if (randomEnabled && g.three_d) for (long n = 0; n < (long)g.k_shift * g.jsize * g.isize; ++n) rng.next();
  for (int k = 0; k < g.ksize; ++k) {
    const double z = p.zMin + p.dz / 2 + (k + g.k_shift - g.gw) * p.dz;
    for (int j = 0; j < g.jsize; ++j) {
      const double y = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = 0; i < g.isize; ++i) {
        const double x = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        if (!g.three_d) {
          const double d = (y > (p.yMin + p.yMax) / 2) ? d1 : d0;
          g.at(i, j, 0, RGPU_ID) = d;
          g.at(i, j, 0, RGPU_IP) = P0 + d * (gx * x + gy * y);
          g.at(i, j, 0, RGPU_IU) = 0.0f;
          if (randomEnabled) g.at(i, j, 0, RGPU_IV) = amplitude * (rng.next() * 1.0 / GlibcRand::kRandMax - 0.5);
          else g.at(i, j, 0, RGPU_IV) = amplitude * (1 + std::cos(2 * M_PI * x / Lx)) * (1 + std::cos(2 * M_PI * y / Ly)) / 4;
        } else {
          const double d = (z > (p.zMin + p.zMax) / 2) ? d1 : d0;
          g.at(i, j, k, RGPU_ID) = d;
          g.at(i, j, k, RGPU_IP) = P0 + d * (gx * x + gy * y + gz * z);
          g.at(i, j, k, RGPU_IU) = 0.0f;
          g.at(i, j, k, RGPU_IV) = 0.0f;
          if (randomEnabled) g.at(i, j, k, RGPU_IW) = amplitude * (rng.next() * 1.0 / GlibcRand::kRandMax - 0.5);
          else g.at(i, j, k, RGPU_IW) = amplitude * (1 + std::cos(2 * M_PI * x / Lx)) * (1 + std::cos(2 * M_PI * y / Ly)) * (1 + std::cos(2 * M_PI * z / Lz)) / 8;
        }
        if (p.mhdEnabled) {
          const double Bx0 = cfg.get_float(S, "bx", 1e-8f), By0 = cfg.get_float(S, "by", 1e-8f), Bz0 = cfg.get_float(S, "bz", 1e-8f);
          g.at(i, j, k, RGPU_IA) = Bx0; g.at(i, j, k, RGPU_IB) = By0; g.at(i, j, k, RGPU_IC) = Bz0;
          g.at(i, j, k, RGPU_IP) += 0.5 * (Bx0 * Bx0 + By0 * By0 + Bz0 * Bz0);
        }
      }
    }
  }
}

// ---- hydro: the 19 two-dimensional Riemann problems of Lax & Liu, SIAM J. Sci. Comput. 19 (1998) 319-340
// (HydroRunBase.cpp:6798-6907; the state table is the paper's: {rho, u, v, p} of quadrants 1..4, single precision
// like the reference's initRiemannConfig2d) --------------------------------------------------------------------------
void init_hydro_riemann2d(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  static const float kLaxLiu[19][4][4] = {
    {{1.0f, 0.0f, 0.0f, 1.0f}, {0.5197f, -0.7259f, 0.0f, 0.4f}, {0.1072f, -0.7259f, -1.4045f, 0.0439f}, {0.2579f, 0.0f, -1.4045f, 0.15f}},   // configuration 1
    {{1.0f, 0.0f, 0.0f, 1.0f}, {0.5197f, -0.7259f, 0.0f, 0.4f}, {1.0f, -0.7259f, -0.7259f, 1.0f}, {0.5197f, 0.0f, -0.7259f, 0.4f}},   // configuration 2
    {{1.5f, 0.0f, 0.0f, 1.5f}, {0.5323f, 1.206f, 0.0f, 0.3f}, {0.138f, 1.206f, 1.206f, 0.029f}, {0.5323f, 0.0f, 1.206f, 0.3f}},   // configuration 3
    {{1.1f, 0.0f, 0.0f, 1.1f}, {0.5065f, 0.8939f, 0.0f, 0.35f}, {1.1f, 0.8939f, 0.8939f, 1.1f}, {0.5065f, 0.0f, 0.8939f, 0.35f}},   // configuration 4
    {{1.0f, -0.75f, -0.5f, 1.0f}, {2.0f, -0.75f, 0.5f, 1.0f}, {1.0f, 0.75f, 0.5f, 1.0f}, {3.0f, 0.75f, -0.5f, 1.0f}},   // configuration 5
    {{1.0f, 0.75f, -0.5f, 1.0f}, {2.0f, 0.75f, 0.5f, 0.5f}, {1.0f, -0.75f, 0.5f, 1.0f}, {3.0f, -0.75f, -0.5f, 1.0f}},   // configuration 6
    {{1.0f, 0.1f, 0.1f, 1.0f}, {0.5197f, -0.6259f, 0.1f, 0.4f}, {0.8f, 0.1f, 0.1f, 0.4f}, {0.5197f, 0.1f, -0.6259f, 0.4f}},   // configuration 7
    {{0.5197f, 0.1f, 0.1f, 0.4f}, {1.0f, -0.6259f, 0.1f, 1.0f}, {0.8f, 0.1f, 0.1f, 1.0f}, {1.0f, 0.1f, -0.6259f, 1.0f}},   // configuration 8
    {{1.0f, 0.0f, 0.3f, 1.0f}, {2.0f, 0.0f, -0.3f, 1.0f}, {1.039f, 0.0f, -0.8133f, 0.4f}, {0.5197f, 0.0f, -0.4259f, 0.4f}},   // configuration 9
    {{1.0f, 0.0f, 0.4297f, 1.0f}, {0.5f, 0.0f, 0.6076f, 1.0f}, {0.2281f, 0.0f, -0.6076f, 0.3333f}, {0.4562f, 0.0f, -0.4259f, 0.3333f}},   // configuration 10
    {{1.0f, 0.1f, 0.0f, 1.0f}, {0.5313f, 0.8276f, 0.0f, 0.4f}, {0.8f, 0.1f, 0.0f, 0.4f}, {0.5313f, 0.1f, 0.7276f, 0.4f}},   // configuration 11
    {{0.5313f, 0.0f, 0.0f, 0.4f}, {1.0f, 0.7276f, 0.0f, 1.0f}, {0.8f, 0.0f, 0.0f, 1.0f}, {1.0f, 0.0f, 0.7276f, 1.0f}},   // configuration 12
    {{1.0f, 0.0f, -0.3f, 1.0f}, {2.0f, 0.0f, 0.3f, 1.0f}, {1.0625f, 0.0f, 0.8145f, 0.4f}, {0.5313f, 0.0f, 0.4276f, 0.4f}},   // configuration 13
    {{2.0f, 0.0f, -0.5606f, 8.0f}, {1.0f, 0.0f, -1.2172f, 8.0f}, {0.4736f, 0.0f, 1.2172f, 2.6667f}, {0.9474f, 0.0f, 1.1606f, 2.6667f}},   // configuration 14
    {{1.0f, 0.1f, -0.3f, 1.0f}, {0.5197f, -0.6259f, -0.3f, 0.4f}, {0.8f, 0.1f, -0.3f, 0.4f}, {0.5313f, 0.1f, 0.4276f, 0.4f}},   // configuration 15
    {{0.5313f, 0.1f, 0.1f, 0.4f}, {1.0222f, -0.6179f, 0.1f, 1.0f}, {0.8f, 0.1f, 0.1f, 1.0f}, {1.0f, 0.1f, 0.8276f, 1.0f}},   // configuration 16
    {{1.0f, 0.0f, -0.4f, 1.0f}, {2.0f, 0.0f, -0.3f, 1.0f}, {1.0625f, 0.0f, 0.2145f, 0.4f}, {0.5197f, 0.0f, -1.1259f, 0.4f}},   // configuration 17
    {{1.0f, 0.0f, 1.0f, 1.0f}, {2.0f, 0.0f, -0.3f, 1.0f}, {1.0625f, 0.0f, 0.2145f, 0.4f}, {0.5197f, 0.0f, 0.2741f, 0.4f}},   // configuration 18
    {{1.0f, 0.0f, 0.3f, 1.0f}, {2.0f, 0.0f, -0.3f, 1.0f}, {1.0625f, 0.0f, 0.2145f, 0.4f}, {0.5197f, 0.0f, -0.4259f, 0.4f}},   // configuration 19
  };
  if (g.three_d) throw std::runtime_error("riemann2d is a 2D problem");
  int nb = static_cast<int>(cfg.get_integer("hydro", "riemann_config_number", 0));
  if (nb < 0) nb = 0; else if (nb > 18) nb = 18;
  const double xt = cfg.get_float("riemann2d", "x", 0.5f), yt = cfg.get_float("riemann2d", "y", 0.5f);
  double q[4][4];   // conservative {rho, E, rho u, rho v} of quadrants 1..4 (primToCons_2D, constoprim.h:221-234)
  for (int n = 0; n < 4; ++n) {
    const double rho = kLaxLiu[nb][n][0], u = kLaxLiu[nb][n][1], v = kLaxLiu[nb][n][2], pr = kLaxLiu[nb][n][3];
    q[n][RGPU_ID] = rho;
    q[n][RGPU_IU] = u * rho;
    q[n][RGPU_IV] = v * rho;
    q[n][RGPU_IP] = pr / (p.gamma0 - 1.0f) + rho * (u * u + v * v) * 0.5f;
  }
  for (int j = g.gw; j < g.jsize - g.gw; ++j) {
    const double y = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
    for (int i = g.gw; i < g.isize - g.gw; ++i) {
      const double x = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
      const int quad = (x < xt) ? ((y < yt) ? 2 : 1) : ((y < yt) ? 3 : 0);
      for (int v = 0; v < 4; ++v) g.at(i, j, 0, v) = q[quad][v];
    }
  }
}

// ---- hydro: Gresho vortex (HydroRunBase.cpp:5688-5838) ------------------------------------------------------------
void init_hydro_gresho(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const char* S = "Gresho_vortex";
  const double center_x = cfg.get_float(S, "center_x", (float)((p.xMax + p.xMin) / 2));
  const double center_y = cfg.get_float(S, "center_y", (float)((p.yMax + p.yMin) / 2));
  const double vbx = cfg.get_float(S, "v_bulk_x", 0.0f), vby = cfg.get_float(S, "v_bulk_y", 0.0f), vbz = cfg.get_float(S, "v_bulk_z", 0.0f);
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        const double r = std::sqrt((xPos - center_x) * (xPos - center_x) + (yPos - center_y) * (yPos - center_y));
        const double phi = std::atan2(yPos - center_y, xPos - center_x);
        double P, v_phi;
        if (r < 0.2) { P = 5 + 12.5 * r * r; v_phi = 5 * r; }
        else if (r < 0.4) { P = 9 + 12.5 * r * r - 20 * r + 4 * std::log(5 * r); v_phi = 2 - 5 * r; }
        else { P = 3 + 4 * std::log(2); v_phi = 0.0; }
        const double mu = -std::sin(phi) * v_phi + vbx, mv = std::cos(phi) * v_phi + vby;
        g.at(i, j, k, RGPU_ID) = 1.0;
        g.at(i, j, k, RGPU_IU) = mu;
        g.at(i, j, k, RGPU_IV) = mv;
        if (g.three_d) {
          g.at(i, j, k, RGPU_IW) = vbz;
          g.at(i, j, k, RGPU_IP) = P / (p.gamma0 - 1.0f) + 0.5 * (mu * mu + mv * mv + vbz * vbz) / 1.0;
        } else {
          g.at(i, j, k, RGPU_IP) = P / (p.gamma0 - 1.0f) + 0.5 * (mu * mu + mv * mv) / 1.0;
        }
      }
    }
}

// ---- hydro: falling bubble, 2D (HydroRunBase.cpp:6633-6714; the 3D branch of the reference writes the density of
// the k=0 plane only and is not reproduced) ------------------------------------------------------------------------
void init_hydro_falling_bubble(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  if (g.three_d) throw std::runtime_error("falling-bubble in 3D is outside the implemented scope");
  const char* S = "falling-bubble";
  const double P0 = 1.0f / (p.gamma0 - 1.0f);
  const double Ly = p.yMax - p.yMin;
  const double radius = cfg.get_float(S, "radius", 0.1f);
  const double x_c = cfg.get_float(S, "center_x", (float)((p.xMin + p.xMax) / 2));
  const double y_c = cfg.get_float(S, "center_y", (float)(p.yMin + 0.8 * Ly));
  const double v0 = cfg.get_float(S, "v0", 0.0f), d0 = cfg.get_float(S, "d0", 2.0f), d1 = cfg.get_float(S, "d1", 1.0f);
  for (int j = 0; j < g.jsize; ++j) {
    const double y = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
    for (int i = 0; i < g.isize; ++i) {
      const double x = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
      double d = (y < p.yMin + 0.3 * Ly) ? d0 : d1;
      const double r2 = (x - x_c) * (x - x_c) + (y - y_c) * (y - y_c);
      if (r2 < radius * radius) d = d0;
      g.at(i, j, 0, RGPU_ID) = d;
      g.at(i, j, 0, RGPU_IP) = P0 + d * (p.gravity_x * x + p.gravity_y * y);
      g.at(i, j, 0, RGPU_IU) = 0.0;
      g.at(i, j, 0, RGPU_IV) = (r2 < radius * radius) ? v0 : 0.0;
    }
  }
}

// ---- hydro: Keplerian disk around a softened point mass (HydroRunBase.cpp:6445-6625), every cell ------------------
// The gravity field of the same routine is produced by init_gravity_field below.
struct KeplerDisk {
  double epsilon, P0, xCenter, yCenter, grav;
  KeplerDisk(const IniConfig& cfg, const rgpu_params& p) {
    const double xMax = p.xMax, yMax = p.yMax;
    epsilon = cfg.get_float("Keplerian-disk", "epsilon", 0.01f);
    P0 = cfg.get_float("Keplerian-disk", "pressure", 1e-6f);
    xCenter = cfg.get_float("Keplerian-disk", "xCenter", static_cast<float>((xMax + p.xMin) / 2.0));
    yCenter = cfg.get_float("Keplerian-disk", "yCenter", static_cast<float>((yMax + p.yMin) / 2.0));
    grav = cfg.get_float("gravity", "g", 1.0f);
  }
};

void init_hydro_keplerian_disk(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const KeplerDisk kd(cfg, p);
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j) {
      const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = 0; i < g.isize; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        const double theta = std::atan2(yPos - kd.yCenter, xPos - kd.xCenter);
        const double r = std::sqrt((xPos - kd.xCenter) * (xPos - kd.xCenter) + (yPos - kd.yCenter) * (yPos - kd.yCenter));
        const double velocity = r * std::pow(r * r + kd.epsilon * kd.epsilon, -3.0 / 4.0);
        double d;
        if (r < 0.5) d = 0.01 + std::pow(r / 0.5, 3.0);
        else if (r <= 2) d = 0.01 + 1;
        else d = 0.01 + std::pow(1 + (r - 2) / 0.1, -3.0);
        g.at(i, j, k, RGPU_ID) = d;
        // the reference binary (g++ -O2) evaluates sin(theta) and cos(theta) with ONE call of glibc's sincos(), whose
        // sine differs from sin()'s in the last bit for some arguments: make the same call
        double sin_t, cos_t;
        ::sincos(theta, &sin_t, &cos_t);
        g.at(i, j, k, RGPU_IU) = -sin_t * velocity * d;
        g.at(i, j, k, RGPU_IV) = cos_t * velocity * d;
        const double mu = g.at(i, j, k, RGPU_IU), mv = g.at(i, j, k, RGPU_IV);
        g.at(i, j, k, RGPU_IP) = kd.P0 / (p.gamma0 - 1.0) + 0.5 * (mu * mu + mv * mv) / d;   // (+ 0*0 of the z momentum in 3D)
      }
    }
}

// ---- turbulence: static solenoidal driving field (turbulenceInit.cpp, after Enzo's turboinit.f) -----------------------
// 16 Fourier modes with fixed amplitudes and phases (the tables below are the reference's data); the y and z phases of
// the four diagonal modes are corrected so that the field is solenoidal.  F: 3 * ncell doubles, every cell.
const int kTurbModes = 16;
const int kTurbMode[kTurbModes][3] = {{1, 1, 1}, {-1, 1, 1}, {1, -1, 1}, {1, 1, -1}, {0, 0, 1}, {0, 1, 0}, {1, 0, 0}, {0, 1, 1},
                                      {1, 0, 1}, {1, 1, 0}, {0, -1, 1}, {-1, 0, 1}, {-1, 1, 0}, {0, 0, 2}, {0, 2, 0}, {2, 0, 0}};
const double kTurbPhaX[kTurbModes] = {4.88271710, 4.55016280, 3.68972560, 5.76067300, 2.02647730, 0.832007770, 1.93749010, 0.0141755510,
                                      5.13556960, 2.77787590, 2.02909450, 0.663769130, 1.80512500, 3.31305960, 1.05063310, 1.75230850};
const double kTurbPhaY[kTurbModes] = {1.40113130, 5.71809960, 3.82072880, 1.00265060, 2.26816680, 2.81446220, 0.990584490, 2.94580650,
                                      3.92715640, 0.896237970, 1.85357800, 2.84606100, 1.63463330, 3.46619220, 5.58599570, 1.59481430};
const double kTurbPhaZ[kTurbModes] = {5.60595510, 4.13909050, 6.22733640, 5.92633250, 3.51874880, 5.42229180, 5.77061890, 4.95180180,
                                      4.46144340, 5.29367540, 5.50741860, 2.39496800, 4.59486870, 2.23851540, 3.19591550, 4.47066500};
const double kTurbAmp[3][kTurbModes] = {
    {0.0755957220, -1.35724380, 0.378455820, -0.383104000, 0.116980840, -1.16079680, 0.0, -0.0280965080,
     0.0, 0.0, -0.232798780, 0.0, 0.0, -0.879534360, -0.604585950, 0.0},
    {1.03223790, 0.530986910, -0.242943420, -0.832715270, -0.607103350, 0.0, -0.278135540, 0.0,
     -1.18019080, 0.0, 0.0, 0.976678430, 0.0, -0.694509390, 0.0, -0.608007610},
    {1.01825800, -0.966076610, 0.211956020, -0.605923650, 0.0, 0.314906060, 0.109417880, 0.0,
     0.0, -1.53612340, 0.0, 0.0, 0.813212160, 0.0, -0.368619380, -0.371489380}};

void turbulence_field(const rgpu_params& p, const Grid& g, double mach, double* F) {
  static const double sign1[4] = {1.0, -1.0, -1.0, 1.0}, sign2[4] = {-1.0, -1.0, 1.0, 1.0};
  const double pi = 2.0 * std::asin(1.0);
  const double aa = 2.0 * pi / p.nx;   // "nbox" = nx (HydroRunBase.cpp:7199-7204)
  double* u = F; double* v = F + g.ncell; double* w = F + 2 * g.ncell;
  const int off = -g.gw;
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j)
      for (int i = 0; i < g.isize; ++i) {
        const size_t index = static_cast<size_t>(i) + static_cast<size_t>(g.isize) * (j + static_cast<size_t>(g.jsize) * k);
        u[index] = 0.0; v[index] = 0.0; w[index] = 0.0;
        for (int m = 0; m < kTurbModes; ++m) {
          const double k1 = kTurbMode[m][0] * (i + off + 1) + kTurbMode[m][1] * (j + off + 1) + kTurbMode[m][2] * (k + g.k_shift + off + 1);
          const double ax = kTurbAmp[0][m], ay = kTurbAmp[1][m], az = kTurbAmp[2][m];
          u[index] = u[index] + ax * std::cos(aa * k1 + kTurbPhaX[m]);
          if (m < 4) {
            const double phayy = kTurbPhaX[m] + sign1[m] * std::acos((az * az - ax * ax - ay * ay) / 2.0 / ax / kTurbMode[m][0] / kTurbMode[m][1] / ay);
            v[index] = v[index] + ay * std::cos(aa * k1 + phayy);
            const double phazz = kTurbPhaX[m] + sign2[m] * std::acos((ay * ay - ax * ax - az * az) / 2.0 / ax / kTurbMode[m][0] / kTurbMode[m][2] / az);
            w[index] = w[index] + az * std::cos(aa * k1 + phazz);
          } else {
            v[index] = v[index] + ay * std::cos(aa * k1 + kTurbPhaY[m]);
            w[index] = w[index] + az * std::cos(aa * k1 + kTurbPhaZ[m]);
          }
        }
        // normalisation to the requested rms Mach number
        u[index] = u[index] / 2.848320 * mach;
        v[index] = v[index] / 2.848320 * mach;
        w[index] = w[index] / 2.848320 * mach;
      }
}

// problem "turbulence" (HydroRunBase.cpp:6916-6964, MHDRunBase.cpp:3045-3098): perturbed density (libc rand()), the
// driving field as initial velocity, uniform pressure; MHD adds a uniform field
void init_turbulence(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  if (!g.three_d) throw std::runtime_error("the turbulence problem is not available in 2D");
  const double d0 = cfg.get_float("turbulence", "density", 1.0f);
  const double ampl = cfg.get_float("turbulence", "initialDensityPerturbationAmplitude", 0.0f);
  const double P0 = cfg.get_float("turbulence", "pressure", 1.0f);
  GlibcRand rng(static_cast<unsigned>(cfg.get_integer("turbulence", "random_seed", 33)));
  for (long n = 0; n < (long)g.k_shift * g.ny * g.nx; ++n) rng.next();   // interior cells of the slabs below
  std::vector<double> F(3 * g.ncell);
  turbulence_field(p, g, cfg.get_float("turbulence", "machNumber", 0.0f), F.data());
  double Bx0 = 0, By0 = 0, Bz0 = 0;
  if (p.mhdEnabled) {
    Bx0 = cfg.get_float("turbulence", "bx", 1e-8f);
    By0 = cfg.get_float("turbulence", "by", 1e-8f);
    Bz0 = cfg.get_float("turbulence", "bz", 1e-8f);
    const double beta = cfg.get_float("turbulence", "beta", 0.0f);
    if (beta > 0) {
      const double cIso2 = p.cIso * p.cIso;
      Bx0 = std::sqrt(2 * cIso2 * d0 / beta);
      By0 = 0.0; Bz0 = 0.0;
      if (cIso2 <= 0.0) Bx0 = cfg.get_float("turbulence", "Bx0", static_cast<float>(2.0 * d0 / beta));
    }
  }
  for (int k = g.gw; k < g.ksize - g.gw; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const size_t o = static_cast<size_t>(i) + static_cast<size_t>(g.isize) * (j + static_cast<size_t>(g.jsize) * k);
        const double d = d0 * (1.0 + ampl * ((float)rng.next() / (float)(GlibcRand::kRandMax) - 0.5));
        g.at(i, j, k, RGPU_ID) = d;
        g.at(i, j, k, RGPU_IU) = d * F[o];
        g.at(i, j, k, RGPU_IV) = d * F[o + g.ncell];
        g.at(i, j, k, RGPU_IW) = d * F[o + 2 * g.ncell];
        const double mu = g.at(i, j, k, RGPU_IU), mv = g.at(i, j, k, RGPU_IV), mw = g.at(i, j, k, RGPU_IW);
        g.at(i, j, k, RGPU_IP) = P0 / (p.gamma0 - 1.0) + 0.5 * (mu * mu + mv * mv + mw * mw) / d;
        if (p.mhdEnabled) {
          g.at(i, j, k, RGPU_IA) = Bx0; g.at(i, j, k, RGPU_IB) = By0; g.at(i, j, k, RGPU_IC) = Bz0;
          g.at(i, j, k, RGPU_IP) += 0.5 * (Bx0 * Bx0 + By0 * By0 + Bz0 * Bz0);
        }
      }
}

// problem "turbulence-Ornstein-Uhlenbeck" (HydroRunBase.cpp:6973-7019, MHDRunBase.cpp:3107-3159): gas at rest, density
// perturbed with libc rand() (double arithmetic here, unlike "turbulence"), uniform pressure; MHD adds a uniform field.
// The forcing process itself lives in the context (rgpu_params::ouForcingEnabled).
void init_turbulence_ou(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const char* S = "turbulence-Ornstein-Uhlenbeck";
  if (!g.three_d) throw std::runtime_error("the turbulence-Ornstein-Uhlenbeck problem is not available in 2D");
  const double d0 = cfg.get_float(S, "density", 1.0f);
  const double ampl = cfg.get_float(S, "initialDensityPerturbationAmplitude", 0.0f);
  const double P0 = cfg.get_float(S, "pressure", 1.0f);
  GlibcRand rng(static_cast<unsigned>(cfg.get_integer(S, "random_seed", 33)));
  for (long n = 0; n < (long)g.k_shift * g.ny * g.nx; ++n) rng.next();   // interior cells of the slabs below
  double Bx0 = 0, By0 = 0, Bz0 = 0;
  if (p.mhdEnabled) {
    Bx0 = cfg.get_float(S, "bx", 1e-8f);
    By0 = cfg.get_float(S, "by", 1e-8f);
    Bz0 = cfg.get_float(S, "bz", 1e-8f);
    const double beta = cfg.get_float(S, "beta", 0.0f);
    if (beta > 0) {
      const double cIso2 = p.cIso * p.cIso;
      Bx0 = std::sqrt(2 * cIso2 * d0 / beta);
      By0 = 0.0; Bz0 = 0.0;
      if (cIso2 <= 0.0) Bx0 = cfg.get_float(S, "Bx0", static_cast<float>(2.0 * d0 / beta));
    }
  }
  for (int k = g.gw; k < g.ksize - g.gw; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        g.at(i, j, k, RGPU_ID) = d0 * (1.0 + ampl * ((1.0 * rng.next()) / GlibcRand::kRandMax - 0.5));
        g.at(i, j, k, RGPU_IU) = 0.0; g.at(i, j, k, RGPU_IV) = 0.0; g.at(i, j, k, RGPU_IW) = 0.0;
        g.at(i, j, k, RGPU_IP) = P0 / (p.gamma0 - 1.0);
        if (p.mhdEnabled) {
          g.at(i, j, k, RGPU_IA) = Bx0; g.at(i, j, k, RGPU_IB) = By0; g.at(i, j, k, RGPU_IC) = Bz0;
          g.at(i, j, k, RGPU_IP) += 0.5 * (Bx0 * Bx0 + By0 * By0 + Bz0 * Bz0);
        }
      }
}

// ---- MHD: compressive shear wave in the shearing box (MHDRunBase.cpp:2574-2658), every cell, ghosts included --------
void init_mhd_shear_wave(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  if (!(p.bc[0] == RGPU_BC_SHEARINGBOX && p.bc[1] == RGPU_BC_SHEARINGBOX))
    throw std::runtime_error("ShearWave needs shearing box conditions along x (boundary_xmin=boundary_xmax=4)");
  const double TwoPi = 4.0 * std::asin(1.0);
  const double d0 = 1.0;
  const double Lx = p.dx * p.nx, Ly = p.dy * p.ny;
  const double energy = cfg.get_float("ShearWave", "energy", 1.0f);
  const double delta_vx = (-4.0e-4) * p.cIso, delta_vy = (1.0e-4) * p.cIso;
  const double kx0 = -4 * TwoPi / Lx, ky0 = TwoPi / Ly;
  const double xi0 = 0.5 * p.Omega0 / d0;
  const double delta_rho = (kx0 * delta_vy - ky0 * delta_vx) / xi0;
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j) {
      const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
      for (int i = 0; i < g.isize; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        const double d = d0 * (1.0 - delta_rho * std::sin(kx0 * xPos + ky0 * yPos));
        g.at(i, j, k, RGPU_ID) = d;
        g.at(i, j, k, RGPU_IP) = energy;
        g.at(i, j, k, RGPU_IU) = d * delta_vx * std::cos(kx0 * xPos + ky0 * yPos);
        g.at(i, j, k, RGPU_IV) = d * delta_vy * std::cos(kx0 * xPos + ky0 * yPos);
      }
    }
}

// ---- MHD: inertial wave in the rotating frame (MHDRunBase.cpp:2503-2558): uniform state, every cell, zero field -----
void init_mhd_inertial_wave(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const double density = cfg.get_float("InertialWave", "density", 1.0f);
  const double energy = cfg.get_float("InertialWave", "energy", 1.0f);
  double delta_vx = cfg.get_float("InertialWave", "delta_vx", 1.0f);
  delta_vx *= p.cIso;
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j)
      for (int i = 0; i < g.isize; ++i) {
        g.at(i, j, k, RGPU_ID) = density;
        g.at(i, j, k, RGPU_IP) = energy;
        g.at(i, j, k, RGPU_IU) = density * delta_vx;
      }
}

// ---- MHD: jet medium with an optional static field (MHDRunBase.cpp:1747-1798) and Sod tube (:1806-1862) ----------
void init_mhd_jet(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const double Bx = cfg.get_float("jet", "BStatic_x", 0.0f), By = cfg.get_float("jet", "BStatic_y", 0.0f);
  const double Bz = cfg.get_float("jet", "BStatic_z", 0.0f);
  const int k0 = g.three_d ? g.gw : 0, k1 = g.three_d ? g.ksize - g.gw : 1;
  for (int k = k0; k < k1; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j)
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        g.at(i, j, k, RGPU_ID) = 1.0f;
        g.at(i, j, k, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f) + (g.three_d ? 0.5 * (Bx * Bx + By * By + Bz * Bz) : 0.5 * (Bx * Bx + By * By));
        g.at(i, j, k, RGPU_IU) = 0.0f; g.at(i, j, k, RGPU_IV) = 0.0f; g.at(i, j, k, RGPU_IW) = 0.0f;
        g.at(i, j, k, RGPU_IA) = Bx; g.at(i, j, k, RGPU_IB) = By; g.at(i, j, k, RGPU_IC) = Bz;
      }
}

void init_mhd_sod(const rgpu_params& p, const Grid& g) {
  if (g.three_d) throw std::runtime_error("MHD sod in 3D: the reference writes the k=0 plane only (2D accessors in its 3D branch)");
  for (int j = g.gw; j < g.jsize - g.gw; ++j)
    for (int i = g.gw; i < g.isize - g.gw; ++i) {
      if (i < g.isize / 2) { g.at(i, j, 0, RGPU_ID) = 1.0f; g.at(i, j, 0, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f); }
      else { g.at(i, j, 0, RGPU_ID) = 0.125f; g.at(i, j, 0, RGPU_IP) = 0.1f / (p.gamma0 - 1.0f); }
    }
}

// ---- MHD: rotor, 2D only (MHDRunBase.cpp:2117-2189) -- including the velocity (not momentum) written to the
// momentum slots and the dx/2 offset of yPos ----------------------------------------------------------------------
void init_mhd_rotor(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  if (g.three_d) return;   // the reference's 3D branch is empty: the state stays zero
  const double FourPi = 8.0 * std::asin(1.0);
  const double r0 = cfg.get_float("rotor", "r0", 0.1f);
  const double r1 = cfg.get_float("rotor", "r1", 0.115f);
  const double u0 = cfg.get_float("rotor", "u0", 2.0f);
  const double p0 = cfg.get_float("rotor", "p0", 1.0f);
  const double b0 = cfg.get_float("rotor", "b0", (float)(5.0 / std::sqrt(FourPi)));
  const double xMax = cfg.get_float("mesh", "xmax", 1.0f), yMax = cfg.get_float("mesh", "ymax", 1.0f);
  const double xCenter = (xMax + p.xMin) / 2, yCenter = (yMax + p.yMin) / 2;
  const double gamma = p.gamma0;
  for (int j = g.gw; j < g.jsize - g.gw; ++j) {
    const double yPos = p.yMin + p.dx / 2 + (j - g.gw) * p.dy;
    for (int i = g.gw; i < g.isize - g.gw; ++i) {
      const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
      const double r = std::sqrt((xPos - xCenter) * (xPos - xCenter) + (yPos - yCenter) * (yPos - yCenter));
      const double f_r = (r1 - r) / (r1 - r0);
      if (r <= r0) {
        g.at(i, j, 0, RGPU_ID) = 10.0;
        g.at(i, j, 0, RGPU_IU) = -u0 * (yPos - yCenter) / r0;
        g.at(i, j, 0, RGPU_IV) = u0 * (xPos - xCenter) / r0;
      } else if (r <= r1) {
        g.at(i, j, 0, RGPU_ID) = 1 + 9 * f_r;
        g.at(i, j, 0, RGPU_IU) = -f_r * u0 * (yPos - yCenter) / r;
        g.at(i, j, 0, RGPU_IV) = f_r * u0 * (xPos - xCenter) / r;
      } else {
        g.at(i, j, 0, RGPU_ID) = 1.0;
        g.at(i, j, 0, RGPU_IU) = 0.0;
        g.at(i, j, 0, RGPU_IV) = 0.0;
      }
      g.at(i, j, 0, RGPU_IW) = 0.0;
      g.at(i, j, 0, RGPU_IA) = b0;
      g.at(i, j, 0, RGPU_IB) = 0.0;
      g.at(i, j, 0, RGPU_IC) = 0.0;
      const double mu = g.at(i, j, 0, RGPU_IU), mv = g.at(i, j, 0, RGPU_IV), mw = g.at(i, j, 0, RGPU_IW);
      g.at(i, j, 0, RGPU_IP) = p0 / (gamma - 1.0) + (mu * mu + mv * mv + mw * mw) / 2 / g.at(i, j, 0, RGPU_ID) +
                               (g.at(i, j, 0, RGPU_IA) * g.at(i, j, 0, RGPU_IA)) / 2;
    }
  }
}

// ---- MHD: field loop advection (MHDRunBase.cpp:2214-2408).  3D: the z component of the vector potential carries
// drand48 noise, one draw per cell of the whole ghost-inclusive box in k,j,i order -------------------------------
void init_mhd_field_loop(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const double radius = cfg.get_float("FieldLoop", "radius", 1.0f);
  const double density_in = cfg.get_float("FieldLoop", "density_in", 1.0f);
  const double amplitude = cfg.get_float("FieldLoop", "amplitude", 1.0f);
  const double vflow = cfg.get_float("FieldLoop", "vflow", 1.0f);
  const double cos_theta = 2.0 / std::sqrt(5.0);
  const double sin_theta = std::sqrt(1 - cos_theta * cos_theta);
  const double dx = p.dx, dy = p.dy, dz = p.dz;
  if (!g.three_d) {
    std::vector<double> Az(static_cast<size_t>(g.isize) * g.jsize, 0.0);
    auto az = [&](int i, int j) -> double& { return Az[static_cast<size_t>(i) + static_cast<size_t>(g.isize) * j]; };
    for (int j = g.gw; j < g.jsize - g.gw + 1; ++j) {
      const double yPos = p.yMin + dy / 2 + (j - g.gw) * dy;
      for (int i = g.gw; i < g.isize - g.gw + 1; ++i) {
        const double xPos = p.xMin + dx / 2 + (i - g.gw) * dx;
        const double r = std::sqrt(xPos * xPos + yPos * yPos);
        az(i, j) = (r < radius) ? amplitude * (radius - r) : 0.0;
      }
    }
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const double yPos = p.yMin + dy / 2 + (j - g.gw) * dy;
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = p.xMin + dx / 2 + (i - g.gw) * dx;
        const double diag = std::sqrt(1.0 * (p.nx * p.nx + p.ny * p.ny + p.nz * p.nz));   // nz = 1 in 2D
        const double r = std::sqrt(xPos * xPos + yPos * yPos);
        const double d = (r < radius) ? density_in : 1.0f;
        g.at(i, j, 0, RGPU_ID) = d;
        g.at(i, j, 0, RGPU_IU) = d * vflow * cos_theta;
        g.at(i, j, 0, RGPU_IV) = d * vflow * sin_theta;
        g.at(i, j, 0, RGPU_IW) = d * vflow * p.nz / diag;
        g.at(i, j, 0, RGPU_IA) = (az(i, j + 1) - az(i, j)) / dy;
        g.at(i, j, 0, RGPU_IB) = -(az(i + 1, j) - az(i, j)) / dx;
        g.at(i, j, 0, RGPU_IC) = 0.0;
        const double A = g.at(i, j, 0, RGPU_IA), B = g.at(i, j, 0, RGPU_IB);
        const double mu = g.at(i, j, 0, RGPU_IU), mv = g.at(i, j, 0, RGPU_IV);
        g.at(i, j, 0, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f) + 0.5 * (A * A + B * B) + 0.5 * (mu * mu + mv * mv) / d;
      }
    }
    return;
  }
  const double amp = cfg.get_float("FieldLoop", "amp", 0.01f);
  const long seed = cfg.get_integer("FieldLoop", "seed", 0);
  Rand48 rng(seed);
  rng.skip(static_cast<unsigned long long>(g.k_shift) * g.jsize * g.isize);   // planes of the slabs below
  // only A_z is non-zero; local plane k is global plane k + k_shift
  std::vector<double> Az(static_cast<size_t>(g.isize) * g.jsize * g.ksize, 0.0);
  auto az = [&](int i, int j, int k) -> double& {
    return Az[static_cast<size_t>(i) + static_cast<size_t>(g.isize) * (j + static_cast<size_t>(g.jsize) * k)];
  };
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j) {
      const double yPos = p.yMin + dy / 2 + (j - g.gw) * dy;
      for (int i = 0; i < g.isize; ++i) {
        const double xPos = p.xMin + dx / 2 + (i - g.gw) * dx;
        az(i, j, k) = 0.0 + amp * (rng.next() - 0.5);
        const double r = std::sqrt(xPos * xPos + yPos * yPos);
        if (r < radius) az(i, j, k) = amplitude * (radius - r);
      }
    }
  for (int k = g.gw; k < g.ksize - g.gw; ++k)
    for (int j = g.gw; j < g.jsize - g.gw; ++j) {
      const double yPos = p.yMin + dy / 2 + (j - g.gw) * dy;
      for (int i = g.gw; i < g.isize - g.gw; ++i) {
        const double xPos = p.xMin + dx / 2 + (i - g.gw) * dx;
        const double r = std::sqrt(xPos * xPos + yPos * yPos);
        const double d = (r < radius) ? density_in : 1.0f;
        g.at(i, j, k, RGPU_ID) = d;
        g.at(i, j, k, RGPU_IU) = d * vflow * cos_theta;
        g.at(i, j, k, RGPU_IV) = d * vflow * sin_theta;
        g.at(i, j, k, RGPU_IW) = 0.0;
        // curl of (0, 0, Az); the A_x = A_y = 0 terms are kept as the reference writes them
        g.at(i, j, k, RGPU_IA) = (az(i, j + 1, k) - az(i, j, k)) / dy - (0.0 - 0.0) / dz;
        g.at(i, j, k, RGPU_IB) = (0.0 - 0.0) / dz - (az(i + 1, j, k) - az(i, j, k)) / dx;
        g.at(i, j, k, RGPU_IC) = (0.0 - 0.0) / dx - (0.0 - 0.0) / dy;
        if (p.cIso > 0) {
          g.at(i, j, k, RGPU_IP) = 0.0;
        } else {
          const double A = g.at(i, j, k, RGPU_IA), B = g.at(i, j, k, RGPU_IB), C = g.at(i, j, k, RGPU_IC);
          const double mu = g.at(i, j, k, RGPU_IU), mv = g.at(i, j, k, RGPU_IV), mw = g.at(i, j, k, RGPU_IW);
          g.at(i, j, k, RGPU_IP) = 1.0f / (p.gamma0 - 1.0f) + 0.5 * (A * A + B * B + C * C) + 0.5 * (mu * mu + mv * mv + mw * mw) / d;
        }
      }
    }
}

// ---- MHD: current sheet (MHDRunBase.cpp:2424-2487): every cell, ghosts included; the energy slot holds beta -----
void init_mhd_current_sheet(const IniConfig& cfg, const rgpu_params& p, const Grid& g) {
  const double A = cfg.get_float("CurrentSheet", "A", 0.1f);
  const double B0 = cfg.get_float("CurrentSheet", "B0", 1.0f);
  const double beta = cfg.get_float("CurrentSheet", "beta", 0.1f);
  for (int k = 0; k < g.ksize; ++k)
    for (int j = 0; j < g.jsize; ++j)
      for (int i = 0; i < g.isize; ++i) {
        const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
        const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
        g.at(i, j, k, RGPU_ID) = 1.0;
        g.at(i, j, k, RGPU_IP) = beta;
        g.at(i, j, k, RGPU_IU) = 1.0 * A * std::sin(M_PI * yPos);
        g.at(i, j, k, RGPU_IV) = 0.0;
        g.at(i, j, k, RGPU_IW) = 0.0;
        g.at(i, j, k, RGPU_IA) = 0.0;
        g.at(i, j, k, RGPU_IB) = (xPos < 0.5 || xPos > 1.5) ? B0 : -B0;
        g.at(i, j, k, RGPU_IC) = 0.0;
      }
}

}  // namespace

// The per-cell static gravity field of the problems that define one (gravityEnabled == 2): hG[3][ksize][jsize][isize]
bool init_gravity_field(const IniConfig& cfg, const rgpu_params& p, double* hG) {
  Grid g = make_grid(p, hG);
  g.nvar = 3;
  std::memset(hG, 0, sizeof(double) * g.ncell * 3);
  const std::string problem = cfg.get_string("hydro", "problem", "unknown");
  if (!p.mhdEnabled && problem == "Keplerian-disk") {
    // g = -grad(Phi), Phi = -(r^2 + epsilon^2)^(-1/2): analytic in 2D (with the reference's xPos, yPos measured from
    // the origin, not from the disk centre), one-sided differences of Phi in 3D (HydroRunBase.cpp:6489-6500, 6575-6597)
    const KeplerDisk kd(cfg, p);
    for (int k = 0; k < g.ksize; ++k)
      for (int j = 0; j < g.jsize; ++j) {
        const double yPos = p.yMin + p.dy / 2 + (j - g.gw) * p.dy;
        for (int i = 0; i < g.isize; ++i) {
          const double xPos = p.xMin + p.dx / 2 + (i - g.gw) * p.dx;
          const double r = std::sqrt((xPos - kd.xCenter) * (xPos - kd.xCenter) + (yPos - kd.yCenter) * (yPos - kd.yCenter));
          if (!g.three_d) {
            const double dphi_dx = xPos * std::pow(r * r + kd.epsilon * kd.epsilon, -3.0 / 2);
            const double dphi_dy = yPos * std::pow(r * r + kd.epsilon * kd.epsilon, -3.0 / 2);
            g.at(i, j, k, 0) = -kd.grav * dphi_dx;
            g.at(i, j, k, 1) = -kd.grav * dphi_dy;
          } else {
            const double phi = -1.0 / std::sqrt(r * r + kd.epsilon * kd.epsilon);
            const double r_x = std::sqrt((xPos + p.dx - kd.xCenter) * (xPos + p.dx - kd.xCenter) + (yPos - kd.yCenter) * (yPos - kd.yCenter));
            const double phi_x = -1.0 / std::sqrt(r_x * r_x + kd.epsilon * kd.epsilon);
            const double r_y = std::sqrt((xPos - kd.xCenter) * (xPos - kd.xCenter) + (yPos + p.dy - kd.yCenter) * (yPos + p.dy - kd.yCenter));
            const double phi_y = -1.0 / std::sqrt(r_y * r_y + kd.epsilon * kd.epsilon);
            g.at(i, j, k, 0) = -(phi - phi_x) / p.dx;
            g.at(i, j, k, 1) = -(phi - phi_y) / p.dy;
          }
        }
      }
    return true;
  }
  if (p.mhdEnabled && (problem == "MRI" || problem == "Mri" || problem == "mri")) {
    // init_mhd_mri_grav_field (MHDRunBase.cpp:3163-3211): g_z = -(Phi(z+dz) - Phi(z-dz)) / (2 dz), Phi = Omega0^2 z^2 / 2,
    // optionally flattened above zFloor; x and y components zero
    const bool smoothGravity = cfg.get_bool("MRI", "smoothGravity", false);
    const double zFloor = cfg.get_float("MRI", "zFloor", 5.0f);
    for (int k = 0; k < g.ksize; ++k) {
      const double zPos = p.zMin + p.dz / 2 + (k + g.k_shift - g.gw) * p.dz;
      double phi0 = 0.5 * p.Omega0 * p.Omega0 * (zPos - p.dz) * (zPos - p.dz);
      double phi1 = 0.5 * p.Omega0 * p.Omega0 * (zPos + p.dz) * (zPos + p.dz);
      if (smoothGravity) {
        if ((zPos - p.dz) > zFloor) phi0 = 0.5 * p.Omega0 * p.Omega0 * zFloor * zFloor;
        if ((zPos + p.dz) > zFloor) phi1 = 0.5 * p.Omega0 * p.Omega0 * zFloor * zFloor;
      }
      for (int j = 0; j < g.jsize; ++j)
        for (int i = 0; i < g.isize; ++i) {
          g.at(i, j, k, 0) = -0.5 * (0.0 - 0.0) / p.dx;
          g.at(i, j, k, 1) = -0.5 * (0.0 - 0.0) / p.dy;
          g.at(i, j, k, 2) = -0.5 * (phi1 - phi0) / p.dz;
        }
    }
    return true;
  }
  return false;
}

// h_randomForcing of the "turbulence" problem: hF[3][ksize][jsize][isize]
bool init_forcing_field(const IniConfig& cfg, const rgpu_params& p, double* hF) {
  Grid g = make_grid(p, hF);
  if (!p.randomForcingEnabled) return false;
  turbulence_field(p, g, cfg.get_float("turbulence", "machNumber", 0.0f), hF);
  return true;
}

void init_condition(const IniConfig& cfg, const rgpu_params& p, double* hU) {
  const Grid g = make_grid(p, hU);
  std::memset(hU, 0, sizeof(double) * g.ncell * g.nvar);
  const std::string problem = cfg.get_string("hydro", "problem", "unknown");
  if (p.mhdEnabled) {
    // dispatch of MHDRunBase::init_simulation (MHDRunBase.cpp:1286-1342)
    if (problem == "Orszag-Tang" || problem == "OrszagTang") init_orszag_tang(cfg, p, g);
    else if (problem == "Brio-Wu" || problem == "BrioWu" || problem == "brio-wu" || problem == "briowu") init_brio_wu(cfg, p, g);
    else if (problem == "MRI" || problem == "Mri" || problem == "mri") init_mri(cfg, p, g);
    else if (problem == "Kelvin-Helmholtz") init_mhd_kelvin_helmholtz(cfg, p, g);
    else if (problem == "Rayleigh-Taylor") init_rayleigh_taylor(cfg, p, g);
    else if (problem == "jet" || problem == "Jet") init_mhd_jet(cfg, p, g);
    else if (problem == "sod") init_mhd_sod(p, g);
    else if (problem == "Rotor" || problem == "rotor") init_mhd_rotor(cfg, p, g);
    else if (problem == "FieldLoop" || problem == "fieldloop" || problem == "Fieldloop" || problem == "field-loop" || problem == "Field-Loop") init_mhd_field_loop(cfg, p, g);
    else if (problem == "CurrentSheet" || problem == "currentsheet" || problem == "Currentsheet" || problem == "current-sheet" || problem == "Current-Sheet") init_mhd_current_sheet(cfg, p, g);
    else if (problem == "ShearWave" || problem == "shearwave" || problem == "Shear-Wave" || problem == "shear-wave" || problem == "Shearwave") init_mhd_shear_wave(cfg, p, g);
    else if (problem == "turbulence") init_turbulence(cfg, p, g);
    else if (problem == "turbulence-Ornstein-Uhlenbeck") init_turbulence_ou(cfg, p, g);
    else if (problem == "InertialWave" || problem == "inertialwave" || problem == "Inertial-Wave" || problem == "inertial-wave" || problem == "Inertialwave") init_mhd_inertial_wave(cfg, p, g);
    else throw std::runtime_error("MHD problem '" + problem + "' is outside the implemented scope");
  } else {
    if (problem == "jet") init_hydro_jet(p, g);
    else if (problem == "implode") init_hydro_implode(cfg, p, g);
    else if (problem == "sod") init_hydro_sod(p, g);
    else if (problem == "Kelvin-Helmholtz") init_hydro_kelvin_helmholtz(cfg, p, g);
    else if (problem == "Rayleigh-Taylor") init_rayleigh_taylor(cfg, p, g);
    else if (problem == "blast") init_hydro_blast(cfg, p, g);
    else if (problem == "Gresho-vortex") init_hydro_gresho(cfg, p, g);
    else if (problem == "riemann2d") init_hydro_riemann2d(cfg, p, g);
    else if (problem == "falling-bubble") init_hydro_falling_bubble(cfg, p, g);
    else if (problem == "Keplerian-disk") init_hydro_keplerian_disk(cfg, p, g);
    else if (problem == "turbulence") init_turbulence(cfg, p, g);
    else if (problem == "turbulence-Ornstein-Uhlenbeck") init_turbulence_ou(cfg, p, g);
    else throw std::runtime_error("hydro problem '" + problem + "' is outside the implemented scope");
  }
}

}  // namespace rgpu_host
