// init_conditions.h -- initial conditions of the problems named by the scope contract:
//   jet, implode                    HydroRunBase.cpp:5282-5350, 5449-5536
//   Orszag-Tang, Brio-Wu, MRI       MHDRunBase.cpp:1378-1720, 1870-2080, 2677-2758
// and, as the first widening beyond it (SURVEY.md 8f-1 "makes every shipped .ini runnable"), the problems of the other
// shipped parameter files that need no source term: sod, blast, Kelvin-Helmholtz (hydro, HydroRunBase.cpp:5358-6252),
// MHD jet, sod, rotor, field loop, current sheet, Kelvin-Helmholtz (MHDRunBase.cpp:1747-1862, 2117-2487, 2814-2984)
// Written for z-slabs: local plane k of slab r is global plane k + r*nz_local, and the MRI random stream is
// skipped ahead so that every slab draws exactly the numbers the single-domain reference run would.
#pragma once
#include <string>

#include "../../../include/rgpu.h"
#include "ini_config.h"

namespace rgpu_host {

// hU: rgpu_state_elems(p) doubles; zero-filled first (the reference memsets h_U in every init routine).
// Throws std::runtime_error for unknown problems.
void init_condition(const IniConfig& cfg, const rgpu_params& p, double* hU);

// hG: 3 * ncell doubles (x, y, z component planes, ghost cells included), zero-filled first; returns false when the
// problem defines no per-cell field.  The reference fills h_gravity inside its init routines (Keplerian-disk:
// HydroRunBase.cpp:6489-6500, 6575-6597).
bool init_gravity_field(const IniConfig& cfg, const rgpu_params& p, double* hG);

// hF: 3 * ncell doubles, the static driving field of the "turbulence" problem (turbulenceInit.cpp); false for every other
// problem.
bool init_forcing_field(const IniConfig& cfg, const rgpu_params& p, double* hF);

// glibc-compatible drand48 stream (48-bit LCG X' = a X + c mod 2^48, a=0x5DEECE66D, c=0xB; srand48(s) sets
// X = (s<<16)|0x330E) with O(log n) skip-ahead.
class Rand48 {
 public:
  explicit Rand48(long seed) : x_(((static_cast<unsigned long long>(seed) & 0xFFFFFFFFULL) << 16) | 0x330EULL) {}
  double next() {
    x_ = (kA * x_ + kC) & kMask;
    return static_cast<double>(x_) * (1.0 / 281474976710656.0);  // exact: X / 2^48
  }
  void skip(unsigned long long n);

 private:
  static const unsigned long long kA = 0x5DEECE66DULL, kC = 0xBULL, kMask = (1ULL << 48) - 1;
  unsigned long long x_;
};

// glibc-compatible rand() after srand(seed): the TYPE_3 additive feedback generator (r[i] = r[i-3] + r[i-31] on 32-bit
// words, 310 values discarded after seeding with the minimal-standard LCG, output = word >> 1; RAND_MAX = 2^31 - 1)
class GlibcRand {
 public:
  explicit GlibcRand(unsigned seed);
  int next();
  static constexpr double kRandMax = 2147483647.0;

 private:
  unsigned r_[34];
  int f_, b_;
};

}  // namespace rgpu_host
