// init_conditions.h -- initial conditions of the problems named by the scope contract:
//   jet, implode                    HydroRunBase.cpp:5282-5350, 5449-5536
//   Orszag-Tang, Brio-Wu, MRI       MHDRunBase.cpp:1378-1720, 1870-2080, 2677-2758
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

}  // namespace rgpu_host
