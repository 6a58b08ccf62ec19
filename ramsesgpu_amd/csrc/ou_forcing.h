// ou_forcing.h -- the Ornstein-Uhlenbeck forcing process of problem "turbulence-Ornstein-Uhlenbeck", host side
// (plain C++, used by the step driver under every backend).
//
// Stands in for ForcingOrnsteinUhlenbeck (hydro/Forcing_OrnsteinUhlenbeck.cpp) with nDim = 3, nCpu = 1:
//   init()     init_forcing, :133-231      31 wave vectors, their signs drawn from the Gaussian generator, projection tensor
//   update(dt) update_forcing_field_mode, CPU branch :531-571    f += forceRMS * (P . xi sqrt(dt) v sqrt(2 w^2/T)/T - f dt/T)
// and for the generator it uses, RandomGen (hydro/RandomGen.cpp): a 48-bit multiplicative congruential generator kept
// as four base-4096 digits (ranf :61-80, ranfModMult :407-429), polar Box-Muller with a spare (gaussDev :140-168),
// seeded by rans(1, init_random, .) = {|init_random|, 0, 0, 0} (:238-252).
// Quirk kept: the "identity" matrix of init_forcing is all zeros (:160-164), so the ksi * ID term never contributes.
#pragma once
#include <cmath>
#include <cstdlib>

namespace rgpu_ou {

enum { NMODE = 31, NDIM = 3 };

// device-side view: what the kernel needs each step
struct OuModes {
  double mode[NDIM * NMODE];    // wave vector components, [dim][mode]
  double force[NDIM * NMODE];   // forcing amplitudes, [dim][mode]
};

class OuProcess {
 public:
  OuModes m;
  double proj[NDIM * NDIM * NMODE];

  void init(int init_random, double timeScaleTurb, double amplitudeTurb, double ksi) {
    T_ = timeScaleTurb; w_ = amplitudeTurb; ksi_ = ksi;
    igauss_ = 0; spare_ = 0.0;
    seed_[0] = init_random == 0 ? 3281 : std::abs(init_random);
    seed_[1] = init_random == 0 ? 4041 : 0;
    seed_[2] = init_random == 0 ? 595 : 0;
    seed_[3] = init_random == 0 ? 2376 : 0;
    // |k| components of the 31 modes, in the reference's order (Forcing_OrnsteinUhlenbeck.cpp:175-205)
    static const unsigned char K[NMODE][3] = {
        {0, 0, 2}, {0, 0, 3}, {0, 1, 2}, {0, 1, 3}, {0, 2, 0}, {0, 2, 1}, {0, 2, 2}, {0, 3, 0}, {0, 3, 1}, {1, 0, 2}, {1, 0, 3},
        {1, 1, 2}, {1, 1, 3}, {1, 2, 0}, {1, 2, 1}, {1, 2, 2}, {1, 3, 0}, {1, 3, 1}, {2, 0, 0}, {2, 0, 1}, {2, 0, 2}, {2, 1, 0},
        {2, 1, 1}, {2, 1, 2}, {2, 2, 0}, {2, 2, 1}, {2, 2, 2}, {3, 0, 0}, {3, 0, 1}, {3, 1, 0}, {3, 1, 1}};
    for (int i = 0; i < NDIM * NMODE; ++i) m.force[i] = 0.0;
    for (int im = 0; im < NMODE; ++im) {
      double sum = 0.0;
      for (int d = 0; d < NDIM; ++d) {
        double r;
        gauss(r);
        const double v = std::copysign((double)K[im][d], r);
        m.mode[d * NMODE + im] = v;
        sum = sum + v * v;
      }
      for (int j = 0; j < NDIM; ++j)
        for (int i = 0; i < NDIM; ++i)
          proj[i * NDIM * NMODE + j * NMODE + im] = ksi_ * 0.0 + (1.0 - 2.0 * ksi_) * m.mode[j * NMODE + im] * m.mode[i * NMODE + im] / sum;
    }
  }

  void update(double dt, double cIso) {
    const double weight = w_;
    const double v = std::sqrt(5.0 / 3.0) * cIso;
    for (int im = 0; im < NMODE; ++im) {
      double A[3] = {0.0, 0.0, 0.0}, B[3] = {0.0, 0.0, 0.0};
      for (int i = 0; i < NDIM; ++i) {
        double r;
        gauss(r);
        A[i] = r * std::sqrt(dt);
      }
      for (int j = 0; j < NDIM; ++j) {
        double s = 0.0;
        for (int i = 0; i < NDIM; ++i) s += proj[i * NDIM * NMODE + j * NMODE + im] * A[i];
        B[j] = s;
      }
      for (int i = 0; i < NDIM; ++i) B[i] = B[i] * v * std::sqrt(2.0 * weight * weight / T_) / T_;
      for (int i = 0; i < NDIM; ++i) B[i] = B[i] - m.force[i * NMODE + im] * dt / T_;
      const double forceRMS = 3.0 / std::sqrt(1 - 2.0 * ksi_ + 3.0 * ksi_ * ksi_);
      for (int i = 0; i < NDIM; ++i) m.force[i * NMODE + im] += forceRMS * B[i];
    }
  }

  // whole state of the process as doubles (restart files; what output_forcing / init_forcing(restart) exchange through
  // an .npz in the reference, Forcing_OrnsteinUhlenbeck.cpp:392-444, 236-352 -- plus the Box-Muller spare, which the
  // reference loses across a restart): mode, force, proj, the four seed digits, the spare flag and value
  enum { STATE_DOUBLES = 2 * NDIM * NMODE + NDIM * NDIM * NMODE + 6 };
  void get_state(double* out) const {
    int n = 0;
    for (int i = 0; i < NDIM * NMODE; ++i) out[n++] = m.mode[i];
    for (int i = 0; i < NDIM * NMODE; ++i) out[n++] = m.force[i];
    for (int i = 0; i < NDIM * NDIM * NMODE; ++i) out[n++] = proj[i];
    for (int i = 0; i < 4; ++i) out[n++] = seed_[i];
    out[n++] = igauss_;
    out[n++] = spare_;
  }
  void set_state(const double* in) {
    int n = 0;
    for (int i = 0; i < NDIM * NMODE; ++i) m.mode[i] = in[n++];
    for (int i = 0; i < NDIM * NMODE; ++i) m.force[i] = in[n++];
    for (int i = 0; i < NDIM * NDIM * NMODE; ++i) proj[i] = in[n++];
    for (int i = 0; i < 4; ++i) seed_[i] = (int)in[n++];
    igauss_ = (int)in[n++];
    spare_ = in[n++];
  }

 private:
  int seed_[4];
  int igauss_;
  double spare_, T_, w_, ksi_;

  void uniform(double& r) {
    r = (float)seed_[3] / 4096.0 + (float)seed_[2] / 16777216.0 + (float)seed_[1] / 68719476736.0 + (float)seed_[0] / 281474976710656.0;
    static const int A[4] = {373, 3707, 1442, 647};   // the multiplier, base-4096 digits, least significant first
    const int j0 = A[0] * seed_[0];
    const int j1 = A[0] * seed_[1] + A[1] * seed_[0];
    const int j2 = A[0] * seed_[2] + A[1] * seed_[1] + A[2] * seed_[0];
    const int j3 = A[0] * seed_[3] + A[1] * seed_[2] + A[2] * seed_[1] + A[3] * seed_[0];
    const int k0 = j0, k1 = j1 + k0 / 4096, k2 = j2 + k1 / 4096, k3 = j3 + k2 / 4096;
    seed_[0] = k0 % 4096; seed_[1] = k1 % 4096; seed_[2] = k2 % 4096; seed_[3] = k3 % 4096;
  }
  void gauss(double& g) {
    if (igauss_ == 0) {
      double rsq = 0.0, v1 = 0.0, v2 = 0.0;
      while (rsq >= 1.0 || rsq <= 0.0) {
        uniform(v1);
        uniform(v2);
        v1 = 2.0 * v1 - 1.0;
        v2 = 2.0 * v2 - 1.0;
        rsq = v1 * v1 + v2 * v2;
      }
      const double fac = std::sqrt(-2.0 * std::log(rsq) / rsq);
      spare_ = v1 * fac;
      g = v2 * fac;
      igauss_ = 1;
    } else {
      g = spare_;
      igauss_ = 0;
    }
  }
};

}  // namespace rgpu_ou
