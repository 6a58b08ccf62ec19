// orc_ou.cpp -- ORACLE (test infrastructure): restatement of the Ornstein-Uhlenbeck forcing of problem
// "turbulence-Ornstein-Uhlenbeck" on the CPU, in the reference's loop and operand order.
//   RandomGen::ranf / ranfModMult / gaussDev / rans(N = 1)     hydro/RandomGen.cpp:61-80, 407-429, 140-168, 238-252
//   ForcingOrnsteinUhlenbeck::init_forcing                      hydro/Forcing_OrnsteinUhlenbeck.cpp:133-231
//   ... ::update_forcing_field_mode (CPU branch)                :531-571
//   ... ::add_forcing_field (CPU version)                       :597-686
// Not used by the product (which has its own host-side process in ramsesgpu_amd/csrc/ou_forcing.h).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "orc_common.h"

namespace orc {

namespace {
const int nMode = 31, nDim = 3;

struct Gen {   // RandomGen restricted to what the forcing uses
  int Seed[4];
  int IGauss;
  double GaussBak;
  void ranf(double& RandNum) {
    static const int Multiplier[4] = {373, 3707, 1442, 647};
    static const double Divisor[4] = {281474976710656.0, 68719476736.0, 16777216.0, 4096.0};
    RandNum = (float)(Seed[3]) / Divisor[3] + (float)(Seed[2]) / Divisor[2] + (float)(Seed[1]) / Divisor[1] + (float)(Seed[0]) / Divisor[0];
    int j0, j1, j2, j3, k0, k1, k2, k3;
    const int* A = Multiplier;
    const int* B = Seed;
    j0 = A[0] * B[0];
    j1 = A[0] * B[1] + A[1] * B[0];
    j2 = A[0] * B[2] + A[1] * B[1] + A[2] * B[0];
    j3 = A[0] * B[3] + A[1] * B[2] + A[2] * B[1] + A[3] * B[0];
    k0 = j0;
    k1 = j1 + k0 / 4096;
    k2 = j2 + k1 / 4096;
    k3 = j3 + k2 / 4096;
    Seed[0] = k0 % 4096; Seed[1] = k1 % 4096; Seed[2] = k2 % 4096; Seed[3] = k3 % 4096;
  }
  void gaussDev(double& GaussNum) {
    double fac, rsq, v1, v2;
    if (IGauss == 0) {
      rsq = 0.0;
      while (rsq >= 1.0 || rsq <= 0.0) {
        ranf(v1);
        ranf(v2);
        v1 = 2.0 * v1 - 1.0;
        v2 = 2.0 * v2 - 1.0;
        rsq = v1 * v1 + v2 * v2;
      }
      fac = sqrt(-2.0 * log(rsq) / rsq);
      GaussBak = v1 * fac;
      GaussNum = v2 * fac;
      IGauss = 1;
    } else {
      GaussNum = GaussBak;
      IGauss = 0;
    }
  }
};

struct State {
  bool on;
  Gen gen;
  double mode[3 * 31], forcingField[3 * 31], projTens[3 * 3 * 31];
  double timeScaleTurb, amplitudeTurb, ksi;
};
State& st() { static State s; return s; }
}  // namespace

void ou_forget() { st().on = false; }

void ou_init(const rgpu_params& p) {
  State& s = st();
  s.on = p.ouForcingEnabled != 0;
  if (!s.on) return;
  s.timeScaleTurb = p.ouTimeScaleTurb; s.amplitudeTurb = p.ouAmplitudeTurb; s.ksi = p.ouKsi;
  for (int i = 0; i < nDim * nMode; i++) { s.mode[i] = 0.0; s.forcingField[i] = 0.0; }
  double ID[3][3];
  for (int j = 0; j < nDim; j++)
    for (int i = 0; i < nDim; i++) ID[j][i] = 0.0;
  for (int i = 0; i < nDim; i++) ID[i][i] = 0.0;   // sic
  // rans(1, init_random, gaussSeed)
  s.gen.IGauss = 0; s.gen.GaussBak = 0.0;
  if (p.ouInitRandom == 0) { s.gen.Seed[0] = 3281; s.gen.Seed[1] = 4041; s.gen.Seed[2] = 595; s.gen.Seed[3] = 2376; }
  else { s.gen.Seed[0] = abs(p.ouInitRandom); s.gen.Seed[1] = 0; s.gen.Seed[2] = 0; s.gen.Seed[3] = 0; }
  double* mode = s.mode;
  mode[ 0]=0.0 ; mode[1*nMode+ 0]=0.0 ; mode[2*nMode+ 0]=2.0;
  mode[ 1]=0.0 ; mode[1*nMode+ 1]=0.0 ; mode[2*nMode+ 1]=3.0;
  mode[ 2]=0.0 ; mode[1*nMode+ 2]=1.0 ; mode[2*nMode+ 2]=2.0;
  mode[ 3]=0.0 ; mode[1*nMode+ 3]=1.0 ; mode[2*nMode+ 3]=3.0;
  mode[ 4]=0.0 ; mode[1*nMode+ 4]=2.0 ; mode[2*nMode+ 4]=0.0;
  mode[ 5]=0.0 ; mode[1*nMode+ 5]=2.0 ; mode[2*nMode+ 5]=1.0;
  mode[ 6]=0.0 ; mode[1*nMode+ 6]=2.0 ; mode[2*nMode+ 6]=2.0;
  mode[ 7]=0.0 ; mode[1*nMode+ 7]=3.0 ; mode[2*nMode+ 7]=0.0;
  mode[ 8]=0.0 ; mode[1*nMode+ 8]=3.0 ; mode[2*nMode+ 8]=1.0;
  mode[ 9]=1.0 ; mode[1*nMode+ 9]=0.0 ; mode[2*nMode+ 9]=2.0;
  mode[10]=1.0 ; mode[1*nMode+10]=0.0 ; mode[2*nMode+10]=3.0;
  mode[11]=1.0 ; mode[1*nMode+11]=1.0 ; mode[2*nMode+11]=2.0;
  mode[12]=1.0 ; mode[1*nMode+12]=1.0 ; mode[2*nMode+12]=3.0;
  mode[13]=1.0 ; mode[1*nMode+13]=2.0 ; mode[2*nMode+13]=0.0;
  mode[14]=1.0 ; mode[1*nMode+14]=2.0 ; mode[2*nMode+14]=1.0;
  mode[15]=1.0 ; mode[1*nMode+15]=2.0 ; mode[2*nMode+15]=2.0;
  mode[16]=1.0 ; mode[1*nMode+16]=3.0 ; mode[2*nMode+16]=0.0;
  mode[17]=1.0 ; mode[1*nMode+17]=3.0 ; mode[2*nMode+17]=1.0;
  mode[18]=2.0 ; mode[1*nMode+18]=0.0 ; mode[2*nMode+18]=0.0;
  mode[19]=2.0 ; mode[1*nMode+19]=0.0 ; mode[2*nMode+19]=1.0;
  mode[20]=2.0 ; mode[1*nMode+20]=0.0 ; mode[2*nMode+20]=2.0;
  mode[21]=2.0 ; mode[1*nMode+21]=1.0 ; mode[2*nMode+21]=0.0;
  mode[22]=2.0 ; mode[1*nMode+22]=1.0 ; mode[2*nMode+22]=1.0;
  mode[23]=2.0 ; mode[1*nMode+23]=1.0 ; mode[2*nMode+23]=2.0;
  mode[24]=2.0 ; mode[1*nMode+24]=2.0 ; mode[2*nMode+24]=0.0;
  mode[25]=2.0 ; mode[1*nMode+25]=2.0 ; mode[2*nMode+25]=1.0;
  mode[26]=2.0 ; mode[1*nMode+26]=2.0 ; mode[2*nMode+26]=2.0;
  mode[27]=3.0 ; mode[1*nMode+27]=0.0 ; mode[2*nMode+27]=0.0;
  mode[28]=3.0 ; mode[1*nMode+28]=0.0 ; mode[2*nMode+28]=1.0;
  mode[29]=3.0 ; mode[1*nMode+29]=1.0 ; mode[2*nMode+29]=0.0;
  mode[30]=3.0 ; mode[1*nMode+30]=1.0 ; mode[2*nMode+30]=1.0;
  for (int iMode = 0; iMode < nMode; iMode++) {
    double sum = 0.0;
    double randomNumber;
    for (int iDim = 0; iDim < nDim; iDim++) {
      s.gen.gaussDev(randomNumber);
      mode[iDim * nMode + iMode] = copysign(mode[iDim * nMode + iMode], randomNumber);
      sum = sum + mode[iDim * nMode + iMode] * mode[iDim * nMode + iMode];
    }
    for (int j = 0; j < nDim; j++)
      for (int i = 0; i < nDim; i++)
        s.projTens[i * nDim * nMode + j * nMode + iMode] =
            s.ksi * ID[i][j] + (1.0 - 2.0 * s.ksi) * mode[j * nMode + iMode] * mode[i * nMode + iMode] / sum;
  }
}

// add_forcing_field(h_UNew, dt), CPU version
void ou_forcing(const Ctx& c, double* U, double dt) {
  State& s = st();
  if (!s.on || !c.three_d) return;
  const rgpu_params& p = c.p;
  // 1. update Fourier modes of the forcing field
  {
    double weight = s.amplitudeTurb;
    double v = sqrt(5.0 / 3.0) * p.cIso;
    for (int iMode = 0; iMode < nMode; iMode++) {
      double AAA[3] = {0.0, 0.0, 0.0};
      double BBB[3] = {0.0, 0.0, 0.0};
      double randomNumber;
      for (int i = 0; i < nDim; i++) {
        s.gen.gaussDev(randomNumber);
        AAA[i] = randomNumber * sqrt(dt);
      }
      for (int j = 0; j < nDim; j++) {
        double summ = 0.0;
        for (int i = 0; i < nDim; i++) summ += s.projTens[i * nDim * nMode + j * nMode + iMode] * AAA[i];
        BBB[j] = summ;
      }
      for (int i = 0; i < nDim; i++) BBB[i] = BBB[i] * v * sqrt(2.0 * weight * weight / s.timeScaleTurb) / s.timeScaleTurb;
      for (int i = 0; i < nDim; i++) BBB[i] = BBB[i] - s.forcingField[i * nMode + iMode] * dt / s.timeScaleTurb;
      double forceRMS = 3.0 / sqrt(1 - 2.0 * s.ksi + 3.0 * s.ksi * s.ksi);
      for (int i = 0; i < nDim; i++) s.forcingField[i * nMode + iMode] += forceRMS * BBB[i];
    }
  }
  // 2. + 3. phases at the cell centres, update of momenta and total energy
  const int ghostWidth = c.gw;
  const size_t N = c.ncell;
  const double twoPi = 2 * M_PI;
  for (int k = ghostWidth; k < c.ksize - ghostWidth; k++) {
    double zPos = p.zMin + p.dz / 2 + (k - ghostWidth + p.nz * 0) * p.dz;   // single domain: mpiPosZ = 0
    for (int j = ghostWidth; j < c.jsize - ghostWidth; j++) {
      double yPos = p.yMin + p.dy / 2 + (j - ghostWidth) * p.dy;
      for (int i = ghostWidth; i < c.isize - ghostWidth; i++) {
        double xPos = p.xMin + p.dx / 2 + (i - ghostWidth) * p.dx;
        double phase[31];
        memset(phase, 0, sizeof(phase));
        for (int iMode = 0; iMode < nMode; iMode++)
          phase[iMode] = xPos * s.mode[0 * nMode + iMode] + yPos * s.mode[1 * nMode + iMode] + zPos * s.mode[2 * nMode + iMode];
        double AAA[3];
        for (int iDim = 0; iDim < nDim; iDim++) {
          double summ = 0.0;
          for (int iMode = 0; iMode < nMode; iMode++) summ += s.forcingField[iDim * nMode + iMode] * cos(twoPi * phase[iMode]);
          AAA[iDim] = summ;
        }
        const size_t o = c.idx(i, j, k);
        double eInt = 0.5 * (U[o + IU * N] * U[o + IU * N] + U[o + IV * N] * U[o + IV * N] + U[o + IW * N] * U[o + IW * N]) / U[o + ID * N];
        eInt = U[o + IP * N] - eInt;
        double rho = U[o + ID * N];
        U[o + IU * N] += AAA[0] * dt * rho;
        U[o + IV * N] += AAA[1] * dt * rho;
        U[o + IW * N] += AAA[2] * dt * rho;
        U[o + IP * N] = eInt + 0.5 * (U[o + IU * N] * U[o + IU * N] + U[o + IV * N] * U[o + IV * N] + U[o + IW * N] * U[o + IW * N]) / U[o + ID * N];
      }
    }
  }
}

}  // namespace orc
