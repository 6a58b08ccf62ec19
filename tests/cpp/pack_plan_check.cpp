// Host-side check of the packed halo exchange for topologies a one-GPU box cannot run: the operation list of every slab
// (csrc/comm/halo_ops.h, what rgpu_comm.cpp posts) + the staging plan (csrc/comm/pack_plan.h, what hip/rg_transport.h executes),
// carried out with memcpy between the ranks' buffers: pack -> ONE message per ordered pair of ranks -> unpack.  Afterwards every ghost
// plane must hold the neighbour's interior plane.  usage: pack_plan_check <nranks> <periodic 0|1> <nvar> <nz per rank>; exit 0 = ok
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "halo_ops.h"
#include "pack_plan.h"

using namespace rgpu_transport;

static double code(int rank, int v, int k_local, size_t cell) { return 1e6 * rank + 1e4 * v + 10.0 * k_local + 1e-3 * (double)cell; }

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 3, periodic = argc > 2 ? std::atoi(argv[2]) : 1, nvar = argc > 3 ? std::atoi(argv[3]) : 8, nz = argc > 4 ? std::atoi(argv[4]) : 7;
  const int gw = 3;
  const size_t plane = 5 * 4;
  const size_t ncell = plane * (size_t)(nz + 2 * gw);
  std::vector<std::vector<double> > U(N, std::vector<double>((size_t)nvar * ncell, NAN));
  std::vector<std::vector<HaloOp> > ops(N);
  std::vector<PackPlan> pl(N);
  std::vector<std::vector<double> > ss(N), sr(N);
  for (int r = 0; r < N; ++r) {
    for (int v = 0; v < nvar; ++v)
      for (int k = gw; k < nz + gw; ++k)
        for (size_t c = 0; c < plane; ++c) U[r][(size_t)v * ncell + plane * k + c] = code(r, v, k, c);
    const bool has_prev = periodic || r > 0, has_next = periodic || r < N - 1;
    halo_ops(plane, gw, nz, nvar, r, N, has_prev, has_next, ops[r]);
    if (pack_plan(ops[r].data(), (int)ops[r].size(), &pl[r])) { std::printf("rank %d: no plan\n", r); return 2; }
    ss[r].assign(pl[r].stage_doubles, NAN); sr[r].assign(pl[r].stage_doubles, NAN);
    for (size_t i = 0; i < ops[r].size(); ++i)   // pack
      if (ops[r][i].send) std::memcpy(&ss[r][pl[r].off[i]], &U[r][ops[r][i].offset], ops[r][i].count * sizeof(double));
  }
  // ONE message per ordered pair (a -> b): a's send region for b lands in b's receive region for a
  int messages = 0;
  for (int a = 0; a < N; ++a)
    for (int q = 0; q < pl[a].npeers; ++q) {
      if (!pl[a].send_total[q]) continue;
      const int b = pl[a].peer[q];
      int qb = 0;
      while (qb < pl[b].npeers && pl[b].peer[qb] != a) ++qb;
      if (qb == pl[b].npeers || pl[b].recv_total[qb] != pl[a].send_total[q]) { std::printf("rank %d -> %d: %zu doubles sent, no matching receive\n", a, b, pl[a].send_total[q]); return 3; }
      std::memcpy(&sr[b][pl[b].recv_base[qb]], &ss[a][pl[a].send_base[q]], pl[a].send_total[q] * sizeof(double));
      ++messages;
    }
  for (int r = 0; r < N; ++r)   // unpack
    for (size_t i = 0; i < ops[r].size(); ++i)
      if (!ops[r][i].send) std::memcpy(&U[r][ops[r][i].offset], &sr[r][pl[r].off[i]], ops[r][i].count * sizeof(double));
  long bad = 0, checked = 0;
  for (int r = 0; r < N; ++r) {
    const int prev = (r - 1 + N) % N, next = (r + 1) % N;
    const bool has_prev = periodic || r > 0, has_next = periodic || r < N - 1;
    for (int v = 0; v < nvar; ++v)
      for (int g = 0; g < gw; ++g)
        for (size_t c = 0; c < plane; ++c) {
          const double lo = U[r][(size_t)v * ncell + plane * g + c], hi = U[r][(size_t)v * ncell + plane * (nz + gw + g) + c];
          if (has_prev) { ++checked; if (lo != code(prev, v, nz + g, c)) ++bad; } else if (lo == lo) ++bad;        // prev's top interior planes
          if (has_next) { ++checked; if (hi != code(next, v, gw + g, c)) ++bad; } else if (hi == hi) ++bad;        // next's bottom interior planes
        }
  }
  const int expect = periodic ? (N == 1 ? 1 : (N == 2 ? 2 : 2 * N)) : 2 * (N - 1);
  std::printf("nranks %d periodic %d: %d messages (expected %d), %ld ghost values checked, %ld wrong\n", N, periodic, messages, expect, checked, bad);
  return (bad == 0 && messages == expect && checked > 0) || (N == 1 && !periodic && messages == 0 && bad == 0) ? 0 : 1;
}
