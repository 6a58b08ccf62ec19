// halo_ops.h -- the send / receive descriptors of one z-halo exchange of a slab, as offsets into its state array.
// Plain C++ (shared by csrc/comm/rgpu_comm.cpp and the host-side check tests/cpp/pack_plan_check.cpp).
//
// State array of a slab: U[v][k][j][i], k slowest inside a variable, nz + 2 gw planes of `plane` cells; the gw planes of one
// variable that go to / come from a neighbour are one contiguous chunk.  Posting order = the order the peer's matching operations
// are posted in (grouped point-to-point operations between two ranks match in order): per variable [low interior planes -> prev]
// [high interior planes -> next], then per variable [high ghost planes <- next] [low ghost planes <- prev] -- what rank r sends "to
// prev" is what rank r - 1 receives "from next".
#pragma once
#include <cstddef>
#include <vector>

namespace rgpu_transport {

struct HaloOp { size_t offset, count; int peer, send; };

inline void halo_ops(size_t plane, int gw, int nz, int nvar, int rank, int nranks, bool has_prev, bool has_next, std::vector<HaloOp>& ops) {
  ops.clear();
  const size_t ncell = plane * (size_t)(nz + 2 * gw);
  const size_t chunk = plane * (size_t)gw;
  const int prev = (rank - 1 + nranks) % nranks, next = (rank + 1) % nranks;
  for (int v = 0; v < nvar; ++v) {
    const size_t b = (size_t)v * ncell;
    if (has_prev) { const HaloOp o = {b + plane * (size_t)gw, chunk, prev, 1}; ops.push_back(o); }          // low interior planes
    if (has_next) { const HaloOp o = {b + plane * (size_t)nz, chunk, next, 1}; ops.push_back(o); }          // high interior planes
  }
  for (int v = 0; v < nvar; ++v) {
    const size_t b = (size_t)v * ncell;
    if (has_next) { const HaloOp o = {b + plane * (size_t)(nz + gw), chunk, next, 0}; ops.push_back(o); }   // high ghost planes
    if (has_prev) { const HaloOp o = {b, chunk, prev, 0}; ops.push_back(o); }                               // low ghost planes
  }
}

}  // namespace rgpu_transport
