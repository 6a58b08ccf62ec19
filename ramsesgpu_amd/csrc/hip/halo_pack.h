// halo_pack.h (HIP / gfx950) -- device side of the packed halo exchange: the copy kernels that gather the chunks going to one
// peer into a staging buffer and scatter the received ones, and the launch record built from the host-side plan
// (comm/pack_plan.h).  Used by the RCCL transport (rg_transport.h); nothing here knows about RCCL, so a test transport that moves
// the staged bytes another way (tests/emu_dev/rg_transport.h: pinned host buffers + gloo on ONE GPU shared by several rank
// processes) runs exactly these kernels and this plan.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "../comm/pack_plan.h"

namespace rgpu_transport {

struct P2P { double* ptr; size_t count; int peer; int send; };

// by-value kernel argument: the chunks of one direction (all sends, or all receives) of one exchange
struct PackDesc {
  double* ptr[kPackMaxChunksPerDirection];
  unsigned long long off[kPackMaxChunksPerDirection];
  unsigned long long count[kPackMaxChunksPerDirection];
  int n;
};
static_assert(kPackMaxOps >= 2 * kPackMaxChunksPerDirection, "pack_plan.h: an exchange holds at most kPackMaxChunksPerDirection sends and as many receives");
static_assert(sizeof(PackDesc) <= 1024, "PackDesc travels as a kernel argument");

__global__ void pack_chunks_kernel(PackDesc d, double* __restrict__ stage, int unpack) {
  const int seg = (int)blockIdx.y;
  if (seg >= d.n) return;
  double* __restrict__ p = d.ptr[seg];
  double* __restrict__ st = stage + d.off[seg];
  const unsigned long long n = d.count[seg];
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
    if (unpack) p[i] = st[i]; else st[i] = p[i];
  }
}

// plan + the two kernel descriptors of one operation list
struct PackedExchange {
  PackPlan pl;
  PackDesc ds, dr;
  unsigned bx;   // blocks along x of the copy kernels
};

// 0, or -1: the list does not fit one plan (more than kPackMaxOps operations, kPackMaxPeers peers or kPackMaxChunksPerDirection
// chunks in one direction) -- the caller then exchanges in place
inline int build_packed(const P2P* ops, int nops, PackedExchange* px) {
  if (nops > kPackMaxOps || pack_plan(ops, nops, &px->pl)) return -1;
  px->ds.n = 0; px->dr.n = 0;
  for (int i = 0; i < nops; ++i) {
    PackDesc& d = ops[i].send ? px->ds : px->dr;
    d.ptr[d.n] = ops[i].ptr; d.count[d.n] = ops[i].count; d.off[d.n] = px->pl.off[i];
    ++d.n;
  }
  unsigned bx = (unsigned)((px->pl.longest + 255) / 256);
  px->bx = bx > 256u ? 256u : (bx < 1u ? 1u : bx);
  return 0;
}

inline int launch_pack(const PackedExchange& px, double* stage_send, hipStream_t s) {
  if (px.ds.n) hipLaunchKernelGGL(pack_chunks_kernel, dim3(px.bx, (unsigned)px.ds.n), dim3(256), 0, s, px.ds, stage_send, 0);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
inline int launch_unpack(const PackedExchange& px, double* stage_recv, hipStream_t s) {
  if (px.dr.n) hipLaunchKernelGGL(pack_chunks_kernel, dim3(px.bx, (unsigned)px.dr.n), dim3(256), 0, s, px.dr, stage_recv, 1);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace rgpu_transport
