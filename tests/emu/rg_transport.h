// rg_transport.h (TEST-ONLY) -- NOT part of the product.  Transport of the z-slab driver for the CPU tests: the driver's
// point-to-point and all-reduce requests are handed to callbacks the test registers (tests/comm_worker.py implements
// them with torch.distributed / gloo), so that the C++ schedule of csrc/comm/rgpu_comm.cpp can be run with world sizes
// 2 and 3 on a machine without GPUs or RCCL devices.  Memory is host memory here (emulation backend).
#pragma once
#include <cstring>
#include <string>
#include <vector>

#define RG_TRANSPORT_NAME "test-callbacks(gloo)"

namespace rgpu_transport {

struct P2P { double* ptr; size_t count; int peer; int send; };

// ops: array of nops P2P records; must complete all of them before returning (0 = ok)
typedef int (*exchange_fn)(const P2P* ops, int nops);
// op: 0 = max, 1 = sum; in place on n doubles
typedef int (*allreduce_fn)(double* data, int n, int op);

struct Callbacks { exchange_fn exchange; allreduce_fn allreduce; };
inline Callbacks& callbacks() { static Callbacks cb = {0, 0}; return cb; }

struct Comm {
  int rank, nranks;
  std::vector<P2P> pending;
  std::string err;
};
inline int fail(Comm* c, const std::string& m) { if (c) c->err = m; return -1; }

inline int unique_id(char* id128) { std::memset(id128, 0, 128); std::memcpy(id128, "test-transport", 14); return 0; }
inline int create(Comm** out, int rank, int nranks, const char*) {
  Comm* c = new Comm();
  c->rank = rank; c->nranks = nranks;
  *out = c;
  if (nranks > 1 && (!callbacks().exchange || !callbacks().allreduce)) return fail(c, "test transport: callbacks not registered");
  return 0;
}
inline void destroy(Comm* c) { delete c; }
// (the packed exchange is a device-side optimisation of the RCCL transport: nothing to stage here)
inline int prepare_exchange(Comm*, const P2P*, int) { return 0; }
inline void disable_pack(Comm*) {}
inline bool packs(const Comm*) { return false; }
// the emulation backend executes kernels at launch, so "behind what the compute stream holds" is now; the transfer itself
// is deferred to exchange_wait so that the overlapped schedule's compute really runs between start and wait
inline int exchange_start(Comm* c, void*, const P2P* ops, int nops) {
  c->pending.assign(ops, ops + nops);
  return 0;
}
inline int exchange_wait(Comm* c, void*) {
  if (c->pending.empty()) return 0;
  int rc = 0;
  if (c->nranks == 1) {   // self ring: every send has its matching receive in the same list, in posting order
    std::vector<P2P> sends, recvs;
    for (size_t i = 0; i < c->pending.size(); ++i) (c->pending[i].send ? sends : recvs).push_back(c->pending[i]);
    // post order of the driver: [send to prev, send to next] per variable, then [recv from next, recv from prev]:
    // what is sent "to prev" arrives "from next"
    for (size_t i = 0; i < sends.size() && i < recvs.size(); ++i) std::memcpy(recvs[i].ptr, sends[i].ptr, sends[i].count * sizeof(double));
  } else {
    rc = callbacks().exchange(c->pending.data(), (int)c->pending.size());
  }
  c->pending.clear();
  return rc ? fail(c, "exchange callback failed") : 0;
}
inline double last_exchange_ms(Comm*) { return -1.0; }
inline int allreduce_max(Comm* c, double* d, int n, void*) {
  if (c->nranks == 1) return 0;
  return callbacks().allreduce(d, n, 0) ? fail(c, "allreduce callback failed") : 0;
}
inline int allreduce_sum_host(Comm* c, double* h, int n, void*) {
  if (c->nranks == 1) return 0;
  return callbacks().allreduce(h, n, 1) ? fail(c, "allreduce callback failed") : 0;
}
inline int barrier(Comm* c, void* s) { double z = 0; return allreduce_sum_host(c, &z, 1, s); }

}  // namespace rgpu_transport

namespace rgpu_transport {
inline int version(const Comm*) { return 0; }
inline void set_device(int) {}
inline int info(Comm* c, int* nranks, int* rank, int* device, char* pci, int pci_len) {
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  if (device) *device = -1;
  if (pci && pci_len > 0) pci[0] = 0;
  return 0;
}
inline void abort_comm(Comm*) {}
inline int poison_slot(Comm*, double* d, void*) { const unsigned long long b = 0x7ff0000000000000ull; std::memcpy(d, &b, sizeof(b)); return 0; }
}

// registered once per process by the test worker before rgpu_comm_create (single translation unit: defined here)
extern "C" void rgpu_comm_test_set_callbacks(rgpu_transport::exchange_fn e, rgpu_transport::allreduce_fn a) {
  rgpu_transport::callbacks().exchange = e;
  rgpu_transport::callbacks().allreduce = a;
}
