// rg_transport.h (TEST-ONLY, device-aware) -- NOT part of the product.  Transport of the z-slab driver for running N > 1 rank
// processes of the PRODUCT libraries (librgpu.so / librgpu_fast.so: the real tiled HIP kernels) on ONE GPU, where RCCL refuses a
// second rank on the same device.  csrc/comm/rgpu_comm.cpp is compiled against this header instead of csrc/hip/rg_transport.h
// (tests/test_comm_device.py builds tests/_build/librgpu_comm_dev[_fast].so); everything the product does around the wire is kept:
//   * the halo stream, ordered against the compute stream by events only (exchange_start does not block the host);
//   * the packed exchange: the product's own plan and copy kernels (csrc/hip/halo_pack.h, csrc/comm/pack_plan.h) gather the chunks
//     for one peer into the device stage and scatter the received stage -- ONE message per peer and direction;
//   * the in-place 1/dt all-reduce of the context's RGPU_DT_SLOTS device slots.
// Only the wire differs: where the RCCL transport calls ncclSend / ncclRecv / ncclAllReduce, this one copies the staged bytes to
// pinned host memory and hands them to callbacks the test registers (tests/comm_worker.py: torch.distributed / gloo).  The
// transfer is deferred to exchange_wait, so the compute the schedule overlaps with the exchange is already queued on the compute
// stream and runs while the planes travel -- the received planes land in the ghost planes while those kernels are in flight, as
// they do with RCCL.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "halo_pack.h"   // the PRODUCT's plan + copy kernels (csrc/hip)

#define RG_TRANSPORT_NAME "test-device-staged(gloo)"

namespace rgpu_transport {

// ops: HOST pointers; must complete all of them before returning (0 = ok)
typedef int (*exchange_fn)(const P2P* ops, int nops);
// op: 0 = max, 1 = sum; in place on n host doubles
typedef int (*allreduce_fn)(double* data, int n, int op);
struct Callbacks { exchange_fn exchange; allreduce_fn allreduce; };
inline Callbacks& callbacks() { static Callbacks cb = {0, 0}; return cb; }

struct Comm {
  int rank, nranks;
  hipStream_t halo;
  hipEvent_t ev_ready, ev_done;
  bool pack;
  double* stage_s; double* stage_r; size_t stage_cap;   // device stages of the packed exchange (doubles)
  double* h_s; double* h_r; size_t h_cap;               // pinned host images of what travels
  double* h_red;                                        // pinned host image of the 1/dt slots / host-value reductions
  // the exchange in flight
  bool pending, pending_packed;
  PackedExchange px;
  std::vector<P2P> dev_ops, host_ops;
  long exchanges, messages;                             // statistics the tests read (rgpu_comm_test_stats)
  std::string err;
};
inline int fail(Comm* c, const std::string& m) { if (c) c->err = m; return -1; }
inline Comm*& last_comm() { static Comm* c = 0; return c; }

inline int unique_id(char* id128) { std::memset(id128, 0, 128); std::memcpy(id128, "test-device-transport", 21); return 0; }

inline int create(Comm** out, int rank, int nranks, const char*) {
  Comm* c = new Comm();
  c->rank = rank; c->nranks = nranks; c->halo = 0; c->ev_ready = 0; c->ev_done = 0;
  c->pack = !(std::getenv("RGPU_COMM_PACK") && std::atoi(std::getenv("RGPU_COMM_PACK")) == 0);
  c->stage_s = 0; c->stage_r = 0; c->stage_cap = 0; c->h_s = 0; c->h_r = 0; c->h_cap = 0; c->h_red = 0;
  c->pending = false; c->pending_packed = false; c->exchanges = 0; c->messages = 0;
  *out = c;
  last_comm() = c;
  if (nranks > 1 && (!callbacks().exchange || !callbacks().allreduce)) return fail(c, "test transport: callbacks not registered");
  if (hipStreamCreateWithPriority(&c->halo, hipStreamNonBlocking, 0) != hipSuccess) return fail(c, "halo stream");   // (normal priority, as the product)
  if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) return fail(c, "events");
  if (hipHostMalloc((void**)&c->h_red, 1024 * sizeof(double), hipHostMallocDefault) != hipSuccess) return fail(c, "pinned reduction buffer");
  return 0;
}

inline void destroy(Comm* c) {
  if (!c) return;
  if (last_comm() == c) last_comm() = 0;
  if (c->stage_s) (void)hipFree(c->stage_s);
  if (c->stage_r) (void)hipFree(c->stage_r);
  if (c->h_s) (void)hipHostFree(c->h_s);
  if (c->h_r) (void)hipHostFree(c->h_r);
  if (c->h_red) (void)hipHostFree(c->h_red);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->halo) (void)hipStreamDestroy(c->halo);
  delete c;
}

inline size_t total_doubles(const P2P* ops, int nops, int send) {
  size_t n = 0;
  for (int i = 0; i < nops; ++i) if (ops[i].send == send) n += ops[i].count;
  return n;
}

// same contract as the RCCL transport: device stages sized at rgpu_comm_create; here also the pinned host images
inline int prepare_exchange(Comm* c, const P2P* ops, int nops) {
  if (nops == 0) return 0;
  const size_t ns = total_doubles(ops, nops, 1), nr = total_doubles(ops, nops, 0);
  const size_t need = ns > nr ? ns : nr;
  if (need > c->h_cap) {
    if (c->h_s) (void)hipHostFree(c->h_s);
    if (c->h_r) (void)hipHostFree(c->h_r);
    c->h_s = 0; c->h_r = 0; c->h_cap = 0;
    if (hipHostMalloc((void**)&c->h_s, need * sizeof(double), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_r, need * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return 1; }
    c->h_cap = need;
  }
  if (!c->pack) return 0;
  PackedExchange px;
  if (build_packed(ops, nops, &px)) return 1;
  if (px.pl.stage_doubles <= c->stage_cap) return 0;
  if (c->stage_s) (void)hipFree(c->stage_s);
  if (c->stage_r) (void)hipFree(c->stage_r);
  c->stage_s = 0; c->stage_r = 0; c->stage_cap = 0;
  if (hipMalloc((void**)&c->stage_s, px.pl.stage_doubles * sizeof(double)) != hipSuccess || hipMalloc((void**)&c->stage_r, px.pl.stage_doubles * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return 1; }
  c->stage_cap = px.pl.stage_doubles;
  return 0;
}
inline void disable_pack(Comm* c) { c->pack = false; }
inline bool packs(const Comm* c) { return c->pack; }

// Behind what the compute stream holds now: gather (packed) and copy the outgoing planes to pinned memory on the halo stream.
// Returns without blocking; the planes travel in exchange_wait.
inline int exchange_start(Comm* c, void* compute_stream, const P2P* ops, int nops) {
  hipStream_t cs = (hipStream_t)compute_stream;
  if (c->pending) return fail(c, "exchange_start: the previous exchange was not waited for");
  if (total_doubles(ops, nops, 1) > c->h_cap || total_doubles(ops, nops, 0) > c->h_cap) return fail(c, "exchange_start: host images were not prepared for this operation list");
  c->dev_ops.assign(ops, ops + nops);
  c->host_ops.clear();
  c->pending_packed = c->pack && build_packed(ops, nops, &c->px) == 0;
  // as in the product's transport: the pack kernel on the compute stream, ahead of the kernels the exchange overlaps with
  if (c->pending_packed && c->px.pl.stage_doubles > c->stage_cap) return fail(c, "packed exchange: the operation list outgrew the stages sized at create");
  if (c->pending_packed && launch_pack(c->px, c->stage_s, cs)) return fail(c, "pack kernel");
  if (hipEventRecord(c->ev_ready, cs) != hipSuccess || hipStreamWaitEvent(c->halo, c->ev_ready, 0) != hipSuccess) return fail(c, "event record / wait");
  if (c->pending_packed) {
    const PackPlan& pl = c->px.pl;
    size_t all_s = 0;
    for (int q = 0; q < pl.npeers; ++q) all_s += pl.send_total[q];
    if (all_s && hipMemcpyAsync(c->h_s, c->stage_s, all_s * sizeof(double), hipMemcpyDeviceToHost, c->halo) != hipSuccess) return fail(c, "D2H of the send stage");
    // the messages, in the RCCL transport's posting order: one send per peer, then one receive per peer
    for (int q = 0; q < pl.npeers; ++q) if (pl.send_total[q]) { const P2P o = {c->h_s + pl.send_base[q], pl.send_total[q], pl.peer[q], 1}; c->host_ops.push_back(o); }
    for (int q = 0; q < pl.npeers; ++q) if (pl.recv_total[q]) { const P2P o = {c->h_r + pl.recv_base[q], pl.recv_total[q], pl.peer[q], 0}; c->host_ops.push_back(o); }
  } else {   // in place: one message per chunk, in posting order
    size_t off_s = 0, off_r = 0;
    for (int i = 0; i < nops; ++i) {
      if (ops[i].send) {
        if (hipMemcpyAsync(c->h_s + off_s, ops[i].ptr, ops[i].count * sizeof(double), hipMemcpyDeviceToHost, c->halo) != hipSuccess) return fail(c, "D2H of a chunk");
        const P2P o = {c->h_s + off_s, ops[i].count, ops[i].peer, 1}; c->host_ops.push_back(o);
        off_s += ops[i].count;
      } else {
        const P2P o = {c->h_r + off_r, ops[i].count, ops[i].peer, 0}; c->host_ops.push_back(o);
        off_r += ops[i].count;
      }
    }
  }
  c->pending = true;
  return 0;
}

inline int exchange_wait(Comm* c, void* compute_stream) {
  hipStream_t cs = (hipStream_t)compute_stream;
  if (!c->pending) return 0;
  c->pending = false;
  if (hipStreamSynchronize(c->halo) != hipSuccess) return fail(c, "exchange_wait: halo stream");
  int rc = 0;
  if (c->nranks == 1) {   // ring of one: the n-th send is the n-th receive
    std::vector<P2P> sends, recvs;
    for (size_t i = 0; i < c->host_ops.size(); ++i) (c->host_ops[i].send ? sends : recvs).push_back(c->host_ops[i]);
    if (sends.size() != recvs.size()) return fail(c, "self ring: sends and receives do not pair up");
    for (size_t i = 0; i < sends.size(); ++i) {
      if (sends[i].count != recvs[i].count) return fail(c, "self ring: a send and its receive differ in size");
      std::memcpy(recvs[i].ptr, sends[i].ptr, sends[i].count * sizeof(double));
    }
  } else {
    rc = callbacks().exchange(c->host_ops.data(), (int)c->host_ops.size());
  }
  if (rc) return fail(c, "exchange callback failed");
  c->exchanges += 1;
  for (size_t i = 0; i < c->host_ops.size(); ++i) if (c->host_ops[i].send) c->messages += 1;
  if (c->pending_packed) {
    const PackPlan& pl = c->px.pl;
    size_t all_r = 0;
    for (int q = 0; q < pl.npeers; ++q) all_r += pl.recv_total[q];
    if (all_r && hipMemcpyAsync(c->stage_r, c->h_r, all_r * sizeof(double), hipMemcpyHostToDevice, c->halo) != hipSuccess) return fail(c, "H2D of the receive stage");
    if (launch_unpack(c->px, c->stage_r, c->halo)) return fail(c, "unpack kernel");
  } else {
    size_t k = 0;
    for (size_t i = 0; i < c->dev_ops.size(); ++i) {
      if (c->dev_ops[i].send) continue;
      while (k < c->host_ops.size() && c->host_ops[k].send) ++k;
      if (k == c->host_ops.size()) return fail(c, "exchange_wait: receive without a host image");
      if (hipMemcpyAsync(c->dev_ops[i].ptr, c->host_ops[k].ptr, c->dev_ops[i].count * sizeof(double), hipMemcpyHostToDevice, c->halo) != hipSuccess) return fail(c, "H2D of a chunk");
      ++k;
    }
  }
  if (hipEventRecord(c->ev_done, c->halo) != hipSuccess || hipStreamWaitEvent(cs, c->ev_done, 0) != hipSuccess) return fail(c, "exchange_wait: event");
  return 0;
}
inline double last_exchange_ms(Comm*) { return -1.0; }

// in place on a device buffer, queued behind `stream`: staged through pinned memory
inline int allreduce_max(Comm* c, double* d, int n, void* stream) {
  if (c->nranks == 1) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (n > 1024) return fail(c, "allreduce_max: more than 1024 values");
  if (hipMemcpyAsync(c->h_red, d, n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(c, "allreduce_max: D2H");
  if (callbacks().allreduce(c->h_red, n, 0)) return fail(c, "allreduce callback failed");
  if (hipMemcpyAsync(d, c->h_red, n * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(c, "allreduce_max: H2D");
  return 0;
}
inline int allreduce_sum_host(Comm* c, double* h, int n, void*) {
  if (c->nranks == 1) return 0;
  return callbacks().allreduce(h, n, 1) ? fail(c, "allreduce callback failed") : 0;
}
inline int barrier(Comm* c, void* s) { double z = 0; return allreduce_sum_host(c, &z, 1, s); }
inline int version(const Comm*) { return 0; }
inline void set_device(int d) { if (d >= 0) (void)hipSetDevice(d); }
inline int poison_slot(Comm* c, double* d, void* stream) {
  const unsigned long long inf_bits = 0x7ff0000000000000ull;
  if (hipMemcpyAsync(d, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(c, "poison_slot");
  return 0;
}
inline int info(Comm* c, int* nranks, int* rank, int* device, char* pci, int pci_len) {
  int d = -1;
  (void)hipGetDevice(&d);
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  if (device) *device = d;
  if (pci && pci_len > 0) { pci[0] = 0; if (hipDeviceGetPCIBusId(pci, pci_len, d) != hipSuccess) pci[0] = 0; }
  return 0;
}
inline void abort_comm(Comm*) {}

}  // namespace rgpu_transport

// registered once per process by the test worker before rgpu_comm_create (single translation unit: defined here)
extern "C" void rgpu_comm_test_set_callbacks(rgpu_transport::exchange_fn e, rgpu_transport::allreduce_fn a) {
  rgpu_transport::callbacks().exchange = e;
  rgpu_transport::callbacks().allreduce = a;
}
// what the last communicator of this process did: [0] exchanges completed, [1] messages sent, [2] 1 = packed exchange
extern "C" void rgpu_comm_test_stats(long* out3) {
  rgpu_transport::Comm* c = rgpu_transport::last_comm();
  out3[0] = c ? c->exchanges : -1; out3[1] = c ? c->messages : -1; out3[2] = c ? (c->pack ? 1 : 0) : -1;
}
