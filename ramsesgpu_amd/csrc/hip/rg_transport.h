// rg_transport.h (HIP / gfx950) -- RCCL transport of the z-slab driver (csrc/comm/rgpu_comm.cpp).
// One communicator per process; point-to-point halo traffic on a dedicated stream, ordered against the compute stream
// with events; collectives on the compute stream itself.  On MI355X RCCL moves the planes over xGMI peer links.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "halo_pack.h"   // P2P, the staging plan and the copy kernels of the packed exchange

#define RG_TRANSPORT_NAME "rccl"

// The one-GPU slab probe (scripts/slab_probe.py) holds the halo stream for the time a real xGMI link would need: measurement code,
// compiled in by `build.py --measure` only (scripts/measure/link_hold.h).  The product has these empty hooks.
#ifdef RG_MEASURE
#include "link_hold.h"
#else
namespace rgpu_transport {
struct LinkHold {};
inline int link_hold_init(LinkHold&) { return 0; }
inline void link_hold_destroy(LinkHold&) {}
inline void link_hold_begin(LinkHold&, size_t) {}
inline int link_hold_fork(LinkHold&, hipStream_t) { return 0; }
inline int link_hold_join(LinkHold&, hipStream_t) { return 0; }
inline int link_hold_behind(LinkHold&, hipStream_t) { return 0; }
}  // namespace rgpu_transport
#endif

namespace rgpu_transport {

struct Comm {
  ncclComm_t comm;
  hipStream_t halo;
  hipEvent_t ev_ready, ev_done, ev_begin;   // ev_begin / ev_done carry time stamps: duration of the last exchange on the halo stream
  double* scratch;   // device scratch for host-value reductions
  int rank, nranks;
  std::string err;
  LinkHold hold;   // (measurement builds only: see above)
  // RGPU_COMM_ONE_STREAM=1: send / recv are issued on the COMPUTE stream instead of the halo stream -- no overlap, and no second stream
  // driving the communicator: the fallback if RCCL's ordering between two streams of one communicator misbehaves on real links
  bool one_stream;
  int rccl_version;
  // Packed exchange (RGPU_COMM_PACK, default on): the chunks that go to one peer are gathered into ONE staging buffer by one small
  // kernel, sent / received as ONE operation per peer and direction, and scattered by a second kernel.  Round 4: RCCL turned the 32
  // in-place send / recv operations of one grouped exchange (a chunk per variable and face) into 8 kernel launches with ~40 us
  // between them (profiles/r04_slab_timeline.txt); two or four operations make one launch.
  bool pack;
  double* stage_s; double* stage_r; size_t stage_cap;   // doubles
};
inline int fail(Comm* c, const std::string& m) { if (c) c->err = m; return -1; }

inline int unique_id(char* id128) {
  static_assert(sizeof(ncclUniqueId) <= 128, "ncclUniqueId does not fit RGPU_COMM_ID_BYTES");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -1;
  std::memset(id128, 0, 128);
  std::memcpy(id128, &id, sizeof(id));
  return 0;
}

inline int create(Comm** out, int rank, int nranks, const char* id128) {
  Comm* c = new Comm();
  c->comm = 0; c->halo = 0; c->ev_ready = 0; c->ev_done = 0; c->ev_begin = 0; c->scratch = 0; c->rank = rank; c->nranks = nranks;
  const char* e_pack = std::getenv("RGPU_COMM_PACK");
  c->pack = !(e_pack && std::atoi(e_pack) == 0);
  const char* e_one = std::getenv("RGPU_COMM_ONE_STREAM");
  c->one_stream = e_one && std::atoi(e_one) != 0;
  c->rccl_version = 0;
  c->stage_s = 0; c->stage_r = 0; c->stage_cap = 0;
  *out = c;
  if (link_hold_init(c->hold)) return fail(c, "link hold (measurement build)");
  (void)ncclGetVersion(&c->rccl_version);
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) return fail(c, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  {   // the communicator RCCL built is the one asked for
    int n = 0, me = -1;
    if (ncclCommCount(c->comm, &n) != ncclSuccess || ncclCommUserRank(c->comm, &me) != ncclSuccess || n != nranks || me != rank)
      return fail(c, "ncclCommInitRank: the communicator reports " + std::to_string(n) + " ranks / rank " + std::to_string(me) + ", asked for " +
                         std::to_string(nranks) + " / " + std::to_string(rank));
  }
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  // NORMAL priority (round 5).  Rounds 2-4 created the halo stream with the highest priority ("the exchange is short and on the
  // critical path of the neighbours"): on this runtime a priority stream takes one of the process's few hardware queues for itself and
  // the compute stream's kernels -- every one of them, the sweep included -- ran 5-9 % slower for as long as the communicator existed
  // (N = 8 slab, no link hold: 4.97 -> 4.55 ms per step with normal priority, 4.61 with GPU_MAX_HW_QUEUES=8 and high priority;
  // profiles/r05_halo_stream_priority.txt).  RGPU_HALO_PRIO=high|low overrides it.
  int prio = 0;
  if (const char* e = std::getenv("RGPU_HALO_PRIO")) prio = std::strcmp(e, "high") == 0 ? hi : std::strcmp(e, "low") == 0 ? lo : 0;
  if (hipStreamCreateWithPriority(&c->halo, hipStreamNonBlocking, prio) != hipSuccess) return fail(c, "halo stream");
  const unsigned tflag = hipEventDefault;   // ev_begin / ev_done carry time stamps (rgpu_comm_last_exchange_ms)
  if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, tflag) != hipSuccess || hipEventCreateWithFlags(&c->ev_begin, tflag) != hipSuccess) return fail(c, "events");
  if (hipMalloc((void**)&c->scratch, 64 * sizeof(double)) != hipSuccess) return fail(c, "scratch");
  return 0;
}

inline void destroy(Comm* c) {
  if (!c) return;
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->stage_s) (void)hipFree(c->stage_s);
  if (c->stage_r) (void)hipFree(c->stage_r);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  link_hold_destroy(c->hold);
  if (c->halo) (void)hipStreamDestroy(c->halo);
  if (c->comm) (void)ncclCommDestroy(c->comm);
  delete c;
}

// Called once per operation list at rgpu_comm_create: size and allocate the two stages of the packed exchange (each holds what a
// rank sends per exchange: 2 x 103 MB at 512 x 1024 x 64 per rank).  0 = ready (or nothing to pack); 1 = the list does not fit a
// plan or the allocation failed -- not an error: the driver then takes every rank to the in-place exchange (disable_pack), because
// packed and in-place ranks would post different message counts.
inline int prepare_exchange(Comm* c, const P2P* ops, int nops) {
  if (!c->pack || nops == 0) return 0;
  PackedExchange px;
  if (build_packed(ops, nops, &px)) return 1;
  if (px.pl.stage_doubles <= c->stage_cap) return 0;
  if (c->stage_s) (void)hipFree(c->stage_s);
  if (c->stage_r) (void)hipFree(c->stage_r);
  c->stage_s = 0; c->stage_r = 0; c->stage_cap = 0;
  if (hipMalloc((void**)&c->stage_s, px.pl.stage_doubles * sizeof(double)) != hipSuccess || hipMalloc((void**)&c->stage_r, px.pl.stage_doubles * sizeof(double)) != hipSuccess) {
    (void)hipGetLastError();
    if (c->stage_s) (void)hipFree(c->stage_s);
    c->stage_s = 0; c->stage_r = 0;
    return 1;
  }
  c->stage_cap = px.pl.stage_doubles;
  return 0;
}
inline void disable_pack(Comm* c) {
  c->pack = false;
  if (c->stage_s) (void)hipFree(c->stage_s);
  if (c->stage_r) (void)hipFree(c->stage_r);
  c->stage_s = 0; c->stage_r = 0; c->stage_cap = 0;
}
inline bool packs(const Comm* c) { return c->pack; }

// all ops as ONE group on the halo stream, behind what the compute stream holds now
inline int exchange_start(Comm* c, void* compute_stream, const P2P* ops, int nops) {
  hipStream_t cs = (hipStream_t)compute_stream;
  if (!c->one_stream && (void*)c->halo == compute_stream) return fail(c, "the halo stream IS the compute stream");
  hipStream_t hs = c->one_stream ? cs : c->halo;   // where the transfers are issued
  // The pack kernel runs on the COMPUTE stream, ahead of the kernels the exchange overlaps with: alone it takes 0.03 ms for the 103 MB of a
  // 512^2 slab; on the halo stream, next to the inner update, it took 0.12 ms and the transfer started that much later -- N = 8 slab probe
  // with the link time beside the copy 4.44 against 4.52 ms per step, level without a link (profiles/r05_slab_pack_stream.txt).
  // The stage is free: the compute stream has waited for the previous exchange.
  PackedExchange px;
  const bool packed = c->pack && build_packed(ops, nops, &px) == 0;
  if (packed && px.pl.stage_doubles > c->stage_cap) return fail(c, "packed exchange: the operation list outgrew the stages sized at create");
  if (packed && launch_pack(px, c->stage_s, cs)) return fail(c, "pack kernel");
  if (hipEventRecord(c->ev_ready, cs) != hipSuccess || hipStreamWaitEvent(hs, c->ev_ready, 0) != hipSuccess) return fail(c, "event record / wait");
  if (hipEventRecord(c->ev_begin, hs) != hipSuccess) return fail(c, "event record");
  {
    size_t sent = 0;
    for (int i = 0; i < nops; ++i) if (ops[i].send) sent += ops[i].count * sizeof(double);
    link_hold_begin(c->hold, sent);
  }
  if (packed) {
    // per peer, in posting order: one region of the send stage and one of the receive stage (comm/pack_plan.h).  The stages were
    // sized and allocated by prepare_exchange at rgpu_comm_create (all ranks pack or none does): no allocation inside a step, where
    // a failure on one rank would leave its peers in ncclRecv
    const PackPlan& pl = px.pl;
    if (link_hold_fork(c->hold, hs)) return fail(c, "link hold");
    ncclResult_t r = ncclGroupStart();
    for (int q = 0; q < pl.npeers && r == ncclSuccess; ++q)
      if (pl.send_total[q]) r = ncclSend(c->stage_s + pl.send_base[q], pl.send_total[q], ncclDouble, pl.peer[q], c->comm, hs);
    for (int q = 0; q < pl.npeers && r == ncclSuccess; ++q)
      if (pl.recv_total[q]) r = ncclRecv(c->stage_r + pl.recv_base[q], pl.recv_total[q], ncclDouble, pl.peer[q], c->comm, hs);
    const ncclResult_t re = ncclGroupEnd();
    if (r != ncclSuccess || re != ncclSuccess) return fail(c, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(r != ncclSuccess ? r : re));
    if (link_hold_join(c->hold, hs)) return fail(c, "link hold");
    if (launch_unpack(px, c->stage_r, hs)) return fail(c, "unpack kernel");
  } else {   // in place: one operation per chunk (rounds 1-3)
    ncclResult_t r = ncclGroupStart();
    for (int i = 0; i < nops && r == ncclSuccess; ++i)
      r = ops[i].send ? ncclSend(ops[i].ptr, ops[i].count, ncclDouble, ops[i].peer, c->comm, hs)
                      : ncclRecv(ops[i].ptr, ops[i].count, ncclDouble, ops[i].peer, c->comm, hs);
    const ncclResult_t re = ncclGroupEnd();
    if (r != ncclSuccess || re != ncclSuccess) return fail(c, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(r != ncclSuccess ? r : re));
  }
  if (link_hold_behind(c->hold, hs)) return fail(c, "link hold");
  if (hipEventRecord(c->ev_done, hs) != hipSuccess) return fail(c, "event record");
  return 0;
}
inline int exchange_wait(Comm* c, void* compute_stream) {
  return hipStreamWaitEvent((hipStream_t)compute_stream, c->ev_done, 0) == hipSuccess ? 0 : fail(c, "stream wait");
}

// duration of the last exchange on the halo stream (from the moment the compute stream released it to the last plane received),
// for diagnosing a multi-GPU run; blocks until that exchange is complete.  < 0: none yet
inline double last_exchange_ms(Comm* c) {
  float ms = -1.0f;
  if (hipEventSynchronize(c->ev_done) != hipSuccess || hipEventElapsedTime(&ms, c->ev_begin, c->ev_done) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
  return (double)ms;
}

// in place on a device buffer, queued on `stream`
inline int allreduce_max(Comm* c, double* d, int n, void* stream) {
  const ncclResult_t r = ncclAllReduce(d, d, (size_t)n, ncclDouble, ncclMax, c->comm, (hipStream_t)stream);
  return r == ncclSuccess ? 0 : fail(c, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
}
// host values: through the device scratch (64 doubles at a time), synchronous
inline int allreduce_sum_host(Comm* c, double* h, int n, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n > 64) {
    for (int o = 0; o < n; o += 64)
      if (allreduce_sum_host(c, h + o, n - o < 64 ? n - o : 64, stream)) return -1;
    return 0;
  }
  if (hipMemcpyAsync(c->scratch, h, n * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) return fail(c, "H2D");
  const ncclResult_t r = ncclAllReduce(c->scratch, c->scratch, (size_t)n, ncclDouble, ncclSum, c->comm, s);
  if (r != ncclSuccess) return fail(c, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  if (hipMemcpyAsync(h, c->scratch, n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return fail(c, "D2H");
  return 0;
}
inline int version(const Comm* c) { return c->rccl_version; }
inline void set_device(int d) { if (d >= 0) (void)hipSetDevice(d); }
// +inf into a device double (a rank in an error state poisons its 1/dt before the MAX all-reduce: every rank then sees it)
inline int poison_slot(Comm* c, double* d, void* stream) {
  const unsigned long long inf_bits = 0x7ff0000000000000ull;
  if (hipMemcpyAsync(d, &inf_bits, sizeof(inf_bits), hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return fail(c, "poison_slot");
  return 0;
}
// what RCCL itself says about the communicator (not what the caller asked for): ranks, this rank, the HIP device the
// communicator is bound to, and that device's PCI bus id.  The reference binds one MPI rank to one device
// (HydroMpiParameters.cpp:196-201); a scaling run can prove it did from these.
inline int info(Comm* c, int* nranks, int* rank, int* device, char* pci, int pci_len) {
  int n = 0, r = -1, d = -1;
  if (ncclCommCount(c->comm, &n) != ncclSuccess || ncclCommUserRank(c->comm, &r) != ncclSuccess || ncclCommCuDevice(c->comm, &d) != ncclSuccess)
    return fail(c, "ncclCommCount / ncclCommUserRank / ncclCommCuDevice");
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (device) *device = d;
  if (pci && pci_len > 0) { pci[0] = 0; if (hipDeviceGetPCIBusId(pci, pci_len, d) != hipSuccess) pci[0] = 0; }
  return 0;
}
// a rank that failed outside a collective tells the others by aborting the communicator: their pending / next RCCL call
// returns an error instead of waiting for ever
inline void abort_comm(Comm* c) { if (c && c->comm) { (void)ncclCommAbort(c->comm); c->comm = 0; } }
inline int barrier(Comm* c, void* stream) {
  double z = 0.0;
  return allreduce_sum_host(c, &z, 1, stream);
}

}  // namespace rgpu_transport
