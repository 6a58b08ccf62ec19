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

namespace rgpu_transport {

struct Comm {
  ncclComm_t comm;
  hipStream_t halo;
  hipEvent_t ev_ready, ev_done, ev_begin;   // ev_begin / ev_done carry time stamps: duration of the last exchange on the halo stream
  double* scratch;   // device scratch for host-value reductions
  int rank, nranks;
  std::string err;
  // MEASUREMENT knob (scripts/slab_probe.py; off unless RGPU_COMM_EMULATE_GBPS is set): a one-GPU probe exchanges its halo planes
  // with itself, device-local, in ~0.03 ms; with the knob the halo stream is held for the time the same bytes would need on ONE
  // xGMI link at that rate -- RGPU_COMM_EMULATE_PEERS = 2: the two neighbours are different GPUs (N >= 3: two links in parallel,
  // the per-peer bytes count), 1: both neighbours are the same GPU (N = 2: all bytes over one link).  The hold is a one-thread
  // kernel on the halo stream spinning on the constant-rate clock (round 4: a hipLaunchHostFunc sleep did NOT hold the stream on
  // ROCm 7.0 -- a 51 ms "link" left the step time unchanged, gpurun_out/r4c/knob.log).
  double emulate_gbps; int emulate_peers; long long wall_khz;
  // RGPU_COMM_EMULATE_MODE=parallel: the hold runs NEXT TO the device-local RCCL copy (on a stream of its own, joined before the
  // unpack) instead of behind it -- on real links the copy IS the transfer, so the exchange takes max(local copy, link time), not
  // their sum; "serial" (default, rounds 4-5 tables) double-counts the ~0.2 ms local copy
  int emulate_parallel; hipStream_t hold_stream; hipEvent_t ev_hold0, ev_hold1;
  // Packed exchange (RGPU_COMM_PACK, default on): the chunks that go to one peer are gathered into ONE staging buffer by one small
  // kernel, sent / received as ONE operation per peer and direction, and scattered by a second kernel.  Round 4: RCCL turned the 32
  // in-place send / recv operations of one grouped exchange (a chunk per variable and face) into 8 kernel launches with ~40 us
  // between them (profiles/r04_slab_timeline.txt); two or four operations make one launch.
  bool pack;
  double* stage_s; double* stage_r; size_t stage_cap;   // doubles
};
__global__ void emulated_link_hold(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

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
  c->emulate_gbps = std::getenv("RGPU_COMM_EMULATE_GBPS") ? std::atof(std::getenv("RGPU_COMM_EMULATE_GBPS")) : 0.0;
  c->emulate_peers = std::getenv("RGPU_COMM_EMULATE_PEERS") ? std::atoi(std::getenv("RGPU_COMM_EMULATE_PEERS")) : 2;
  c->wall_khz = 0;
  c->emulate_parallel = (std::getenv("RGPU_COMM_EMULATE_MODE") && std::strcmp(std::getenv("RGPU_COMM_EMULATE_MODE"), "parallel") == 0) ? 1 : 0;
  c->hold_stream = 0; c->ev_hold0 = 0; c->ev_hold1 = 0;
  c->pack = !(std::getenv("RGPU_COMM_PACK") && std::atoi(std::getenv("RGPU_COMM_PACK")) == 0);
  c->stage_s = 0; c->stage_r = 0; c->stage_cap = 0;
  if (c->emulate_gbps > 0) {
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    c->wall_khz = khz > 0 ? khz : 100000;   // 100 MHz on gfx9
  }
  *out = c;
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) return fail(c, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
  // NORMAL priority (round 5).  Rounds 2-4 created the halo stream with the highest priority ("the exchange is short and on the
  // critical path of the neighbours"): on this runtime a priority stream takes one of the process's few hardware queues for itself and
  // the compute stream's kernels -- every one of them, the sweep included -- ran 5-9 % slower for as long as the communicator existed
  // (N = 8 slab, no link hold: 4.97 -> 4.55 ms per step with normal priority, 4.61 with GPU_MAX_HW_QUEUES=8 and high priority;
  // profiles/r05_halo_stream_priority.txt).  RGPU_HALO_PRIO=high|low for experiments.
  int prio = 0;
  if (const char* e = std::getenv("RGPU_HALO_PRIO")) prio = std::strcmp(e, "high") == 0 ? hi : std::strcmp(e, "low") == 0 ? lo : 0;
  if (hipStreamCreateWithPriority(&c->halo, hipStreamNonBlocking, prio) != hipSuccess) return fail(c, "halo stream");
  // ev_begin / ev_done carry time stamps (rgpu_comm_last_exchange_ms) unless RGPU_COMM_NO_TIMING=1 (ordering only)
  const unsigned tflag = (std::getenv("RGPU_COMM_NO_TIMING") && std::atoi(std::getenv("RGPU_COMM_NO_TIMING"))) ? hipEventDisableTiming : hipEventDefault;
  if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, tflag) != hipSuccess || hipEventCreateWithFlags(&c->ev_begin, tflag) != hipSuccess) return fail(c, "events");
  if (hipMalloc((void**)&c->scratch, 64 * sizeof(double)) != hipSuccess) return fail(c, "scratch");
  if (c->emulate_gbps > 0 && c->emulate_parallel &&
      (hipStreamCreateWithPriority(&c->hold_stream, hipStreamNonBlocking, 0) != hipSuccess || hipEventCreateWithFlags(&c->ev_hold0, hipEventDisableTiming) != hipSuccess ||
       hipEventCreateWithFlags(&c->ev_hold1, hipEventDisableTiming) != hipSuccess)) return fail(c, "hold stream");
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
  if (c->ev_hold0) (void)hipEventDestroy(c->ev_hold0);
  if (c->ev_hold1) (void)hipEventDestroy(c->ev_hold1);
  if (c->hold_stream) (void)hipStreamDestroy(c->hold_stream);
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
  // The pack kernel runs on the COMPUTE stream, ahead of the kernels the exchange overlaps with: alone it takes 0.03 ms for the 103 MB of a
  // 512^2 slab; on the halo stream, next to the inner update, it took 0.12 ms and the transfer started that much later -- N = 8 slab probe
  // with the link time beside the copy 4.44 against 4.52 ms per step, level without a link (profiles/r05_slab_pack_stream.txt).
  // RGPU_COMM_PACK_STREAM=halo: on the halo stream (rounds 4-5).  The stage is free: the compute stream has waited for the previous exchange.
  static const bool pack_on_compute = !(std::getenv("RGPU_COMM_PACK_STREAM") && std::string(std::getenv("RGPU_COMM_PACK_STREAM")) == "halo");
  PackedExchange px;
  const bool packed = c->pack && build_packed(ops, nops, &px) == 0;
  if (packed && px.pl.stage_doubles > c->stage_cap) return fail(c, "packed exchange: the operation list outgrew the stages sized at create");
  if (packed && pack_on_compute && launch_pack(px, c->stage_s, cs)) return fail(c, "pack kernel");
  if (hipEventRecord(c->ev_ready, cs) != hipSuccess || hipStreamWaitEvent(c->halo, c->ev_ready, 0) != hipSuccess) return fail(c, "event record / wait");
  if (hipEventRecord(c->ev_begin, c->halo) != hipSuccess) return fail(c, "event record");
  long long hold_ticks = 0;
  if (c->emulate_gbps > 0) {   // measurement knob, see Comm
    size_t sent = 0;
    for (int i = 0; i < nops; ++i) if (ops[i].send) sent += ops[i].count * sizeof(double);
    const double ns = (double)sent / (c->emulate_peers >= 2 ? 2.0 : 1.0) / c->emulate_gbps;   // bytes / (GB/s) = ns
    hold_ticks = (long long)(ns * 1e-6 * (double)c->wall_khz);
  }
  const bool hold_beside = hold_ticks > 0 && c->emulate_parallel && c->hold_stream;
  bool held = false;
  if (packed) {
    // per peer, in posting order: one region of the send stage and one of the receive stage (comm/pack_plan.h).  The stages were
    // sized and allocated by prepare_exchange at rgpu_comm_create (all ranks pack or none does): no allocation inside a step, where
    // a failure on one rank would leave its peers in ncclRecv
    const PackPlan& pl = px.pl;
    if (!pack_on_compute && launch_pack(px, c->stage_s, c->halo)) return fail(c, "pack kernel");
    if (hold_beside) {   // the emulated link time starts with the transfer
      if (hipEventRecord(c->ev_hold0, c->halo) != hipSuccess || hipStreamWaitEvent(c->hold_stream, c->ev_hold0, 0) != hipSuccess) return fail(c, "hold fork");
      hipLaunchKernelGGL(emulated_link_hold, dim3(1), dim3(1), 0, c->hold_stream, hold_ticks);
      if (hipGetLastError() != hipSuccess || hipEventRecord(c->ev_hold1, c->hold_stream) != hipSuccess) return fail(c, "emulated_link_hold");
      held = true;
    }
    ncclResult_t r = ncclGroupStart();
    for (int q = 0; q < pl.npeers && r == ncclSuccess; ++q)
      if (pl.send_total[q]) r = ncclSend(c->stage_s + pl.send_base[q], pl.send_total[q], ncclDouble, pl.peer[q], c->comm, c->halo);
    for (int q = 0; q < pl.npeers && r == ncclSuccess; ++q)
      if (pl.recv_total[q]) r = ncclRecv(c->stage_r + pl.recv_base[q], pl.recv_total[q], ncclDouble, pl.peer[q], c->comm, c->halo);
    const ncclResult_t re = ncclGroupEnd();
    if (r != ncclSuccess || re != ncclSuccess) return fail(c, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(r != ncclSuccess ? r : re));
    if (hold_beside && hipStreamWaitEvent(c->halo, c->ev_hold1, 0) != hipSuccess) return fail(c, "hold join");
    if (launch_unpack(px, c->stage_r, c->halo)) return fail(c, "unpack kernel");
  } else {   // in place: one operation per chunk (rounds 1-3)
    ncclResult_t r = ncclGroupStart();
    for (int i = 0; i < nops && r == ncclSuccess; ++i)
      r = ops[i].send ? ncclSend(ops[i].ptr, ops[i].count, ncclDouble, ops[i].peer, c->comm, c->halo)
                      : ncclRecv(ops[i].ptr, ops[i].count, ncclDouble, ops[i].peer, c->comm, c->halo);
    const ncclResult_t re = ncclGroupEnd();
    if (r != ncclSuccess || re != ncclSuccess) return fail(c, std::string("ncclSend / ncclRecv: ") + ncclGetErrorString(r != ncclSuccess ? r : re));
  }
  if (hold_ticks > 0 && !held) {   // serial form: behind the device-local transfer (also the in-place exchange)
    hipLaunchKernelGGL(emulated_link_hold, dim3(1), dim3(1), 0, c->halo, hold_ticks);
    if (hipGetLastError() != hipSuccess) return fail(c, "emulated_link_hold");
  }
  if (hipEventRecord(c->ev_done, c->halo) != hipSuccess) return fail(c, "event record");
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
