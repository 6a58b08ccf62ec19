// link_hold.h -- MEASUREMENT ONLY, not part of the product: the emulated xGMI link of the one-GPU slab probe (scripts/slab_probe.py).
// Compiled into ramsesgpu_amd/librgpu_comm_measure.so by `build.py --measure` (-DRG_MEASURE -I scripts/measure); the shipped
// librgpu_comm.so does not contain it (csrc/hip/rg_transport.h has empty hooks instead).
//
// A one-GPU probe exchanges its halo planes with itself, device-local, in ~0.03 ms; with RGPU_COMM_EMULATE_GBPS=<rate> the halo stream is
// held for the time the same bytes would need on ONE xGMI link at that rate -- RGPU_COMM_EMULATE_PEERS = 2: the two neighbours are
// different GPUs (N >= 3: two links in parallel, the per-peer bytes count), 1: both neighbours are the same GPU (N = 2: all bytes
// over one link).  The hold is a one-thread kernel spinning on the constant-rate clock (round 4: a hipLaunchHostFunc sleep did NOT
// hold the stream on ROCm 7.0).  RGPU_COMM_EMULATE_MODE=parallel: the hold runs NEXT TO the device-local RCCL copy (on a stream of
// its own, joined before the unpack) instead of behind it -- on real links the copy IS the transfer.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

namespace rgpu_transport {

__global__ void emulated_link_hold(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

struct LinkHold {
  double gbps = 0.0; int peers = 2; long long wall_khz = 0; int parallel = 0;
  hipStream_t stream = 0; hipEvent_t ev0 = 0, ev1 = 0;
  long long ticks = 0; bool held = false;
};
inline int link_hold_init(LinkHold& h) {
  h.gbps = std::getenv("RGPU_COMM_EMULATE_GBPS") ? std::atof(std::getenv("RGPU_COMM_EMULATE_GBPS")) : 0.0;
  h.peers = std::getenv("RGPU_COMM_EMULATE_PEERS") ? std::atoi(std::getenv("RGPU_COMM_EMULATE_PEERS")) : 2;
  h.parallel = (std::getenv("RGPU_COMM_EMULATE_MODE") && std::strcmp(std::getenv("RGPU_COMM_EMULATE_MODE"), "parallel") == 0) ? 1 : 0;
  if (h.gbps > 0) {
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    h.wall_khz = khz > 0 ? khz : 100000;   // 100 MHz on gfx9
    if (h.parallel && (hipStreamCreateWithPriority(&h.stream, hipStreamNonBlocking, 0) != hipSuccess || hipEventCreateWithFlags(&h.ev0, hipEventDisableTiming) != hipSuccess ||
                       hipEventCreateWithFlags(&h.ev1, hipEventDisableTiming) != hipSuccess)) return -1;
  }
  return 0;
}
inline void link_hold_destroy(LinkHold& h) {
  if (h.ev0) (void)hipEventDestroy(h.ev0);
  if (h.ev1) (void)hipEventDestroy(h.ev1);
  if (h.stream) (void)hipStreamDestroy(h.stream);
}
// the exchange is about to send `bytes`: how long the link would be busy
inline void link_hold_begin(LinkHold& h, size_t bytes) {
  h.ticks = 0; h.held = false;
  if (h.gbps > 0) {
    const double ns = (double)bytes / (h.peers >= 2 ? 2.0 : 1.0) / h.gbps;   // bytes / (GB/s) = ns
    h.ticks = (long long)(ns * 1e-6 * (double)h.wall_khz);
  }
}
// packed exchange: the link time starts with the transfer, beside it ...
inline int link_hold_fork(LinkHold& h, hipStream_t halo) {
  if (!(h.ticks > 0 && h.parallel && h.stream)) return 0;
  if (hipEventRecord(h.ev0, halo) != hipSuccess || hipStreamWaitEvent(h.stream, h.ev0, 0) != hipSuccess) return -1;
  hipLaunchKernelGGL(emulated_link_hold, dim3(1), dim3(1), 0, h.stream, h.ticks);
  if (hipGetLastError() != hipSuccess || hipEventRecord(h.ev1, h.stream) != hipSuccess) return -1;
  h.held = true;
  return 0;
}
// ... and joins before the unpack
inline int link_hold_join(LinkHold& h, hipStream_t halo) {
  if (!h.held) return 0;
  return hipStreamWaitEvent(halo, h.ev1, 0) == hipSuccess ? 0 : -1;
}
// serial form: behind the device-local transfer (also the in-place exchange)
inline int link_hold_behind(LinkHold& h, hipStream_t halo) {
  if (!(h.ticks > 0) || h.held) return 0;
  hipLaunchKernelGGL(emulated_link_hold, dim3(1), dim3(1), 0, halo, h.ticks);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace rgpu_transport
