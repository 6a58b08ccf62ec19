// rg_backend.h (TEST-ONLY host emulation) -- NOT part of the product.
//
// tests/ compile the per-cell kernel bodies of ramsesgpu_amd/csrc/kernels_*.h and the step driver against THIS
// header instead of ramsesgpu_amd/csrc/hip/rg_backend.h, with g++, so that the device logic (index ranges,
// gather order, compact-state reconstruction) can be unit-tested against the oracle on machines without a GPU.
// The resulting tests/_build/librgpu_emu.so is only ever loaded by tests marked "not gpu"; the shipped
// librgpu.so contains the HIP backend only and fails with RGPU_ENODEVICE when no GPU is present.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <string>

#define RG_DEVFN inline
#define RG_BACKEND_NAME "host-emulation(test-only)"
#define RG_SYNC_LAUNCH 1   // a "launch" is a host loop: its results are there when it returns (the step driver may read a StepClock record on the host)
#define RG_STREAM_STORE(ptr, val) (*(ptr) = (val))
#define RG_ASSUME(cond) ((void)0)

namespace rgpu_dev {
using std::signbit;
}

namespace rgpu {

typedef int rg_stream_t;
typedef double rg_event_t;

// host stand-ins of the shared-reciprocal helpers: plain IEEE operations (the device versions return the same bits)
struct rg_recip_t { double d; };
inline rg_recip_t rg_recip(double d) { rg_recip_t R; R.d = d; return R; }
inline double rg_div(double n, const rg_recip_t& R) { return n / R.d; }
inline void rg_recip2(double a, double b, rg_recip_t& A, rg_recip_t& B) { A.d = a; B.d = b; }
inline void rg_recip4(double a, double b, double c, double d, rg_recip_t& A, rg_recip_t& B, rg_recip_t& Cc, rg_recip_t& D) { A.d = a; B.d = b; Cc.d = c; D.d = d; }
inline double rg_sqrt(double x) { return std::sqrt(x); }
inline double rg_sqrt_pos(double x) { return std::sqrt(x); }
inline double rg_sqrt_radicand(double x) { return std::sqrt(x); }
inline rg_recip_t rg_recip_sqrt_pos(double x) { return rg_recip(std::sqrt(x)); }

inline bool rg_wave_any(bool pred) { return pred; }   // one lane per "wave" on the host

inline void rg_slot_max(unsigned long long* slot, double v) {
  double cur;
  std::memcpy(&cur, slot, sizeof(double));
  if (v > cur) std::memcpy(slot, &v, sizeof(double));
}
inline void rg_slot_max_wave(unsigned long long* slot, double v) { rg_slot_max(slot, v); }
enum { RG_DT_SLOTS = 1024 };

template <int BLOCK, int MINW = 1, class K>
inline int rg_launch(rg_stream_t, unsigned n, const K& k) {
  for (unsigned idx = 0; idx < n; ++idx) k(idx);
  return 0;
}
template <int BLOCK, int MINW = 1, class K>
inline int rg_launch_range(rg_stream_t, unsigned idx0, unsigned n, const K& k) {
  for (unsigned off = 0; off < n; ++off) k(idx0 + off);
  return 0;
}
// failure injection (tests/test_comm_driver.py: a step piece that fails on ONE rank): the n-th plane-range launch from now fails
// once, like a launch error would; 0 = disarmed.  Armed through rgpu_emu_fail_launch_after() below.
inline int& rg_fail_countdown() { static int n = 0; return n; }
template <int BLOCK, int MINW = 1, class K>
inline int rg_launch_planes(rg_stream_t, unsigned idx0, unsigned plane_cells, unsigned nplanes, const K& k, unsigned = 0) {
  if (rg_fail_countdown() > 0 && --rg_fail_countdown() == 0) return -1;
  for (unsigned off = 0; off < plane_cells * nplanes; ++off) k(idx0 + off);
  return 0;
}
template <class K>
inline int rg_reduce_max(rg_stream_t, unsigned n, const K& k, unsigned long long* out, unsigned idx0 = 0, bool reset = true) {
  double v = 0.0;
  if (!reset) std::memcpy(&v, out, sizeof(double));
  for (unsigned off = 0; off < n; ++off) v = std::fmax(v, k(idx0 + off));
  std::memcpy(out, &v, sizeof(double));
  return 0;
}
inline int rg_device_count() { return 1; }
inline int rg_current_device() { return 0; }
inline void rg_set_device(int) {}
inline int rg_pointer_device(const void*) { return -1; }
inline int rg_malloc(void** p, size_t bytes) { *p = std::malloc(bytes ? bytes : 1); return *p ? 0 : -1; }
inline void rg_free(void* p) { std::free(p); }
inline int rg_host_alloc(void** p, size_t bytes) { return rg_malloc(p, bytes); }
inline void rg_host_free(void* p) { std::free(p); }
inline int rg_memset_async(void* p, int v, size_t bytes, rg_stream_t) { std::memset(p, v, bytes); return 0; }
inline int rg_copy_h2d(void* d, const void* h, size_t bytes, rg_stream_t) { std::memcpy(d, h, bytes); return 0; }
inline int rg_copy_d2h(void* h, const void* d, size_t bytes, rg_stream_t) { std::memcpy(h, d, bytes); return 0; }
inline int rg_copy_d2d(void* d, const void* s_, size_t bytes, rg_stream_t) { std::memcpy(d, s_, bytes); return 0; }
inline rg_stream_t rg_stream_from_handle(void*) { return 0; }
inline void* rg_stream_to_handle(rg_stream_t) { return 0; }
inline int rg_stream_sync(rg_stream_t) { return 0; }
inline const char* rg_last_error_string() { return "emulation"; }
inline int rg_stream_create(rg_stream_t* s, int = 0) { *s = 1; return 0; }
inline void rg_stream_destroy(rg_stream_t) {}
inline int rg_order_event_create(rg_event_t* e) { *e = 0; return 0; }
inline int rg_stream_wait_event(rg_stream_t, rg_event_t) { return 0; }
inline int rg_event_create(rg_event_t* e) { *e = 0; return 0; }
inline void rg_event_destroy(rg_event_t) {}
inline int rg_event_record(rg_event_t, rg_stream_t) { return 0; }
inline double rg_event_elapsed_ms(rg_event_t, rg_event_t) { return 0.0; }

}  // namespace rgpu

// TEST-ONLY entry point of the emulation library (single translation unit: defined here)
extern "C" inline __attribute__((used, visibility("default"))) void rgpu_emu_fail_launch_after(int n) { rgpu::rg_fail_countdown() = n; }
