// fail_shim.cpp -- TEST-ONLY.  Failure injection for the slab driver on the DEVICE backend, without a line of it in the product:
// preloaded (LD_PRELOAD) into the rank processes of tests/test_comm_device.py, this library defines rgpu_step_fill_planes_pair -- one
// of the step pieces librgpu_comm*.so calls in librgpu*.so, once per step of the overlapped and boundary-first schedules -- and so stands
// between the two product libraries: the n-th call after rgpu_test_fail_after(n) returns RGPU_EHIP like a failed launch would, every
// other call goes to the real entry point (looked up in the product library named by RGPU_SHIM_REAL, which the process has loaded).
// The product's own calls inside librgpu*.so are bound inside that library (-Bsymbolic-functions) and never come here.
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>

struct rgpu_ctx;
typedef int (*fill_pair_fn)(rgpu_ctx*, int, double, double, int, int, int, int);

static int g_countdown = 0;
static int g_calls = 0;

extern "C" __attribute__((visibility("default"))) void rgpu_test_fail_after(int n) { g_countdown = n; }
extern "C" __attribute__((visibility("default"))) int rgpu_test_shim_calls(void) { return g_calls; }

extern "C" __attribute__((visibility("default"))) int rgpu_step_fill_planes_pair(rgpu_ctx* c, int nStep, double dt, double totalTime, int k_lo, int k_hi, int k_lo2, int k_hi2) {
  static fill_pair_fn real = 0;
  if (!real) {
    const char* path = std::getenv("RGPU_SHIM_REAL");
    void* h = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD) : 0;
    if (!h && path) h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    real = h ? (fill_pair_fn)dlsym(h, "rgpu_step_fill_planes_pair") : 0;
    if (!real) { std::fprintf(stderr, "fail_shim: cannot find the real rgpu_step_fill_planes_pair (RGPU_SHIM_REAL=%s)\n", path ? path : "(unset)"); std::abort(); }
  }
  ++g_calls;
  if (g_countdown > 0 && --g_countdown == 0) return -4;   // RGPU_EHIP (include/rgpu.h)
  return real(c, nStep, dt, totalTime, k_lo, k_hi, k_lo2, k_hi2);
}
