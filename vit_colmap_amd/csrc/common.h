// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "../../include/vitcolmap_hip.h"

namespace vc {

inline int& last_hip_error_slot() {
  static thread_local int e = 0;
  return e;
}
inline int fail(hipError_t e) {
  last_hip_error_slot() = (int)e;
  return VC_ERR_LAUNCH;
}
// Launch errors are reported synchronously by hipGetLastError(); execution errors surface at
// the caller's next synchronisation (the ABI never synchronises).
inline int check_launch() {
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? VC_OK : fail(e);
}

// hipFuncSetAttribute applies to the CURRENT device only, so "this kernel is configured" is remembered
// per device (one bit per device ordinal), not per thread: a process that drives several GPUs
// configures every kernel once on each.  Usage:
//   static vc::PerDeviceOnce once;
//   if (int st = once.run([] { return hipFuncSetAttribute(...); })) return st;
struct PerDeviceOnce {
  std::atomic<unsigned long long> done{0};
  template <typename F>
  int run(F configure) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail(e);
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return VC_OK;
    e = configure();   // idempotent: two threads racing here both set the same attribute
    if (e != hipSuccess) return fail(e);
    done.fetch_or(bit, std::memory_order_release);
    return VC_OK;
  }
};

}  // namespace vc
