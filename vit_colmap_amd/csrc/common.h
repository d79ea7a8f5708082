// Shared host-side helpers for the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>

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

}  // namespace vc
