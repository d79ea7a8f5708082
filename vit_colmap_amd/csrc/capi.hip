// ABI bookkeeping entry points (no kernels).
#include "common.h"

extern "C" {

int vc_abi_version(void) { return VC_ABI_VERSION; }

const char* vc_status_string(int status) {
  switch (status) {
    case VC_OK: return "ok";
    case VC_ERR_INVALID_ARG: return "invalid argument";
    case VC_ERR_UNSUPPORTED: return "size not supported by the gfx950 kernels";
    case VC_ERR_LAUNCH: return "kernel launch failed (see vc_last_hip_error)";
    case VC_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown status";
  }
}

int vc_last_hip_error(void) { return vc::last_hip_error_slot(); }

}  // extern "C"
