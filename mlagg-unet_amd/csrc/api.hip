// Library-level entry points of include/mlagg_hip.h.
#include <hip/hip_runtime.h>

#include "mlagg_hip.h"

extern "C" const char *mlagg_version(void) { return "mlagg_hip 0.1 (gfx950)"; }

extern "C" const char *mlagg_error_string(int code)
{
    switch (code) {
    case 0: return "success";
    case MLAGG_E_UNSUPPORTED: return "unsupported shape for the gfx950 kernels";
    case MLAGG_E_NULLPTR: return "required pointer argument is NULL";
    case MLAGG_E_WORKSPACE: return "workspace too small";
    default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown mlagg error";
    }
}
