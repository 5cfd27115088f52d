// opental_amd/csrc/core.hip -- ABI version + error strings for libopental_hip.so.
#include "common.h"

extern "C" int otal_abi_version(void) { return OTAL_ABI_VERSION; }

extern "C" const char* otal_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case OTAL_E_NULL: return "null pointer argument";
        case OTAL_E_SHAPE: return "non-positive or inconsistent size";
        case OTAL_E_ODD_C: return "BoundaryMaxPooling needs an even channel count";
        case OTAL_E_BATCH: return "segments batch differs from feature batch";
        case OTAL_E_DTYPE: return "unsupported dtype code";
        case OTAL_E_LEVELS: return "bad level table";
        case OTAL_E_UNSUPPORTED: return "configuration not supported by this build";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}
