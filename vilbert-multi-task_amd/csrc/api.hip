// ABI version and error strings of libvilbert_hip.so.
#include "common.h"

namespace {
const uint64_t* g_seed_epoch = nullptr;
}
const uint64_t* vb_seed_epoch() { return g_seed_epoch; }

extern "C" int vb_set_seed_epoch(const uint64_t* device_counter) {
    g_seed_epoch = device_counter;
    return 0;
}

extern "C" int vb_abi_version(void) { return VB_ABI_VERSION; }

extern "C" const char* vb_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case VB_E_BADARG: return "VB_E_BADARG: null pointer or non-positive size";
        case VB_E_ALIGN: return "VB_E_ALIGN: pointer / leading dimension not 16-byte aligned or size not a multiple of 4";
        case VB_E_RANGE: return "VB_E_RANGE: size outside the compiled range";
        case VB_E_SEGMENT: return "VB_E_SEGMENT: bad weight-segment description";
        case VB_E_WORKSPACE: return "VB_E_WORKSPACE: deterministic mode - the registered workspace is too small for this launch";
        default: break;
    }
    if (code > 0) return hipGetErrorString(static_cast<hipError_t>(code));
    return "unknown vb error";
}
