// ABI version, build digest and error strings.
#include "dyk_common.h"
#include "build_sha.h"

extern "C" int dyk_abi_version(void) { return DYK_ABI_VERSION; }

extern "C" const char* dyk_build_sha(void) { return DYK_BUILD_SHA; }

extern "C" const char* dyk_error_string(int code) {
    switch (code) {
    case DYK_OK: return "ok";
    case DYK_ERR_ARG: return "invalid argument or unsupported shape";
    case DYK_ERR_HIP: return "HIP runtime call or kernel launch failed";
    case DYK_ERR_UNSUPPORTED: return "unsupported configuration";
    case DYK_ERR_STATE: return "object used in the wrong state";
    default: return "unknown error";
    }
}
