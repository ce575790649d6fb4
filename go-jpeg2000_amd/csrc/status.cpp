// status.cpp -- the status strings and the version of the C ABI (include/j2kgfx.h).  Host-only (no HIP), like t2.cpp and
// assemble.cpp: the three build into the sanitised libj2khost_asan.so of `make asan-host` as well.
#include "../../include/j2kgfx.h"

extern "C" const char *j2k_status_string(int s) {
    switch (s) {
        case J2K_OK: return "ok";
        case J2K_ERR_INVALID_ARG: return "invalid argument";
        case J2K_ERR_NO_DEVICE: return "no usable HIP device (this library has no CPU fallback)";
        case J2K_ERR_HIP: return "HIP runtime error";
        case J2K_ERR_CAPACITY: return "output capacity too small";
        case J2K_ERR_GO_PANIC: return "input on which the reference panics or never returns";
        case J2K_ERR_UNSUPPORTED: return "unsupported";
    }
    return "unknown status";
}
extern "C" const char *j2k_version(void) { return "j2kgfx 0.1 (gfx950)"; }
