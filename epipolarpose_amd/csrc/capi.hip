// Library identification and error strings for libepipolar_hip.
#include "common.h"

extern "C" const char* epi_version(void) { return "epipolar_hip 0.1.0 (gfx950)"; }

extern "C" const char* epi_status_string(int status) {
    switch (status) {
        case EPI_OK: return "ok";
        case EPI_ERR_INVALID_ARGUMENT: return "invalid argument";
        case EPI_ERR_UNSUPPORTED: return "unsupported shape or dtype";
        case EPI_ERR_WORKSPACE: return "workspace too small";
        case EPI_ERR_LAUNCH: return "kernel launch failed";
        default: return "unknown status";
    }
}
