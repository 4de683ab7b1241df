// Library identification and error strings for libepipolar_hip.
#include "common.h"
#include <mutex>

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

// ---- deterministic mode: the reference's CUDNN.DETERMINISTIC key (lib/core/config.py:21, scripts/train.py:80) ----
// Off (default): BatchNorm batch sums and the BatchNorm-backward sums are accumulated with fp32 atomics from GEMM epilogues / reduction workgroups
// in whatever order the workgroups finish -- two runs of the same step differ in the last bits, and a flipped bf16 rounding early in the network
// moves the loss of a random-weight network by up to 0.6 %.
// On: (a) no GEMM epilogue accumulates column sums (launch_gemm withholds the fused statistics / reduction and reports *_done = 0, so the callers
// run their own passes); (b) those passes write one partial sum per workgroup into the scratch below and a second, single-pass kernel adds them in
// index order.  Costs ~60 extra launches per ResNet-50 step.  The scratch has two halves: one for the BatchNorm passes, which the callers must keep
// on ONE stream (the training path does: they are links of the forward / backward chain), one for epi_column_sums_* (the final layer's bias
// gradient, which the training path runs on its weight-gradient stream).  Bit-identical reruns: tests/test_hip_deterministic.py.
namespace epi {
static int g_det = 0;
static const size_t DET_FLOATS = (size_t)4 << 20;       // 16 MB per half: 1024 row blocks x 2 x 2048 channels
// ONE scratch per DEVICE, allocated the first time a launch on that device asks for it: a process that drives device 1 (GPUS: '1', or a rank whose
// launcher sets the device after epi_set_deterministic) must not write its partial sums into device 0's memory (round-4 advisor finding).
static const int DET_MAX_DEVICES = 64;
static float* g_det_buf[DET_MAX_DEVICES] = {};
static std::mutex g_det_mutex;
bool deterministic() { return g_det != 0; }
float* det_scratch(size_t floats, bool column_sums) {
    if (!g_det || floats > DET_FLOATS) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DET_MAX_DEVICES) return nullptr;
    float* buf = g_det_buf[dev];
    if (!buf) {
        std::lock_guard<std::mutex> lock(g_det_mutex);
        buf = g_det_buf[dev];
        if (!buf) {
            if (hipMalloc(&buf, 2 * DET_FLOATS * sizeof(float)) != hipSuccess) return nullptr;
            g_det_buf[dev] = buf;
        }
    }
    return buf + (column_sums ? DET_FLOATS : 0);
}
}  // namespace epi

extern "C" int epi_set_deterministic(int on) {
    const int before = epi::g_det;
    if (on < 0) return before;              // query
    epi::g_det = on ? 1 : 0;
    if (on && !epi::det_scratch(1, false)) { epi::g_det = before; return -1; }      // (the current device's scratch now; other devices' at their first use)
    return before;
}
