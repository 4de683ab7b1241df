// Library identification and error strings for libepipolar_hip.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

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
// index order.  Costs ~60 extra launches per ResNet-50 step.  The scratch is per stream (det_scratch below).  Bit-identical reruns:
// tests/test_hip_deterministic.py.
namespace epi {
static int g_det = 0;
static const size_t DET_FLOATS = (size_t)4 << 20;       // 16 MB per scratch: 1024 row blocks x 2 x 2048 channels
// ONE scratch per (DEVICE, STREAM), allocated the first time a launch on that pair asks for it.  Per device: a process that drives device 1 (GPUS: '1',
// or a rank whose launcher sets the device after epi_set_deterministic) must not write its partial sums into device 0's memory (round-4 advisor
// finding).  Per stream (round 6; rounds 4-5 had two fixed halves, "BatchNorm passes" and "column sums"): launches of one stream are ordered, so two
// of them never use the scratch at once, while launches of different streams -- the backward chain, the weight-gradient stream, the projection
// branch of a residual unit (csrc/torch_glue.cpp BranchScope) -- may run side by side.  The table is only touched under the mutex.
static std::map<std::pair<int, void*>, float*> g_det_buf;
static std::mutex g_det_mutex;
bool deterministic() { return g_det != 0; }
float* det_scratch(size_t floats, hipStream_t stream) {
    if (!g_det || floats > DET_FLOATS) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_det_mutex);
    float*& buf = g_det_buf[std::make_pair(dev, (void*)stream)];
    if (!buf && hipMalloc(&buf, DET_FLOATS * sizeof(float)) != hipSuccess) { buf = nullptr; return nullptr; }
    return buf;
}
}  // namespace epi

extern "C" int epi_set_deterministic(int on) {
    const int before = epi::g_det;
    if (on < 0) return before;              // query
    epi::g_det = on ? 1 : 0;
    if (on && !epi::det_scratch(1, (hipStream_t)nullptr)) { epi::g_det = before; return -1; }      // (the current device's scratch now; other devices' at their first use)
    return before;
}
