// Weighted joint-location loss (value + gradient w.r.t. predicted coordinates) for gfx950.
//
// Replaces lib/core/integral_loss.py:7-47 of the reference (weighted_mse_loss / weighted_l1_loss /
// weighted_smooth_l1_loss: 6-8 tiny launches + autograd) with one single-workgroup kernel.  The tensors
// are [B][3J] (1 632 floats at B=32, J=17): launch-latency bound, so everything is one launch.
#include "common.h"

namespace epi {

constexpr int JL_THREADS = 1024;

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(JL_THREADS) void joint_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                                const float* __restrict__ weight, int total, int B, int kind,
                                                                int norm, int size_average, float* __restrict__ loss,
                                                                float* __restrict__ grad) {
    __shared__ float red[32];
    const int tid = threadIdx.x;
    float n_p = 1.f, n_t = 1.f;
    if (norm) {                                   // torch.norm(x, 1) of the WHOLE tensor (integral_loss.py:9-11)
        float sp = 0.f, st = 0.f;
        for (int i = tid; i < total; i += JL_THREADS) { sp += fabsf(pred[i]); st += fabsf(target[i]); }
        n_p = block_sum(sp, red);
        n_t = block_sum(st, red);
    }
    const float inv_b = size_average ? 1.f / (float)B : 1.f;   // sum / len(input)  (:16,29,45)
    float acc = 0.f, dot = 0.f;
    for (int i = tid; i < total; i += JL_THREADS) {
        const float p = pred[i], w = weight[i];
        const float diff = p / n_p - target[i] / n_t;
        float l, g;
        const float a = fabsf(diff);
        if (kind == EPI_LOSS_L1) { l = a; g = sgnf(diff); }                                  // :26
        else if (kind == EPI_LOSS_L2) { l = diff * diff; g = 2.f * diff; }                   // :13
        else { const bool q = a < 1.f; l = q ? 0.5f * diff * diff : a - 0.5f; g = q ? diff : sgnf(diff); }   // :39-41
        acc += l * w;
        g *= w * inv_b;
        dot += g * p;
        if (grad && !norm) grad[i] = g;
    }
    acc = block_sum(acc, red);
    if (tid == 0) loss[0] = acc * inv_b;
    if (grad && norm) {
        // p_hat = p / n_p  ->  dL/dp_j = g_j / n_p - sign(p_j) * sum_i(g_i p_i) / n_p^2
        dot = block_sum(dot, red);
        for (int i = tid; i < total; i += JL_THREADS) {
            const float p = pred[i], w = weight[i];
            const float diff = p / n_p - target[i] / n_t;
            const float a = fabsf(diff);
            float g;
            if (kind == EPI_LOSS_L1) g = sgnf(diff);
            else if (kind == EPI_LOSS_L2) g = 2.f * diff;
            else g = (a < 1.f) ? diff : sgnf(diff);
            g *= w * inv_b;
            grad[i] = g / n_p - sgnf(p) * dot / (n_p * n_p);
        }
    }
}

}  // namespace epi

extern "C" int epi_joint_loss(const float* pred, const float* target, const float* weight, int B, int n, int kind, int norm,
                              int size_average, float* loss, float* grad_pred, epi_stream_t stream) {
    if (!pred || !target || !weight || !loss || B <= 0 || n <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (kind != EPI_LOSS_L1 && kind != EPI_LOSS_L2 && kind != EPI_LOSS_SMOOTH_L1) return EPI_ERR_INVALID_ARGUMENT;
    if ((long long)B * n > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::joint_loss_kernel, dim3(1), dim3(epi::JL_THREADS), 0, (hipStream_t)stream, pred, target, weight,
                       B * n, B, kind, norm, size_average, loss, grad_pred);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
