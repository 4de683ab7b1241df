// Training-mode BatchNorm2d + ReLU on NHWC bf16 activations for the deconvolution head (gfx950).
//
// Replaces `nn.BatchNorm2d(momentum=0.1)` + `nn.ReLU` after each ConvTranspose2d of the reference
// (lib/models/pose3d_resnet.py:180-181).  All kernels are HBM streaming kernels over x [R][C]
// (R = B*H*W rows, C channels contiguous), 16 bytes (8 channels) per lane per access, with the
// per-channel reductions done in registers -> LDS -> one fp32 atomic per channel per workgroup.
//   forward : stats (1 read) -> finalize (tiny) -> apply (1 read + 1 write)
//   backward: reduce (2 reads) -> apply (2 reads + 1 write)
#include "common.h"

namespace epi {

constexpr int BN_THREADS = 256;
constexpr int BN_ROWS_PER_WG = 256;

// sums[0..C) = sum x, sums[C..2C) = sum x^2      (pre-zeroed by the caller)
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const unsigned short* __restrict__ x, long long R, int C,
                                                              float* __restrict__ sums) {
    extern __shared__ float red[];                 // [rlanes][2][C]
    const int cg = C >> 3;                          // 8-channel groups per row
    const int rlanes = BN_THREADS / cg;             // rows processed concurrently by the workgroup
    const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
    const long long r0 = (long long)blockIdx.x * BN_ROWS_PER_WG;
    const long long r1 = (r0 + BN_ROWS_PER_WG < R) ? r0 + BN_ROWS_PER_WG : R;
    if (rl < rlanes) {
        for (long long r = r0 + rl; r < r1; r += rlanes) {
            float v[8];
            Elem<unsigned short>::load(x + r * C + g * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += v[k]; q[k] += v[k] * v[k]; }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[(rl * 2 + 0) * C + g * 8 + k] = s[k]; red[(rl * 2 + 1) * C + g * 8 + k] = q[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += BN_THREADS) {
        const int which = c / C, ch = c - which * C;
        float t = 0.f;
        for (int l = 0; l < rlanes; ++l) t += red[(l * 2 + which) * C + ch];
        atomicAdd(sums + c, t);
    }
}

// mean / rstd, running statistics (momentum, unbiased variance), fused affine scale/shift
__global__ void bn_finalize_kernel(const float* __restrict__ sums, long long R, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ rstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = (double)sums[c] / (double)R;
    double var = (double)sums[C + c] / (double)R - m * m;
    if (var < 0) var = 0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = rs;
    const float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    if (running_mean) {
        const double unbiased = (R > 1) ? var * (double)R / (double)(R - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// y = relu(x * scale[c] + shift[c])
__global__ __launch_bounds__(BN_THREADS) void bn_relu_apply_kernel(const unsigned short* __restrict__ x, long long nvec, int C,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   unsigned short* __restrict__ y) {
    const int cg = C >> 3;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * BN_THREADS) {
        const int g = (int)(i % cg);
        float v[8];
        Elem<unsigned short>::load(x + i * 8, v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] * scale[g * 8 + k] + shift[g * 8 + k], 0.f);
        Elem<unsigned short>::store(y + i * 8, v);
    }
}

// sums[0..C) = sum dz, sums[C..2C) = sum dz * xhat,  dz = dy * (x*scale+shift > 0),  xhat = (x - mean) * rstd
__global__ __launch_bounds__(BN_THREADS) void bn_relu_bwd_reduce_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                        long long R, int C, const float* __restrict__ scale,
                                                                        const float* __restrict__ shift, const float* __restrict__ mean,
                                                                        const float* __restrict__ rstd, float* __restrict__ sums) {
    extern __shared__ float red[];
    const int cg = C >> 3;
    const int rlanes = BN_THREADS / cg;
    const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
    float s[8], q[8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        s[k] = 0.f; q[k] = 0.f;
        sc[k] = scale[g * 8 + k]; sh[k] = shift[g * 8 + k]; mu[k] = mean[g * 8 + k]; rs[k] = rstd[g * 8 + k];
    }
    const long long r0 = (long long)blockIdx.x * BN_ROWS_PER_WG;
    const long long r1 = (r0 + BN_ROWS_PER_WG < R) ? r0 + BN_ROWS_PER_WG : R;
    if (rl < rlanes) {
        for (long long r = r0 + rl; r < r1; r += rlanes) {
            float xv[8], gv[8];
            Elem<unsigned short>::load(x + r * C + g * 8, xv);
            Elem<unsigned short>::load(dy + r * C + g * 8, gv);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dz = (xv[k] * sc[k] + sh[k] > 0.f) ? gv[k] : 0.f;
                s[k] += dz;
                q[k] += dz * (xv[k] - mu[k]) * rs[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { red[(rl * 2 + 0) * C + g * 8 + k] = s[k]; red[(rl * 2 + 1) * C + g * 8 + k] = q[k]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += BN_THREADS) {
        const int which = c / C, ch = c - which * C;
        float t = 0.f;
        for (int l = 0; l < rlanes; ++l) t += red[(l * 2 + which) * C + ch];
        atomicAdd(sums + c, t);
    }
}

// dx = gamma*rstd * (dz - dbeta/R - xhat * dgamma/R);  also emits dgamma = sums[C+c], dbeta = sums[c] (fp32)
__global__ __launch_bounds__(BN_THREADS) void bn_relu_bwd_apply_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                       long long nvec, long long R, int C, const float* __restrict__ gamma,
                                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                       const float* __restrict__ sums, unsigned short* __restrict__ dx) {
    const int cg = C >> 3;
    const float inv_r = 1.f / (float)R;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * BN_THREADS) {
        const int g = (int)(i % cg);
        float xv[8], gv[8], o[8];
        Elem<unsigned short>::load(x + i * 8, xv);
        Elem<unsigned short>::load(dy + i * 8, gv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = g * 8 + k;
            const float dz = (xv[k] * scale[c] + shift[c] > 0.f) ? gv[k] : 0.f;
            const float xh = (xv[k] - mean[c]) * rstd[c];
            o[k] = gamma[c] * rstd[c] * (dz - sums[c] * inv_r - xh * sums[C + c] * inv_r);
        }
        Elem<unsigned short>::store(dx + i * 8, o);
    }
}

static inline bool bn_shape_ok(long long R, int C) { return R > 0 && C > 0 && C % 8 == 0 && (C >> 3) <= BN_THREADS; }
static inline unsigned stream_grid(long long nvec) {
    const long long want = (nvec + BN_THREADS - 1) / BN_THREADS;
    return (unsigned)(want < 4096 ? want : 4096);
}

}  // namespace epi

using namespace epi;

extern "C" int epi_bn_relu_fwd(const void* x, long long R, int C, const float* gamma, const float* beta, float eps, float momentum,
                               float* running_mean, float* running_var, float* mean, float* rstd, float* scale_shift,
                               float* sums_ws, void* y, epi_stream_t stream) {
    if (!x || !gamma || !beta || !mean || !rstd || !scale_shift || !sums_ws || !y) return EPI_ERR_INVALID_ARGUMENT;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EPI_ERR_INVALID_ARGUMENT;
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums_ws, 0, 2 * (size_t)C * sizeof(float), st) != hipSuccess) return EPI_ERR_LAUNCH;
    const int rlanes = BN_THREADS / (C >> 3);
    const unsigned gridr = (unsigned)((R + BN_ROWS_PER_WG - 1) / BN_ROWS_PER_WG);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(gridr), dim3(BN_THREADS), (size_t)rlanes * 2 * C * sizeof(float), st,
                       (const unsigned short*)x, R, C, sums_ws);
    EPI_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, st, sums_ws, R, C, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, rstd, scale_shift, scale_shift + C);
    EPI_CHECK_LAUNCH();
    const long long nvec = R * (C >> 3);
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(stream_grid(nvec)), dim3(BN_THREADS), 0, st, (const unsigned short*)x, nvec, C,
                       scale_shift, scale_shift + C, (unsigned short*)y);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_bn_relu_bwd(const void* dy, const void* x, long long R, int C, const float* gamma, const float* mean,
                               const float* rstd, const float* scale_shift, float* dgamma_dbeta_ws, void* dx, epi_stream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !scale_shift || !dgamma_dbeta_ws || !dx) return EPI_ERR_INVALID_ARGUMENT;
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dgamma_dbeta_ws, 0, 2 * (size_t)C * sizeof(float), st) != hipSuccess) return EPI_ERR_LAUNCH;
    const int rlanes = BN_THREADS / (C >> 3);
    const unsigned gridr = (unsigned)((R + BN_ROWS_PER_WG - 1) / BN_ROWS_PER_WG);
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(gridr), dim3(BN_THREADS), (size_t)rlanes * 2 * C * sizeof(float), st,
                       (const unsigned short*)dy, (const unsigned short*)x, R, C, scale_shift, scale_shift + C, mean, rstd,
                       dgamma_dbeta_ws);
    EPI_CHECK_LAUNCH();
    const long long nvec = R * (C >> 3);
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(stream_grid(nvec)), dim3(BN_THREADS), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, nvec, R, C, gamma, scale_shift, scale_shift + C, mean, rstd, dgamma_dbeta_ws,
                       (unsigned short*)dx);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
