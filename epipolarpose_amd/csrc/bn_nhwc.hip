// BatchNorm2d (+ residual add) (+ ReLU) on NHWC bf16 activations, training and inference, for gfx950.
//
// Replaces, for every BatchNorm of the reference network (lib/models/pose3d_resnet.py: bn1, layer*.bn{1,2,3},
// downsample.1, deconv_layers.{1,4,7}), the cuDNN/MIOpen batch-norm kernels (3 forward + 3 backward launches
// per layer) plus the separate ReLU, residual-add and threshold-backward element-wise kernels.  At batch 32 these
// HBM-bound passes -- not the convolutions -- dominate the step (profiles/), so they are fused:
//   forward : stats (1 read) -> finalize (per-channel, tiny) -> apply: y = relu(x*scale + shift [+ res])
//   backward: reduce (dbeta, dgamma: reads dy, x [, y]) -> apply: dx [and dres] in one pass
// x [R][C]: R = B*H*W rows, C channels contiguous (C % 8 == 0); 16 bytes (8 channels) per lane per access;
// per-channel reductions: registers -> LDS -> one fp32 atomic per channel per workgroup.
#include "common.h"

namespace epi {

constexpr int BN_THREADS = 256;

enum { BN_MASK_NONE = 0, BN_MASK_FROM_X = 1, BN_MASK_FROM_Y = 2 };

// Reduction kernels use a 2-D partition: blockIdx.x = 64-channel slab (8 lanes x 16 B = one 128-byte line per row),
// blockIdx.y = row block.  256 threads = 8 channel groups x 32 row lanes, four rows in flight per lane.  Each workgroup
// ends with 128 fp32 atomics (64 channels x 2 sums); the partition keeps the total number of atomics per tensor ~64 K
// whatever the shape (C = 64, R = 524 288 ... C = 2048, R = 2048).
constexpr int BN_SLAB = 64;                 // channels per slab
constexpr int BN_RLANES = BN_THREADS / 8;   // 32 row lanes

__device__ __forceinline__ void slab_reduce_and_add(const float (&s)[8], const float (&q)[8], int g, int rl, int slab, int C,
                                                    float* __restrict__ sums, float* red /* [32][128] */) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[rl * 128 + g * 8 + k] = s[k]; red[rl * 128 + 64 + g * 8 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        float t = 0.f;
#pragma unroll 8
        for (int l = 0; l < BN_RLANES; ++l) t += red[l * 128 + threadIdx.x];
        const int which = threadIdx.x >> 6, ch = slab * BN_SLAB + (threadIdx.x & 63);
        if (ch < C) atomicAdd(sums + which * C + ch, t);
    }
}

// Per-channel epilogue of the statistics: mean / rstd from the batch sums, running statistics (momentum, unbiased
// variance), fused affine scale/shift; clears the forward sums and this layer's backward accumulator.
// Inference (sums == nullptr): scale/shift from the running statistics.
struct BnFinalizeArgs {
    const float* gamma; const float* beta; float eps, momentum; float* running_mean; float* running_var; long long* num_batches;
    float* mean; float* rstd; float* scale; float* shift; float* bwd_sums;
};

__device__ __forceinline__ void bn_finalize_channel(int c, float* __restrict__ sums, float s1, float s2, long long R, int C,
                                                    const BnFinalizeArgs& f) {
    if (f.bwd_sums) { f.bwd_sums[c] = 0.f; f.bwd_sums[C + c] = 0.f; }     // accumulator of this layer's NEXT backward pass
    double m, var;
    if (sums) {                        // (s1, s2) = (sum x, sum x^2) of channel c, already read by the caller
        m = (double)s1 / (double)R;
        var = (double)s2 / (double)R - m * m;
        if (var < 0) var = 0;
        sums[c] = 0.f;                 // leave the accumulator clean for the next call (and graph replays)
        sums[C + c] = 0.f;
        if (f.running_mean) {
            const double unbiased = (R > 1) ? var * (double)R / (double)(R - 1) : var;
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)m;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unbiased;
        }
    } else {
        m = f.running_mean[c];
        var = f.running_var[c];
    }
    const float rs = (float)(1.0 / sqrt(var + (double)f.eps));
    if (f.mean) { f.mean[c] = (float)m; f.rstd[c] = rs; }
    const float sc = f.gamma[c] * rs;
    f.scale[c] = sc;
    f.shift[c] = f.beta[c] - (float)m * sc;
}

__global__ void bn_finalize_kernel(float* __restrict__ sums, long long R, int C, BnFinalizeArgs f) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && f.num_batches && sums) f.num_batches[0] += 1;
    if (c < C) bn_finalize_channel(c, sums, sums ? sums[c] : 0.f, sums ? sums[C + c] : 0.f, R, C, f);
}

// sums[0..C) += sum x, sums[C..2C) += sum x^2     (sums zero on entry).  FINALIZE: the last workgroup to arrive (agent-scope
// ticket after an agent-scope release of its atomics; acquire before it reads the sums -- HIP guide, Guideline 16) runs the
// per-channel epilogue, so the forward pass needs no separate finalize launch.  ticket word = sums[2C] (as uint), kept 0.
template <bool FINALIZE>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const unsigned short* __restrict__ x, long long R, int C,
                                                              int rows_per_wg, float* __restrict__ sums, BnFinalizeArgs fin) {
    __shared__ float red[BN_RLANES * 128];
    __shared__ int is_last;
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    const int ch0 = slab * BN_SLAB + g * 8;
    float s[8], q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    if (ch0 < C) {
        long long r = r0 + rl;
        for (; r + 3LL * BN_RLANES < r1; r += 4LL * BN_RLANES) {
            uint4v raw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) raw[u] = *reinterpret_cast<const uint4v*>(x + (r + (long long)u * BN_RLANES) * C + ch0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int w[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = __uint_as_float(w[i] << 16), hi = __uint_as_float(w[i] & 0xffff0000u);
                    s[2 * i] += lo; q[2 * i] = fmaf(lo, lo, q[2 * i]);
                    s[2 * i + 1] += hi; q[2 * i + 1] = fmaf(hi, hi, q[2 * i + 1]);
                }
            }
        }
        for (; r < r1; r += BN_RLANES) {
            float v[8];
            Elem<unsigned short>::load(x + r * C + ch0, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) { s[k] += v[k]; q[k] = fmaf(v[k], v[k], q[k]); }
        }
    }
    slab_reduce_and_add(s, q, g, rl, slab, C, sums, red);
    if (FINALIZE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every wave drains the atomics it issued
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned int* ticket = reinterpret_cast<unsigned int*>(sums + 2 * C);
            const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == gridDim.x * gridDim.y - 1u);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next call
                if (fin.num_batches) fin.num_batches[0] += 1;
            }
            is_last = last;
        }
        __syncthreads();
        if (is_last)
            for (int c = threadIdx.x; c < C; c += BN_THREADS) {
                // the sums were produced by L2 atomics of other workgroups: read them past this CU's L1
                const float s1 = __hip_atomic_load(sums + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float s2 = __hip_atomic_load(sums + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bn_finalize_channel(c, sums, s1, s2, R, C, fin);
            }
    }
}

// y = act(x * scale[c] + shift[c] (+ res))
template <bool RELU, bool RES>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ res,
                                                              long long nvec, int C, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, unsigned short* __restrict__ y) {
    const int cg = C >> 3;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * BN_THREADS) {
        const int g = (int)(i % cg);
        float v[8], r[8];
        Elem<unsigned short>::load(x + i * 8, v);
        if (RES) Elem<unsigned short>::load(res + i * 8, r);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = v[k] * scale[g * 8 + k] + shift[g * 8 + k];
            if (RES) t += r[k];
            v[k] = RELU ? fmaxf(t, 0.f) : t;
        }
        Elem<unsigned short>::store(y + i * 8, v);
    }
}

// sums[0..C) += sum dz, sums[C..2C) += sum dz * xhat,   dz = dy * mask,   xhat = (x - mean) * rstd   (sums zero on entry)
//   MASK_FROM_X: mask = (x*scale + shift > 0)     MASK_FROM_Y: mask = (y > 0)     MASK_NONE: 1
template <int MASK>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                   const unsigned short* __restrict__ y, long long R, int C,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   int rows_per_wg, float* __restrict__ sums) {
    __shared__ float red[BN_RLANES * 128];
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    const int ch0 = slab * BN_SLAB + g * 8;
    float s[8], q[8], sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; sc[k] = sh[k] = mu[k] = rs[k] = 0.f; }
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    if (ch0 < C) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale[ch0 + k]; sh[k] = shift[ch0 + k]; mu[k] = mean[ch0 + k]; rs[k] = rstd[ch0 + k]; }
        auto accum = [&](const float (&xv)[8], const float (&gv)[8], const float (&yv)[8]) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                bool on = true;
                if (MASK == BN_MASK_FROM_X) on = xv[k] * sc[k] + sh[k] > 0.f;
                if (MASK == BN_MASK_FROM_Y) on = yv[k] > 0.f;
                const float dz = on ? gv[k] : 0.f;
                s[k] += dz;
                q[k] = fmaf(dz, (xv[k] - mu[k]) * rs[k], q[k]);
            }
        };
        long long r = r0 + rl;
        for (; r + BN_RLANES < r1; r += 2LL * BN_RLANES) {        // two rows (4-6 independent 16-byte loads) in flight per lane
            float xa[8], ga[8], ya[8], xb[8], gb[8], yb[8];
            Elem<unsigned short>::load(x + r * C + ch0, xa);
            Elem<unsigned short>::load(dy + r * C + ch0, ga);
            Elem<unsigned short>::load(x + (r + BN_RLANES) * C + ch0, xb);
            Elem<unsigned short>::load(dy + (r + BN_RLANES) * C + ch0, gb);
            if (MASK == BN_MASK_FROM_Y) {
                Elem<unsigned short>::load(y + r * C + ch0, ya);
                Elem<unsigned short>::load(y + (r + BN_RLANES) * C + ch0, yb);
            }
            accum(xa, ga, ya);
            accum(xb, gb, yb);
        }
        for (; r < r1; r += BN_RLANES) {
            float xv[8], gv[8], yv[8];
            Elem<unsigned short>::load(x + r * C + ch0, xv);
            Elem<unsigned short>::load(dy + r * C + ch0, gv);
            if (MASK == BN_MASK_FROM_Y) Elem<unsigned short>::load(y + r * C + ch0, yv);
            accum(xv, gv, yv);
        }
    }
    slab_reduce_and_add(s, q, g, rl, slab, C, sums, red);
}

// dx = gamma*rstd * (dz - dbeta/R - xhat * dgamma/R);  dres = dz (residual branch gradient) when requested
template <int MASK, bool DRES>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                                  const unsigned short* __restrict__ y, long long nvec, long long R, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ sums,
                                                                  unsigned short* __restrict__ dx, unsigned short* __restrict__ dres) {
    const int cg = C >> 3;
    const float inv_r = 1.f / (float)R;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * BN_THREADS) {
        const int g = (int)(i % cg);
        float xv[8], gv[8], yv[8], o[8], z[8];
        Elem<unsigned short>::load(x + i * 8, xv);
        Elem<unsigned short>::load(dy + i * 8, gv);
        if (MASK == BN_MASK_FROM_Y) Elem<unsigned short>::load(y + i * 8, yv);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = g * 8 + k;
            bool on = true;
            if (MASK == BN_MASK_FROM_X) on = xv[k] * scale[c] + shift[c] > 0.f;
            if (MASK == BN_MASK_FROM_Y) on = yv[k] > 0.f;
            const float dz = on ? gv[k] : 0.f;
            const float xh = (xv[k] - mean[c]) * rstd[c];
            z[k] = dz;
            o[k] = gamma[c] * rstd[c] * (dz - sums[c] * inv_r - xh * sums[C + c] * inv_r);
        }
        Elem<unsigned short>::store(dx + i * 8, o);
        if (DRES) Elem<unsigned short>::store(dres + i * 8, z);
    }
}

static inline bool bn_shape_ok(long long R, int C) { return R > 0 && C > 0 && C % 8 == 0 && R < (1LL << 40); }
static inline unsigned stream_grid(long long nvec) {
    const long long want = (nvec + BN_THREADS - 1) / BN_THREADS;
    return (unsigned)(want < 8192 ? want : 8192);
}
// 2-D blocking of the reduction kernels: nslab x nrb workgroups, ~512-1024 in total, >= 128 rows each
static inline void reduce_blocking(long long R, int C, int* rows_per_wg, dim3* grid) {
    const int nslab = (C + BN_SLAB - 1) / BN_SLAB;
    long long nrb = (nslab >= 2 ? 1024 : 512) / nslab;
    const long long max_rb = (R + 127) / 128;
    if (nrb > max_rb) nrb = max_rb;
    if (nrb < 1) nrb = 1;
    long long rpw = (R + nrb - 1) / nrb;
    rpw = (rpw + BN_RLANES - 1) / BN_RLANES * BN_RLANES;
    nrb = (R + rpw - 1) / rpw;
    *rows_per_wg = (int)rpw;
    *grid = dim3((unsigned)nslab, (unsigned)nrb);
}

}  // namespace epi

using namespace epi;

extern "C" int epi_bn_act_fwd(const void* x, const void* residual, long long R, int C, const float* gamma, const float* beta,
                              float eps, float momentum, int training, int relu, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float* mean, float* rstd, float* scale_shift, float* sums_ws,
                              float* bwd_sums, void* y, epi_stream_t stream) {
    if (!x || !gamma || !beta || !scale_shift || !y) return EPI_ERR_INVALID_ARGUMENT;
    if (training && (!mean || !rstd || !sums_ws)) return EPI_ERR_INVALID_ARGUMENT;
    if (!training && (!running_mean || !running_var)) return EPI_ERR_INVALID_ARGUMENT;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EPI_ERR_INVALID_ARGUMENT;
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    BnFinalizeArgs fin;
    fin.gamma = gamma; fin.beta = beta; fin.eps = eps; fin.momentum = momentum; fin.running_mean = running_mean;
    fin.running_var = running_var; fin.num_batches = num_batches_tracked; fin.mean = mean; fin.rstd = rstd;
    fin.scale = scale_shift; fin.shift = scale_shift + C; fin.bwd_sums = training ? bwd_sums : nullptr;
    if (training) {       // statistics + per-channel epilogue in one launch (the last workgroup finalizes)
        int rpw = 0;
        dim3 rgrid;
        reduce_blocking(R, C, &rpw, &rgrid);
        hipLaunchKernelGGL((bn_stats_kernel<true>), rgrid, dim3(BN_THREADS), 0, st, (const unsigned short*)x, R, C, rpw, sums_ws, fin);
        EPI_CHECK_LAUNCH();
    } else {
        hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, st, nullptr, R, C, fin);
        EPI_CHECK_LAUNCH();
    }
    const long long nvec = R * (C >> 3);
    const dim3 grid(stream_grid(nvec)), block(BN_THREADS);
    const unsigned short* xs = (const unsigned short*)x;
    const unsigned short* rs = (const unsigned short*)residual;
    unsigned short* ys = (unsigned short*)y;
    if (relu && residual) hipLaunchKernelGGL((bn_apply_kernel<true, true>), grid, block, 0, st, xs, rs, nvec, C, scale_shift, scale_shift + C, ys);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<true, false>), grid, block, 0, st, xs, rs, nvec, C, scale_shift, scale_shift + C, ys);
    else if (residual) hipLaunchKernelGGL((bn_apply_kernel<false, true>), grid, block, 0, st, xs, rs, nvec, C, scale_shift, scale_shift + C, ys);
    else hipLaunchKernelGGL((bn_apply_kernel<false, false>), grid, block, 0, st, xs, rs, nvec, C, scale_shift, scale_shift + C, ys);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_bn_act_bwd(const void* dy, const void* x, const void* y, long long R, int C, const float* gamma, const float* mean,
                              const float* rstd, const float* scale_shift, int relu, float* dbeta_dgamma, void* dx, void* dres,
                              epi_stream_t stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !scale_shift || !dbeta_dgamma || !dx) return EPI_ERR_INVALID_ARGUMENT;
    if (dres && relu && !y) return EPI_ERR_INVALID_ARGUMENT;        // residual + ReLU: the mask comes from the saved output
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rpw = 0;
    dim3 rgrid;
    reduce_blocking(R, C, &rpw, &rgrid);
    const int mask = !relu ? BN_MASK_NONE : (y ? BN_MASK_FROM_Y : BN_MASK_FROM_X);
    const unsigned short *dys = (const unsigned short*)dy, *xs = (const unsigned short*)x, *ys = (const unsigned short*)y;
    const float *sc = scale_shift, *sh = scale_shift + C;
#define EPI_BN_RED(M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<M>), rgrid, dim3(BN_THREADS), 0, st, dys, xs, ys, R, C, sc, sh, mean, rstd, rpw, dbeta_dgamma)
    if (mask == BN_MASK_NONE) EPI_BN_RED(BN_MASK_NONE);
    else if (mask == BN_MASK_FROM_X) EPI_BN_RED(BN_MASK_FROM_X);
    else EPI_BN_RED(BN_MASK_FROM_Y);
#undef EPI_BN_RED
    EPI_CHECK_LAUNCH();
    const long long nvec = R * (C >> 3);
    const dim3 grid(stream_grid(nvec)), block(BN_THREADS);
    unsigned short *dxs = (unsigned short*)dx, *drs = (unsigned short*)dres;
#define EPI_BN_APP(M, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<M, D>), grid, block, 0, st, dys, xs, ys, nvec, R, C, gamma, sc, sh, mean, rstd, dbeta_dgamma, dxs, drs)
    if (mask == BN_MASK_NONE) { if (dres) EPI_BN_APP(BN_MASK_NONE, true); else EPI_BN_APP(BN_MASK_NONE, false); }
    else if (mask == BN_MASK_FROM_X) { if (dres) EPI_BN_APP(BN_MASK_FROM_X, true); else EPI_BN_APP(BN_MASK_FROM_X, false); }
    else { if (dres) EPI_BN_APP(BN_MASK_FROM_Y, true); else EPI_BN_APP(BN_MASK_FROM_Y, false); }
#undef EPI_BN_APP
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// Per-column sum and sum of squares of a bf16 matrix x [R][C], ACCUMULATED into sums [2C] f32 (caller zeroes it)
// (bias gradient of the final conv: db = sum over rows of dlogits).
extern "C" int epi_column_sums_bf16(const void* x, long long R, int C, float* sums, epi_stream_t stream) {
    if (!x || !sums) return EPI_ERR_INVALID_ARGUMENT;
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rpw = 0;
    dim3 rgrid;
    reduce_blocking(R, C, &rpw, &rgrid);
    BnFinalizeArgs none = {};
    hipLaunchKernelGGL((bn_stats_kernel<false>), rgrid, dim3(BN_THREADS), 0, st, (const unsigned short*)x, R, C, rpw, sums, none);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
