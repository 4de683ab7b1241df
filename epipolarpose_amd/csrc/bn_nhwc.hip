// BatchNorm2d (+ residual add) (+ ReLU) on NHWC bf16 activations, training and inference, for gfx950.
//
// Replaces, for every BatchNorm of the reference network (lib/models/pose3d_resnet.py: bn1, layer*.bn{1,2,3},
// downsample.1, deconv_layers.{1,4,7}), the cuDNN/MIOpen batch-norm kernels (3 forward + 3 backward launches
// per layer) plus the separate ReLU, residual-add and threshold-backward element-wise kernels.  At batch 32 these
// HBM-bound passes -- not the convolutions -- dominate the step (profiles/), so they are fused:
//   forward : stats (1 read) -> apply: y = relu(x*scale + shift [+ res]); every apply workgroup derives scale/shift for
//             all channels from the batch sums into LDS (a few hundred flops), workgroup 0 also writes the saved
//             statistics, updates the running estimates and clears the NEXT backward's accumulator -- no separate
//             per-channel "finalize" launch (56 launches / step at ~4 us each)
//   backward: reduce (dbeta, dgamma: reads dy, x [, y]) -> apply: dx [and dres] in one pass; workgroup 0 clears the
//             forward accumulator for the next step
// x [R][C]: R = B*H*W rows, C channels contiguous (C % 8 == 0); 16 bytes (8 channels) per lane per access;
// per-channel reductions: registers -> LDS -> one fp32 atomic per channel per workgroup.
#include "common.h"
#include "bn_affine.h"
#include <cstdlib>

namespace epi {

constexpr int BN_THREADS = 256;

enum { BN_MASK_NONE = 0, BN_MASK_FROM_X = 1, BN_MASK_FROM_Y = 2 };

// Reduction kernels use a 2-D partition: blockIdx.x = 64-channel slab (8 lanes x 16 B = one 128-byte line per row),
// blockIdx.y = row block.  256 threads = 8 channel groups x 32 row lanes, four rows in flight per lane.  Each workgroup
// ends with 128 fp32 atomics (64 channels x 2 sums); the partition keeps the total number of atomics per tensor ~64 K
// whatever the shape (C = 64, R = 524 288 ... C = 2048, R = 2048).
constexpr int BN_SLAB = 64;                 // channels per slab (bf16 storage; 32 with the fp32 storage of the fp32-grade mode)
constexpr int BN_RLANES = BN_THREADS / 8;   // 32 row lanes

// Storage type T: unsigned short = bf16 (the training path) or float (the fp32-grade verification mode, models/precise.py: the same
// kernels, 16 bytes = 4 channels per lane instead of 8).  V = channels per lane, SLAB = 8 V channels per workgroup column.
// det_part (deterministic mode, csrc/capi.hip): [row blocks][2C] partial sums, one plain store per channel and workgroup instead of the atomic;
// ordered_sum_kernel then adds the row blocks in index order
template <int V>
__device__ __forceinline__ void slab_reduce_and_add(const float (&s)[V], const float (&q)[V], int g, int rl, int slab, int C,
                                                    float* __restrict__ sums, float* red /* [32][16 V] */, float* __restrict__ det_part = nullptr) {
    constexpr int SLAB = 8 * V;
#pragma unroll
    for (int k = 0; k < V; ++k) { red[rl * (2 * SLAB) + g * V + k] = s[k]; red[rl * (2 * SLAB) + SLAB + g * V + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < 2 * SLAB) {
        float t = 0.f;
#pragma unroll 8
        for (int l = 0; l < BN_RLANES; ++l) t += red[l * (2 * SLAB) + threadIdx.x];
        const int which = threadIdx.x / SLAB, ch = slab * SLAB + (threadIdx.x % SLAB);
        if (ch < C) {
            if (det_part) det_part[(long long)blockIdx.y * 2 * C + which * C + ch] = t;
            else atomicAdd(sums + which * C + ch, t);
        }
    }
}

// dst[i] += part[0][i] + part[1][i] + ... (index order), i < n
__global__ __launch_bounds__(256) void ordered_sum_kernel(const float* __restrict__ part, int nblocks, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double t = 0.0;                         // (up to 1024 partial sums per channel: float64 keeps the ordered sum at least as accurate as the atomics)
    for (int k = 0; k < nblocks; ++k) t += (double)part[(long long)k * n + i];
    dst[i] += (float)t;
}

// sums[0..C) += sum x, sums[C..2C) += sum x^2     (sums zero on entry)
// (ncopies accumulator copies [ncopies][2C], row block b adds into copy b % ncopies: fewer atomics per 128-byte line; 1 = plain [2C])
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(const T* __restrict__ x, long long R, int C,
                                                              int rows_per_wg, float* __restrict__ sums, int ncopies, float* __restrict__ det_part) {
    constexpr int V = Elem<T>::VEC, SLAB = 8 * V;
    __shared__ float red[BN_RLANES * 2 * SLAB];
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    const int ch0 = slab * SLAB + g * V;
    float s[V], q[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s[k] = 0.f; q[k] = 0.f; }
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    if (ch0 < C) {
        long long r = r0 + rl;
        for (; r + 3LL * BN_RLANES < r1; r += 4LL * BN_RLANES) {        // four rows (independent 16-byte loads) in flight per lane
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) Elem<T>::load(x + (r + (long long)u * BN_RLANES) * C + ch0, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < V; ++k) { s[k] += v[u][k]; q[k] = fmaf(v[u][k], v[u][k], q[k]); }
        }
        for (; r < r1; r += BN_RLANES) {
            float v[V];
            Elem<T>::load(x + r * C + ch0, v);
#pragma unroll
            for (int k = 0; k < V; ++k) { s[k] += v[k]; q[k] = fmaf(v[k], v[k], q[k]); }
        }
    }
    slab_reduce_and_add<V>(s, q, g, rl, slab, C, sums + (long long)(blockIdx.y % ncopies) * 2 * C, red, det_part);
}


// y = act(x * scale[c] + shift[c] (+ res))
template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                              long long nvec, int C, BnAffine a, T* __restrict__ y) {
    constexpr int V = Elem<T>::VEC;
    extern __shared__ float ss[];                 // [C] scale | [C] shift
    const bool lead = blockIdx.x == 0;
    const double inv_r = a.inv_r;
    for (int c = threadIdx.x; c < C; c += BN_THREADS) {
        // every workgroup repeats this for all C channels: keep it to one float64 multiply + fma (E[x^2] - m^2 cancels in
        // float32) and float32 for the rest -- no float64 divide / sqrt
        double m, var;
        if (a.sums) {
            float s1 = a.sums[c], s2 = a.sums[C + c];
            for (int k = 1; k < a.ncopies; ++k) { s1 += a.sums[2 * k * C + c]; s2 += a.sums[(2 * k + 1) * C + c]; }
            m = (double)s1 * inv_r;
            var = fma(-m, m, (double)s2 * inv_r);
            if (var < 0) var = 0;
        } else {
            m = a.running_mean[c];
            var = a.running_var[c];
        }
        const float rs = 1.0f / sqrtf((float)var + a.eps);
        const float sc = a.gamma[c] * rs, sh = a.beta[c] - (float)m * sc;
        ss[c] = sc;
        ss[C + c] = sh;
        if (lead) {
            if (a.mean) { a.mean[c] = (float)m; a.rstd[c] = rs; }
            a.scale[c] = sc;
            a.shift[c] = sh;
            if (a.sums && a.running_mean) {
                const double unbiased = (a.R > 1) ? var * (double)a.R / (double)(a.R - 1) : var;
                a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
                a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
            }
            if (a.bwd_sums) { a.bwd_sums[c] = 0.f; a.bwd_sums[C + c] = 0.f; }
        }
    }
    if (lead && threadIdx.x == 0 && a.num_batches && a.sums) a.num_batches[0] += 1;
    __syncthreads();
    const int cg = C / V;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * BN_THREADS) {
        const int g = (int)(i % cg);
        float v[V], r[V], sc[V], sh[V];
        Elem<T>::load(x + i * V, v);
        if (RES) Elem<T>::load(res + i * V, r);
#pragma unroll
        for (int k = 0; k < V; k += 4) {
            const float4v s4 = *reinterpret_cast<const float4v*>(ss + g * V + k), h4 = *reinterpret_cast<const float4v*>(ss + C + g * V + k);
            sc[k] = s4.x; sc[k + 1] = s4.y; sc[k + 2] = s4.z; sc[k + 3] = s4.w;
            sh[k] = h4.x; sh[k + 1] = h4.y; sh[k + 2] = h4.z; sh[k + 3] = h4.w;
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float t = v[k] * sc[k] + sh[k];
            if (RES) t += r[k];
            v[k] = RELU ? fmaxf(t, 0.f) : t;
        }
        Elem<T>::store(y + i * V, v);
    }
}

// sums[0..C) += sum dz, sums[C..2C) += sum dz * xhat,   dz = dy * mask,   xhat = (x - mean) * rstd   (sums zero on entry)
//   MASK_FROM_X: mask = (x*scale + shift > 0)     MASK_FROM_Y: mask = (y > 0)     MASK_NONE: 1
template <typename T, int MASK>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                   const T* __restrict__ y, long long R, int C,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                   int rows_per_wg, float* __restrict__ sums, float* __restrict__ det_part) {
    constexpr int V = Elem<T>::VEC, SLAB = 8 * V;
    __shared__ float red[BN_RLANES * 2 * SLAB];
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    const int ch0 = slab * SLAB + g * V;
    float s[V], q[V], sc[V], sh[V], mu[V], rs[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { s[k] = 0.f; q[k] = 0.f; sc[k] = sh[k] = mu[k] = rs[k] = 0.f; }
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    if (ch0 < C) {
#pragma unroll
        for (int k = 0; k < V; ++k) { sc[k] = scale[ch0 + k]; sh[k] = shift[ch0 + k]; mu[k] = mean[ch0 + k]; rs[k] = rstd[ch0 + k]; }
        auto accum = [&](const float (&xv)[V], const float (&gv)[V], const float (&yv)[V]) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
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
            float xa[V], ga[V], ya[V], xb[V], gb[V], yb[V];
            Elem<T>::load(x + r * C + ch0, xa);
            Elem<T>::load(dy + r * C + ch0, ga);
            Elem<T>::load(x + (r + BN_RLANES) * C + ch0, xb);
            Elem<T>::load(dy + (r + BN_RLANES) * C + ch0, gb);
            if (MASK == BN_MASK_FROM_Y) {
                Elem<T>::load(y + r * C + ch0, ya);
                Elem<T>::load(y + (r + BN_RLANES) * C + ch0, yb);
            }
            accum(xa, ga, ya);
            accum(xb, gb, yb);
        }
        for (; r < r1; r += BN_RLANES) {
            float xv[V], gv[V], yv[V];
            Elem<T>::load(x + r * C + ch0, xv);
            Elem<T>::load(dy + r * C + ch0, gv);
            if (MASK == BN_MASK_FROM_Y) Elem<T>::load(y + r * C + ch0, yv);
            accum(xv, gv, yv);
        }
    }
    slab_reduce_and_add<V>(s, q, g, rl, slab, C, sums, red, det_part);
}

// dx = gamma*rstd * (dz - dbeta/R - xhat * dgamma/R);  dres = dz (residual branch gradient) when requested
template <typename T, int MASK, bool DRES>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                  const T* __restrict__ y, long long nvec, long long R, int C,
                                                                  const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ sums,
                                                                  T* __restrict__ dx, T* __restrict__ dres,
                                                                  float* __restrict__ fwd_sums_clear, int fwd_sums_copies,
                                                                  float* __restrict__ param_grads) {
    if (blockIdx.x == 0) {
        if (fwd_sums_clear)                        // the forward statistics accumulator of this layer's NEXT forward pass
            for (int c = threadIdx.x; c < 2 * C * fwd_sums_copies; c += BN_THREADS) fwd_sums_clear[c] = 0.f;
        if (param_grads)                           // (dbeta | dgamma) handed to the caller in memory the accumulator protocol never touches
            for (int c = threadIdx.x; c < 2 * C; c += BN_THREADS) param_grads[c] = sums[c];
    }
    constexpr int V = Elem<T>::VEC;
    const int cg = C / V;
    const float inv_r = 1.f / (float)R;
    for (long long i = (long long)blockIdx.x * BN_THREADS + threadIdx.x; i < nvec; i += (long long)gridDim.x * BN_THREADS) {
        const int g = (int)(i % cg);
        float xv[V], gv[V], yv[V], o[V], z[V];
        Elem<T>::load(x + i * V, xv);
        Elem<T>::load(dy + i * V, gv);
        if (MASK == BN_MASK_FROM_Y) Elem<T>::load(y + i * V, yv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const int c = g * V + k;
            bool on = true;
            if (MASK == BN_MASK_FROM_X) on = xv[k] * scale[c] + shift[c] > 0.f;
            if (MASK == BN_MASK_FROM_Y) on = yv[k] > 0.f;
            const float dz = on ? gv[k] : 0.f;
            const float xh = (xv[k] - mean[c]) * rstd[c];
            z[k] = dz;
            o[k] = gamma[c] * rstd[c] * (dz - sums[c] * inv_r - xh * sums[C + c] * inv_r);
        }
        Elem<T>::store(dx + i * V, o);
        if (DRES) Elem<T>::store(dres + i * V, z);
    }
}

// ---- 2-D partitioned apply kernels (round 3) ------------------------------------------------------------------------------------
// The kernels above walk the tensor with a grid-stride loop: every workgroup first derives the affine of ALL C channels into LDS (a few
// float64 operations per channel and workgroup) and the backward one re-loads seven per-channel arrays for every 16-byte vector it
// touches.  On the large early tensors that is noise; on the ~110 small launches of a ResNet-50 step (<= 8 MB tensors, 4 .. 9 us each, where
// the payload itself is 1 .. 2 us of HBM time) it is most of the kernel.  These variants use the partition of the reduction kernels instead:
// blockIdx.x = a slab of 8 V channels, blockIdx.y = a row block, a thread owns V fixed channels and walks rows -- its coefficients live in
// registers for the whole kernel, the per-workgroup prologue is ONE channel per thread of the first 8 V threads.
template <typename T, bool RELU, bool RES>
__global__ __launch_bounds__(BN_THREADS) void bn_apply2d_kernel(const T* __restrict__ x, const T* __restrict__ res, long long R, int C, int rows_per_wg,
                                                                BnAffine a, T* __restrict__ y) {
    constexpr int V = Elem<T>::VEC, SLAB = 8 * V;
    __shared__ float ss[2 * SLAB];                // scale | shift of this slab
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    const bool lead = blockIdx.y == 0;
    // (round 5) the first four rows of every lane are REQUESTED before the per-channel prologue: the prologue is a dependent chain of its own (batch sums
    // from L2 -> a few float64 operations -> LDS -> barrier, ~1 us) and the ~110 small BatchNorm launches of a step are latency, not bandwidth
    const int ch0 = slab * SLAB + g * V;
    const bool act = ch0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    long long r = r0 + rl;
    const bool pre = act && r + 3LL * BN_RLANES < r1;
    float pv[4][V], pq[4][V];
    if (pre) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Elem<T>::load(x + (r + (long long)u * BN_RLANES) * C + ch0, pv[u]);
            if (RES) Elem<T>::load(res + (r + (long long)u * BN_RLANES) * C + ch0, pq[u]);
        }
    }
    if (threadIdx.x < SLAB) {
        const int c = slab * SLAB + threadIdx.x;
        float sc = 0.f, sh = 0.f;
        if (c < C) {
            double m, var;
            if (a.sums) {
                float s1 = a.sums[c], s2 = a.sums[C + c];
                for (int k = 1; k < a.ncopies; ++k) { s1 += a.sums[2 * k * C + c]; s2 += a.sums[(2 * k + 1) * C + c]; }
                m = (double)s1 * a.inv_r;
                var = fma(-m, m, (double)s2 * a.inv_r);
                if (var < 0) var = 0;
            } else {
                m = a.running_mean[c];
                var = a.running_var[c];
            }
            const float rs = 1.0f / sqrtf((float)var + a.eps);
            sc = a.gamma[c] * rs;
            sh = a.beta[c] - (float)m * sc;
            if (lead) {
                if (a.mean) { a.mean[c] = (float)m; a.rstd[c] = rs; }
                a.scale[c] = sc;
                a.shift[c] = sh;
                if (a.sums && a.running_mean) {
                    const double unbiased = (a.R > 1) ? var * (double)a.R / (double)(a.R - 1) : var;
                    a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
                    a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
                }
                if (a.bwd_sums) { a.bwd_sums[c] = 0.f; a.bwd_sums[C + c] = 0.f; }
            }
        }
        ss[threadIdx.x] = sc;
        ss[SLAB + threadIdx.x] = sh;
    }
    if (lead && slab == 0 && threadIdx.x == 0 && a.num_batches && a.sums) a.num_batches[0] += 1;
    __syncthreads();
    if (!act) return;
    float sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = ss[g * V + k]; sh[k] = ss[SLAB + g * V + k]; }
    auto one = [&](float (&v)[V], const float (&r)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float t = v[k] * sc[k] + sh[k];
            if (RES) t += r[k];
            v[k] = RELU ? fmaxf(t, 0.f) : t;
        }
    };
    if (pre) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            one(pv[u], pq[u]);
            Elem<T>::store(y + (r + (long long)u * BN_RLANES) * C + ch0, pv[u]);
        }
        r += 4LL * BN_RLANES;
    }
    for (; r + 3LL * BN_RLANES < r1; r += 4LL * BN_RLANES) {             // four rows (4 .. 8 independent 16-byte loads) in flight per lane
        float v[4][V], q[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Elem<T>::load(x + (r + (long long)u * BN_RLANES) * C + ch0, v[u]);
            if (RES) Elem<T>::load(res + (r + (long long)u * BN_RLANES) * C + ch0, q[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            one(v[u], q[u]);
            Elem<T>::store(y + (r + (long long)u * BN_RLANES) * C + ch0, v[u]);
        }
    }
    for (; r < r1; r += BN_RLANES) {
        float v[V], q[V];
        Elem<T>::load(x + r * C + ch0, v);
        if (RES) Elem<T>::load(res + r * C + ch0, q);
        one(v, q);
        Elem<T>::store(y + r * C + ch0, v);
    }
}

template <typename T, int MASK, bool DRES>
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply2d_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y, long long R, int C,
                                                                    int rows_per_wg, const float* __restrict__ gamma, const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, const float* __restrict__ sums,
                                                                    T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ fwd_sums_clear,
                                                                    int fwd_sums_copies, float* __restrict__ param_grads) {
    constexpr int V = Elem<T>::VEC, SLAB = 8 * V;
    __shared__ float cs[5 * SLAB];                // P | Q | S | scale | shift of this slab:  dx = P dz + Q x + S
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    // (round 5) the first two rows of every lane are requested before the per-channel prologue (see bn_apply2d_kernel)
    const int ch0 = slab * SLAB + g * V;
    const bool act = ch0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    long long r = r0 + rl;
    const bool pre = act && r + BN_RLANES < r1;
    float pxa[V], pga[V], pya[V], pxb[V], pgb[V], pyb[V];
    if (pre) {
        Elem<T>::load(x + r * C + ch0, pxa);
        Elem<T>::load(dy + r * C + ch0, pga);
        Elem<T>::load(x + (r + BN_RLANES) * C + ch0, pxb);
        Elem<T>::load(dy + (r + BN_RLANES) * C + ch0, pgb);
        if (MASK == BN_MASK_FROM_Y) {
            Elem<T>::load(y + r * C + ch0, pya);
            Elem<T>::load(y + (r + BN_RLANES) * C + ch0, pyb);
        }
    }
    if (threadIdx.x < SLAB) {
        const int c = slab * SLAB + threadIdx.x;
        float P = 0.f, Q = 0.f, S = 0.f, sc = 0.f, sh = 0.f;
        if (c < C) {
            // dx = gamma rstd (dz - s1/R - xhat s2/R),  xhat = (x - mean) rstd
            const float inv_r = 1.f / (float)R, rs = rstd[c], gr = gamma[c] * rs;
            const float a = sums[c] * inv_r, b = sums[C + c] * inv_r * rs;
            P = gr;
            Q = -gr * b;
            S = gr * (b * mean[c] - a);
            sc = scale[c];
            sh = shift[c];
        }
        cs[threadIdx.x] = P; cs[SLAB + threadIdx.x] = Q; cs[2 * SLAB + threadIdx.x] = S; cs[3 * SLAB + threadIdx.x] = sc; cs[4 * SLAB + threadIdx.x] = sh;
    }
    if (blockIdx.x == 0 && blockIdx.y == 0) {       // (`sums` is only read by this launch; these are other buffers)
        if (fwd_sums_clear)
            for (int c = threadIdx.x; c < 2 * C * fwd_sums_copies; c += BN_THREADS) fwd_sums_clear[c] = 0.f;
        if (param_grads)
            for (int c = threadIdx.x; c < 2 * C; c += BN_THREADS) param_grads[c] = sums[c];
    }
    __syncthreads();
    if (!act) return;
    float P[V], Q[V], S[V], sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        P[k] = cs[g * V + k]; Q[k] = cs[SLAB + g * V + k]; S[k] = cs[2 * SLAB + g * V + k];
        sc[k] = MASK == BN_MASK_FROM_X ? cs[3 * SLAB + g * V + k] : 0.f;
        sh[k] = MASK == BN_MASK_FROM_X ? cs[4 * SLAB + g * V + k] : 0.f;
    }
    // in place: xv becomes dx, gv becomes dz
    auto one = [&](float (&xv)[V], float (&gv)[V], const float (&yv)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            bool on = true;
            if (MASK == BN_MASK_FROM_X) on = xv[k] * sc[k] + sh[k] > 0.f;
            if (MASK == BN_MASK_FROM_Y) on = yv[k] > 0.f;
            const float dz = on ? gv[k] : 0.f;
            xv[k] = fmaf(P[k], dz, fmaf(Q[k], xv[k], S[k]));
            gv[k] = dz;
        }
    };
    if (pre) {
        one(pxa, pga, pya);
        one(pxb, pgb, pyb);
        Elem<T>::store(dx + r * C + ch0, pxa);
        Elem<T>::store(dx + (r + BN_RLANES) * C + ch0, pxb);
        if (DRES) {
            Elem<T>::store(dres + r * C + ch0, pga);
            Elem<T>::store(dres + (r + BN_RLANES) * C + ch0, pgb);
        }
        r += 2LL * BN_RLANES;
    }
    for (; r + BN_RLANES < r1; r += 2LL * BN_RLANES) {                   // two rows (4 .. 6 independent 16-byte loads) in flight per lane
        float xa[V], ga[V], ya[V], xb[V], gb[V], yb[V];
        Elem<T>::load(x + r * C + ch0, xa);
        Elem<T>::load(dy + r * C + ch0, ga);
        Elem<T>::load(x + (r + BN_RLANES) * C + ch0, xb);
        Elem<T>::load(dy + (r + BN_RLANES) * C + ch0, gb);
        if (MASK == BN_MASK_FROM_Y) {
            Elem<T>::load(y + r * C + ch0, ya);
            Elem<T>::load(y + (r + BN_RLANES) * C + ch0, yb);
        }
        one(xa, ga, ya);
        one(xb, gb, yb);
        Elem<T>::store(dx + r * C + ch0, xa);
        Elem<T>::store(dx + (r + BN_RLANES) * C + ch0, xb);
        if (DRES) {
            Elem<T>::store(dres + r * C + ch0, ga);
            Elem<T>::store(dres + (r + BN_RLANES) * C + ch0, gb);
        }
    }
    for (; r < r1; r += BN_RLANES) {
        float xv[V], gv[V], yv[V];
        Elem<T>::load(x + r * C + ch0, xv);
        Elem<T>::load(dy + r * C + ch0, gv);
        if (MASK == BN_MASK_FROM_Y) Elem<T>::load(y + r * C + ch0, yv);
        one(xv, gv, yv);
        Elem<T>::store(dx + r * C + ch0, xv);
        if (DRES) Elem<T>::store(dres + r * C + ch0, gv);
    }
}

static inline bool bn_shape_ok(long long R, int C) { return R > 0 && C > 0 && C % 8 == 0 && R < (1LL << 40); }
static inline unsigned stream_grid(long long nvec) {
    const long long want = (nvec + BN_THREADS - 1) / BN_THREADS;
    return (unsigned)(want < 8192 ? want : 8192);
}
// 2-D blocking of the reduction kernels: nslab x nrb workgroups, ~512-1024 in total, >= 128 rows each
static inline void reduce_blocking(long long R, int C, int* rows_per_wg, dim3* grid, int slab_width = BN_SLAB) {
    const int nslab = (C + slab_width - 1) / slab_width;
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

// blocking of the 2-D apply kernels: enough workgroups to fill the chip several times over on the large tensors (they are HBM-bound and
// run 8 workgroups per CU), >= 128 rows (four per lane) each
static inline void apply_blocking(long long R, int C, int* rows_per_wg, dim3* grid, int slab_width) {
    const int nslab = (C + slab_width - 1) / slab_width;
    const long long target = 2048;
    long long nrb = target / nslab;
    const long long max_rb = (R + 127) / 128;
    if (nrb > max_rb) nrb = max_rb;
    if (nrb < 1) nrb = 1;
    long long rpw = (R + nrb - 1) / nrb;
    rpw = (rpw + BN_RLANES - 1) / BN_RLANES * BN_RLANES;
    nrb = (R + rpw - 1) / rpw;
    *rows_per_wg = (int)rpw;
    *grid = dim3((unsigned)nslab, (unsigned)nrb);
}
// EPI_BN_2D: bit 0 forward apply, bit 1 backward apply on the 2-D kernels (default 3 = both; 0 = the grid-stride kernels)
static int g_bn_2d = -1;
static inline int bn_2d_mode() {
    if (g_bn_2d < 0) g_bn_2d = 3;
    return g_bn_2d;
}

}  // namespace epi

using namespace epi;

// One statistics pass: sums_ws [ncopies][2C] += per-channel (sum, sum of squares) of x.  Deterministic mode: per-row-block partials into the
// library's scratch, then added in index order into copy 0.
template <typename T>
static int launch_stats(const T* x, long long R, int C, float* sums, int ncopies, hipStream_t st, bool column_sums = false) {
    int rpw = 0;
    dim3 rgrid;
    reduce_blocking(R, C, &rpw, &rgrid, 8 * Elem<T>::VEC);
    float* part = nullptr;
    if (deterministic()) {
        part = det_scratch((size_t)rgrid.y * 2 * C, st);
        if (!part) return EPI_ERR_WORKSPACE;
    }
    hipLaunchKernelGGL(bn_stats_kernel<T>, rgrid, dim3(BN_THREADS), 0, st, x, R, C, rpw, sums, ncopies, part);
    if (part) hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)((2 * C + 255) / 256)), dim3(256), 0, st, part, (int)rgrid.y, 2 * C, sums);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// The forward statistics accumulator sums_ws is [epi_bn_sum_copies(C)][2C]: producers (the statistics kernel's row blocks, a
// convolution epilogue's M tiles) spread their atomics over the copies; the apply kernel adds the copies up.  One copy for wide
// layers (few rows, so few producers -- and every apply workgroup reads all of them).
extern "C" int epi_bn_sum_copies(int C) { return C <= 512 ? 4 : 1; }

// y = relu(x * scale[c] + shift[c] + xd * scale_d[c] + shift_d[c]): the LAST BatchNorm of a residual unit whose shortcut is a projection
// (`residual = self.downsample(x)` = conv -> BatchNorm without activation, pose3d_resnet.py:79-86) and that projection's BatchNorm in ONE pass over
// the two raw convolution outputs: the projection's normalised output -- a C-wide tensor written once and read once -- is never materialised.
// Partition and register-resident coefficients as bn_apply2d_kernel; both layers' saved statistics, running estimates and accumulator
// hand-overs are done here.
template <typename T>
__global__ __launch_bounds__(BN_THREADS) void bn_apply_dual_kernel(const T* __restrict__ x, const T* __restrict__ xd, long long R, int C, int rows_per_wg,
                                                                   BnAffine a, BnAffine d, T* __restrict__ y) {
    constexpr int V = Elem<T>::VEC, SLAB = 8 * V;
    __shared__ float ss[3 * SLAB];                // scale | scale of the projection | both shifts added
    const int g = threadIdx.x & 7, rl = threadIdx.x >> 3, slab = blockIdx.x;
    const bool lead = blockIdx.y == 0;
    // (round 5) the first two rows of every lane are requested before the per-channel prologue (see bn_apply2d_kernel)
    const int ch0 = slab * SLAB + g * V;
    const bool act = ch0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_wg;
    const long long r1 = (r0 + rows_per_wg < R) ? r0 + rows_per_wg : R;
    long long r = r0 + rl;
    const bool pre = act && r + BN_RLANES < r1;
    float pva[V], pda[V], pvb[V], pdb[V];
    if (pre) {
        Elem<T>::load(x + r * C + ch0, pva);
        Elem<T>::load(xd + r * C + ch0, pda);
        Elem<T>::load(x + (r + BN_RLANES) * C + ch0, pvb);
        Elem<T>::load(xd + (r + BN_RLANES) * C + ch0, pdb);
    }
    if (threadIdx.x < SLAB) {
        const int c = slab * SLAB + threadIdx.x;
        float sc = 0.f, sh = 0.f, scd = 0.f, shd = 0.f;
        if (c < C) {
            bn_derive(a, C, c, lead, sc, sh);
            bn_derive(d, C, c, lead, scd, shd);
        }
        ss[threadIdx.x] = sc;
        ss[SLAB + threadIdx.x] = scd;
        ss[2 * SLAB + threadIdx.x] = sh + shd;
    }
    if (lead && slab == 0 && threadIdx.x == 0) {
        if (a.num_batches && a.sums) a.num_batches[0] += 1;
        if (d.num_batches && d.sums) d.num_batches[0] += 1;
    }
    __syncthreads();
    if (!act) return;
    float sc[V], scd[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { sc[k] = ss[g * V + k]; scd[k] = ss[SLAB + g * V + k]; sh[k] = ss[2 * SLAB + g * V + k]; }
    if (pre) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            pva[k] = fmaxf(pva[k] * sc[k] + pda[k] * scd[k] + sh[k], 0.f);
            pvb[k] = fmaxf(pvb[k] * sc[k] + pdb[k] * scd[k] + sh[k], 0.f);
        }
        Elem<T>::store(y + r * C + ch0, pva);
        Elem<T>::store(y + (r + BN_RLANES) * C + ch0, pvb);
        r += 2LL * BN_RLANES;
    }
    for (; r + BN_RLANES < r1; r += 2LL * BN_RLANES) {            // two rows (four independent 16-byte loads) in flight per lane
        float va[V], da[V], vb[V], db[V];
        Elem<T>::load(x + r * C + ch0, va);
        Elem<T>::load(xd + r * C + ch0, da);
        Elem<T>::load(x + (r + BN_RLANES) * C + ch0, vb);
        Elem<T>::load(xd + (r + BN_RLANES) * C + ch0, db);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            va[k] = fmaxf(va[k] * sc[k] + da[k] * scd[k] + sh[k], 0.f);
            vb[k] = fmaxf(vb[k] * sc[k] + db[k] * scd[k] + sh[k], 0.f);
        }
        Elem<T>::store(y + r * C + ch0, va);
        Elem<T>::store(y + (r + BN_RLANES) * C + ch0, vb);
    }
    for (; r < r1; r += BN_RLANES) {
        float v[V], dv[V];
        Elem<T>::load(x + r * C + ch0, v);
        Elem<T>::load(xd + r * C + ch0, dv);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = fmaxf(v[k] * sc[k] + dv[k] * scd[k] + sh[k], 0.f);
        Elem<T>::store(y + r * C + ch0, v);
    }
}

// The per-channel part of a BatchNorm forward call alone (bn_derive for every channel with the lead's side effects): saved statistics, scale / shift,
// running estimates, the accumulator hand-over -- for a consumer that applies the affine itself (the stem's max-pool, csrc/pool.hip)
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnAffine a, int C) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < C; c += gridDim.x * blockDim.x) {
        float sc, sh;
        bn_derive(a, C, c, true, sc, sh);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.num_batches && a.sums) a.num_batches[0] += 1;
}

template <typename T>
static int bn_act_fwd_impl(const void* x, const void* residual, long long R, int C, const float* gamma, const float* beta,
                           float eps, float momentum, int training, int relu, float* running_mean, float* running_var,
                           long long* num_batches_tracked, float* mean, float* rstd, float* scale_shift, float* sums_ws,
                           float* bwd_sums, void* y, epi_stream_t stream) {
    constexpr int V = Elem<T>::VEC;
    if (!x || !gamma || !beta || !scale_shift || !y) return EPI_ERR_INVALID_ARGUMENT;
    if (training && (!mean || !rstd || !sums_ws)) return EPI_ERR_INVALID_ARGUMENT;
    if (!training && (!running_mean || !running_var)) return EPI_ERR_INVALID_ARGUMENT;
    if ((running_mean == nullptr) != (running_var == nullptr)) return EPI_ERR_INVALID_ARGUMENT;
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (training && training != 2) {             // 2: the producer already accumulated the batch sums into sums_ws
        const int rc = launch_stats<T>((const T*)x, R, C, sums_ws, epi_bn_sum_copies(C), st);
        if (rc != EPI_OK) return rc;
    }
    const long long nvec = R * (C / V);
    const dim3 grid(stream_grid(nvec)), block(BN_THREADS);
    const size_t lds = (size_t)2 * C * sizeof(float);
    if (lds > 65536) return EPI_ERR_UNSUPPORTED;
    BnAffine a;
    a.sums = training ? sums_ws : nullptr; a.ncopies = epi_bn_sum_copies(C); a.R = R; a.inv_r = 1.0 / (double)R; a.gamma = gamma; a.beta = beta; a.eps = eps; a.momentum = momentum;
    a.running_mean = running_mean; a.running_var = running_var; a.num_batches = num_batches_tracked;
    a.mean = mean; a.rstd = rstd; a.scale = scale_shift; a.shift = scale_shift + C; a.bwd_sums = bwd_sums;
    const T* xs = (const T*)x;
    const T* rs = (const T*)residual;
    T* ys = (T*)y;
    if (bn_2d_mode() & 1) {        // 2-D partition: per-channel coefficients in registers (EPI_BN_2D: see bn_2d_mode)
        int rpw = 0;
        dim3 g2;
        apply_blocking(R, C, &rpw, &g2, 8 * V);
        if (relu && residual) hipLaunchKernelGGL((bn_apply2d_kernel<T, true, true>), g2, block, 0, st, xs, rs, R, C, rpw, a, ys);
        else if (relu) hipLaunchKernelGGL((bn_apply2d_kernel<T, true, false>), g2, block, 0, st, xs, rs, R, C, rpw, a, ys);
        else if (residual) hipLaunchKernelGGL((bn_apply2d_kernel<T, false, true>), g2, block, 0, st, xs, rs, R, C, rpw, a, ys);
        else hipLaunchKernelGGL((bn_apply2d_kernel<T, false, false>), g2, block, 0, st, xs, rs, R, C, rpw, a, ys);
        EPI_CHECK_LAUNCH();
        return EPI_OK;
    }
    if (relu && residual) hipLaunchKernelGGL((bn_apply_kernel<T, true, true>), grid, block, lds, st, xs, rs, nvec, C, a, ys);
    else if (relu) hipLaunchKernelGGL((bn_apply_kernel<T, true, false>), grid, block, lds, st, xs, rs, nvec, C, a, ys);
    else if (residual) hipLaunchKernelGGL((bn_apply_kernel<T, false, true>), grid, block, lds, st, xs, rs, nvec, C, a, ys);
    else hipLaunchKernelGGL((bn_apply_kernel<T, false, false>), grid, block, lds, st, xs, rs, nvec, C, a, ys);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_bn_act_fwd(const void* x, const void* residual, long long R, int C, const float* gamma, const float* beta,
                              float eps, float momentum, int training, int relu, float* running_mean, float* running_var,
                              long long* num_batches_tracked, float* mean, float* rstd, float* scale_shift, float* sums_ws,
                              float* bwd_sums, void* y, epi_stream_t stream) {
    return bn_act_fwd_impl<unsigned short>(x, residual, R, C, gamma, beta, eps, momentum, training, relu, running_mean, running_var,
                                           num_batches_tracked, mean, rstd, scale_shift, sums_ws, bwd_sums, y, stream);
}
static void bn_affine_from(const EpiBnLayer& l, long long R, int C, int training, float eps, float momentum, BnAffine* a) {
    a->sums = training ? l.sums_ws : nullptr; a->ncopies = epi_bn_sum_copies(C); a->R = R; a->inv_r = 1.0 / (double)R; a->gamma = l.gamma; a->beta = l.beta;
    a->eps = eps; a->momentum = momentum; a->running_mean = l.running_mean; a->running_var = l.running_var; a->num_batches = l.num_batches_tracked;
    a->mean = l.mean; a->rstd = l.rstd; a->scale = l.scale_shift; a->shift = l.scale_shift + C; a->bwd_sums = training ? l.bwd_sums : nullptr;
}
// training: 2 = the producer delivered the batch sums into layer->sums_ws (the only training mode: there is no tensor here to take them from), 0 = running statistics
extern "C" int epi_bn_finalize(const EpiBnLayer* layer, long long R, int C, float eps, float momentum, int training, epi_stream_t stream) {
    if (!layer || !layer->gamma || !layer->beta || !layer->scale_shift || R <= 0 || C <= 0 || (training != 0 && training != 2)) return EPI_ERR_INVALID_ARGUMENT;
    if (training && (!layer->mean || !layer->rstd || !layer->sums_ws)) return EPI_ERR_INVALID_ARGUMENT;
    if (!training && (!layer->running_mean || !layer->running_var)) return EPI_ERR_INVALID_ARGUMENT;
    BnAffine a;
    bn_affine_from(*layer, R, C, training, eps, momentum, &a);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, C);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
extern "C" int epi_bn_act_fwd_dual(const void* x, const void* x_proj, long long R, int C, const EpiBnLayer* main_bn, const EpiBnLayer* proj_bn,
                                   float eps, float momentum, int training, void* y, epi_stream_t stream) {
    typedef unsigned short T;
    constexpr int V = Elem<T>::VEC;
    if (!x || !x_proj || !main_bn || !proj_bn || !y) return EPI_ERR_INVALID_ARGUMENT;
    for (const EpiBnLayer* l : {main_bn, proj_bn}) {
        if (!l->gamma || !l->beta || !l->scale_shift) return EPI_ERR_INVALID_ARGUMENT;
        if (training && (!l->mean || !l->rstd || !l->sums_ws)) return EPI_ERR_INVALID_ARGUMENT;
        if (!training && (!l->running_mean || !l->running_var)) return EPI_ERR_INVALID_ARGUMENT;
        if ((l->running_mean == nullptr) != (l->running_var == nullptr)) return EPI_ERR_INVALID_ARGUMENT;
    }
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (training == 1)            // (2: both producers already accumulated their batch sums)
        for (int k = 0; k < 2; ++k) {
            const int rc = launch_stats<T>((const T*)(k ? x_proj : x), R, C, (k ? proj_bn : main_bn)->sums_ws, epi_bn_sum_copies(C), st);
            if (rc != EPI_OK) return rc;
        }
    BnAffine a, d;
    bn_affine_from(*main_bn, R, C, training, eps, momentum, &a);
    bn_affine_from(*proj_bn, R, C, training, eps, momentum, &d);
    int rpw2 = 0;
    dim3 g2;
    apply_blocking(R, C, &rpw2, &g2, 8 * V);
    hipLaunchKernelGGL(bn_apply_dual_kernel<T>, g2, dim3(BN_THREADS), 0, st, (const T*)x, (const T*)x_proj, R, C, rpw2, a, d, (T*)y);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
// the same on fp32 activations (x, residual, y are float [R][C]): the fp32-grade verification mode (models/precise.py)
extern "C" int epi_bn_act_fwd_f32(const void* x, const void* residual, long long R, int C, const float* gamma, const float* beta,
                                  float eps, float momentum, int training, int relu, float* running_mean, float* running_var,
                                  long long* num_batches_tracked, float* mean, float* rstd, float* scale_shift, float* sums_ws,
                                  float* bwd_sums, void* y, epi_stream_t stream) {
    return bn_act_fwd_impl<float>(x, residual, R, C, gamma, beta, eps, momentum, training, relu, running_mean, running_var,
                                  num_batches_tracked, mean, rstd, scale_shift, sums_ws, bwd_sums, y, stream);
}

// reduced: dy is already dz and dbeta_dgamma already holds the two sums (the *_bnred backward-data entries of csrc/head_gemm.hip):
// the apply pass alone, without a mask
template <typename T>
static int bn_act_bwd_impl(const void* dy, const void* x, const void* y, long long R, int C, const float* gamma, const float* mean,
                           const float* rstd, const float* scale_shift, int relu, float* dbeta_dgamma, void* dx, void* dres,
                           float* fwd_sums_clear, float* param_grads, epi_stream_t stream, bool reduced = false) {
    constexpr int V = Elem<T>::VEC;
    if (!dy || !x || !gamma || !mean || !rstd || !scale_shift || !dbeta_dgamma || !dx) return EPI_ERR_INVALID_ARGUMENT;
    if (dres && relu && !y) return EPI_ERR_INVALID_ARGUMENT;        // residual + ReLU: the mask comes from the saved output
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rpw = 0;
    dim3 rgrid;
    reduce_blocking(R, C, &rpw, &rgrid, 8 * V);
    const int mask = (!relu || reduced) ? BN_MASK_NONE : (y ? BN_MASK_FROM_Y : BN_MASK_FROM_X);
    const T *dys = (const T*)dy, *xs = (const T*)x, *ys = (const T*)y;
    const float *sc = scale_shift, *sh = scale_shift + C;
    if (!reduced) {
        float* part = nullptr;
        if (deterministic()) {           // per-row-block partial sums, added in index order (csrc/capi.hip)
            part = det_scratch((size_t)rgrid.y * 2 * C, st);
            if (!part) return EPI_ERR_WORKSPACE;
        }
#define EPI_BN_RED(M) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, M>), rgrid, dim3(BN_THREADS), 0, st, dys, xs, ys, R, C, sc, sh, mean, rstd, rpw, dbeta_dgamma, part)
        if (mask == BN_MASK_NONE) EPI_BN_RED(BN_MASK_NONE);
        else if (mask == BN_MASK_FROM_X) EPI_BN_RED(BN_MASK_FROM_X);
        else EPI_BN_RED(BN_MASK_FROM_Y);
#undef EPI_BN_RED
        if (part) hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)((2 * C + 255) / 256)), dim3(256), 0, st, part, (int)rgrid.y, 2 * C, dbeta_dgamma);
        EPI_CHECK_LAUNCH();
    }
    const long long nvec = R * (C / V);
    const dim3 grid(stream_grid(nvec)), block(BN_THREADS);
    T *dxs = (T*)dx, *drs = (T*)dres;
    if (bn_2d_mode() & 2) {
        int rpw2 = 0;
        dim3 g2;
        apply_blocking(R, C, &rpw2, &g2, 8 * V);
#define EPI_BN_APP2(M, D) hipLaunchKernelGGL((bn_bwd_apply2d_kernel<T, M, D>), g2, block, 0, st, dys, xs, ys, R, C, rpw2, gamma, sc, sh, mean, rstd, dbeta_dgamma, dxs, drs, fwd_sums_clear, epi_bn_sum_copies(C), param_grads)
        if (mask == BN_MASK_NONE) { if (dres) EPI_BN_APP2(BN_MASK_NONE, true); else EPI_BN_APP2(BN_MASK_NONE, false); }
        else if (mask == BN_MASK_FROM_X) { if (dres) EPI_BN_APP2(BN_MASK_FROM_X, true); else EPI_BN_APP2(BN_MASK_FROM_X, false); }
        else { if (dres) EPI_BN_APP2(BN_MASK_FROM_Y, true); else EPI_BN_APP2(BN_MASK_FROM_Y, false); }
#undef EPI_BN_APP2
        EPI_CHECK_LAUNCH();
        return EPI_OK;
    }
#define EPI_BN_APP(M, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, M, D>), grid, block, 0, st, dys, xs, ys, nvec, R, C, gamma, sc, sh, mean, rstd, dbeta_dgamma, dxs, drs, fwd_sums_clear, epi_bn_sum_copies(C), param_grads)
    if (mask == BN_MASK_NONE) { if (dres) EPI_BN_APP(BN_MASK_NONE, true); else EPI_BN_APP(BN_MASK_NONE, false); }
    else if (mask == BN_MASK_FROM_X) { if (dres) EPI_BN_APP(BN_MASK_FROM_X, true); else EPI_BN_APP(BN_MASK_FROM_X, false); }
    else { if (dres) EPI_BN_APP(BN_MASK_FROM_Y, true); else EPI_BN_APP(BN_MASK_FROM_Y, false); }
#undef EPI_BN_APP
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_bn_act_bwd(const void* dy, const void* x, const void* y, long long R, int C, const float* gamma, const float* mean,
                              const float* rstd, const float* scale_shift, int relu, float* dbeta_dgamma, void* dx, void* dres,
                              float* fwd_sums_clear, float* param_grads, epi_stream_t stream) {
    return bn_act_bwd_impl<unsigned short>(dy, x, y, R, C, gamma, mean, rstd, scale_shift, relu, dbeta_dgamma, dx, dres, fwd_sums_clear, param_grads, stream);
}
extern "C" int epi_bn_act_bwd_reduced(const void* dz, const void* x, long long R, int C, const float* gamma, const float* mean, const float* rstd,
                                      const float* scale_shift, const float* dbeta_dgamma, void* dx, float* fwd_sums_clear, float* param_grads,
                                      epi_stream_t stream) {
    return bn_act_bwd_impl<unsigned short>(dz, x, nullptr, R, C, gamma, mean, rstd, scale_shift, 0, const_cast<float*>(dbeta_dgamma), dx, nullptr,
                                           fwd_sums_clear, param_grads, stream, true);
}
extern "C" int epi_bn_act_bwd_f32(const void* dy, const void* x, const void* y, long long R, int C, const float* gamma, const float* mean,
                                  const float* rstd, const float* scale_shift, int relu, float* dbeta_dgamma, void* dx, void* dres,
                                  float* fwd_sums_clear, float* param_grads, epi_stream_t stream) {
    return bn_act_bwd_impl<float>(dy, x, y, R, C, gamma, mean, rstd, scale_shift, relu, dbeta_dgamma, dx, dres, fwd_sums_clear, param_grads, stream);
}

// Per-column sum and sum of squares of a matrix x [R][C] (bf16; _f32: float), ACCUMULATED into sums [2C] f32 (caller zeroes it)
// (bias gradient of the final conv: db = sum over rows of dlogits).
template <typename T>
static int column_sums_impl(const void* x, long long R, int C, float* sums, epi_stream_t stream) {
    if (!x || !sums) return EPI_ERR_INVALID_ARGUMENT;
    if (!bn_shape_ok(R, C)) return EPI_ERR_UNSUPPORTED;
    return launch_stats<T>((const T*)x, R, C, sums, 1, (hipStream_t)stream, true);
}
extern "C" int epi_column_sums_bf16(const void* x, long long R, int C, float* sums, epi_stream_t stream) {
    return column_sums_impl<unsigned short>(x, R, C, sums, stream);
}
extern "C" int epi_column_sums_f32(const void* x, long long R, int C, float* sums, epi_stream_t stream) {
    return column_sums_impl<float>(x, R, C, sums, stream);
}
