// Fused multi-tensor Adam for gfx950: ONE launch updates every parameter of the network.
//
// Replaces torch.optim.Adam's ~25 multi_tensor_apply launches per step (lib/utils/utils.py:55-59 of the reference
// builds `optim.Adam(model.parameters(), lr=cfg.TRAIN.LR)`), and -- for convolution weights that train through a
// bf16 copy -- also the per-step fp32->bf16 weight casts and bf16->fp32 gradient casts of autocast: the kernel reads
// the bf16 gradient directly and writes the fp32 master weight AND its bf16 shadow.
// Arithmetic follows torch.optim.Adam (no amsgrad, no weight decay):
//   m = m + (g - m)(1 - b1);  v = v*b2 + g*g*(1 - b2);  p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// HBM-bound: 16 B read (+4/2 B grad) and 12 B (+2 B shadow) written per element.
#include "common.h"

namespace epi {

struct AdamTensor {            // one row of the device-resident table (48 bytes); `g` is refreshed every step
    float* p; const void* g; float* m; float* v; unsigned short* shadow; long long n; int g_bf16; int pad;
};
constexpr int ADAM_THREADS = 256;
constexpr int ADAM_CHUNK = 16384;       // elements per workgroup

// sum of squares of every gradient (the total norm torch.nn.utils.clip_grad_norm_ computes; refiner/main.py:53): one atomic per chunk
__global__ __launch_bounds__(ADAM_THREADS) void grad_sumsq_kernel(const AdamTensor* __restrict__ table, const int2* __restrict__ chunks,
                                                                  float* __restrict__ sumsq, float* __restrict__ part) {
    __shared__ float red[17];
    const int2 ck = chunks[blockIdx.x];
    const AdamTensor t = table[ck.x];
    const long long base = (long long)ck.y * ADAM_CHUNK;
    const long long end = (base + ADAM_CHUNK < t.n) ? base + ADAM_CHUNK : t.n;
    float s = 0.f;
    for (long long j = base + threadIdx.x; j < end; j += ADAM_THREADS) {
        const float g = t.g_bf16 ? bf16_to_f32(reinterpret_cast<const unsigned short*>(t.g)[j]) : reinterpret_cast<const float*>(t.g)[j];
        s = fmaf(g, g, s);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        if (part) part[blockIdx.x] = s;         // deterministic mode: one partial per chunk, added in index order by sumsq_total_kernel
        else atomicAdd(sumsq, s);
    }
}

// deterministic mode (csrc/capi.hip): *sumsq += part[0] + part[1] + ... in a FIXED order (thread t adds part[t], part[t + 256], ...; then the fixed tree of block_sum)
__global__ __launch_bounds__(ADAM_THREADS) void sumsq_total_kernel(const float* __restrict__ part, int n, float* __restrict__ sumsq) {
    __shared__ float red[17];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += ADAM_THREADS) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) sumsq[0] += s;
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_multi_kernel(const AdamTensor* __restrict__ table, const int2* __restrict__ chunks,
                                                                  float b1, float b2, float eps, float step_size, float inv_sqrt_bc2,
                                                                  const float* __restrict__ clip_sumsq, float max_norm) {
    const int2 ck = chunks[blockIdx.x];
    const AdamTensor t = table[ck.x];
    // gradient clipping by total norm (clip_grad_norm_): every gradient is scaled by min(1, max_norm / (norm + 1e-6))
    const float gscale = clip_sumsq ? fminf(1.f, max_norm / (sqrtf(*clip_sumsq) + 1e-6f)) : 1.f;
    const long long base = (long long)ck.y * ADAM_CHUNK;
    const long long end = (base + ADAM_CHUNK < t.n) ? base + ADAM_CHUNK : t.n;
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15u) == 0 &&
                        (reinterpret_cast<uintptr_t>(t.g) & (t.g_bf16 ? 7u : 15u)) == 0 &&
                        (t.shadow == nullptr || (reinterpret_cast<uintptr_t>(t.shadow) & 7u) == 0);
    long long i = base + (long long)threadIdx.x * 4;
    if (vec_ok) {
        for (; i + 3 < end; i += ADAM_THREADS * 4) {
            float4v p = *reinterpret_cast<const float4v*>(t.p + i);
            float4v m = *reinterpret_cast<const float4v*>(t.m + i);
            float4v v = *reinterpret_cast<const float4v*>(t.v + i);
            float g[4];
            if (t.g_bf16) {
                const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(t.g) + i);
                g[0] = __uint_as_float(r.x << 16); g[1] = __uint_as_float(r.x & 0xffff0000u);
                g[2] = __uint_as_float(r.y << 16); g[3] = __uint_as_float(r.y & 0xffff0000u);
            } else {
                const float4v r = *reinterpret_cast<const float4v*>(reinterpret_cast<const float*>(t.g) + i);
                g[0] = r.x; g[1] = r.y; g[2] = r.z; g[3] = r.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] *= gscale;
            float pp[4] = {p.x, p.y, p.z, p.w}, mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mm[k] = mm[k] + (g[k] - mm[k]) * (1.f - b1);
                vv[k] = vv[k] * b2 + g[k] * g[k] * (1.f - b2);
                pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) * inv_sqrt_bc2 + eps);
            }
            p.x = pp[0]; p.y = pp[1]; p.z = pp[2]; p.w = pp[3];
            m.x = mm[0]; m.y = mm[1]; m.z = mm[2]; m.w = mm[3];
            v.x = vv[0]; v.y = vv[1]; v.z = vv[2]; v.w = vv[3];
            *reinterpret_cast<float4v*>(t.p + i) = p;
            *reinterpret_cast<float4v*>(t.m + i) = m;
            *reinterpret_cast<float4v*>(t.v + i) = v;
            if (t.shadow) {
                uint2 o;
                o.x = pack_bf16x2(pp[0], pp[1]);
                o.y = pack_bf16x2(pp[2], pp[3]);
                *reinterpret_cast<uint2*>(t.shadow + i) = o;
            }
        }
    }
    // scalar path: the <= 3 leftover elements of a vectorised chunk, or every element of an unaligned tensor
    const long long sbegin = vec_ok ? end - ((end - base) & 3) : base;
    for (long long j = sbegin + threadIdx.x; j < end; j += ADAM_THREADS) {
        const float g = gscale * (t.g_bf16 ? bf16_to_f32(reinterpret_cast<const unsigned short*>(t.g)[j]) : reinterpret_cast<const float*>(t.g)[j]);
        float m = t.m[j], v = t.v[j], p = t.p[j];
        m = m + (g - m) * (1.f - b1);
        v = v * b2 + g * g * (1.f - b2);
        p -= step_size * m / (sqrtf(v) * inv_sqrt_bc2 + eps);
        t.m[j] = m; t.v[j] = v; t.p[j] = p;
        if (t.shadow) t.shadow[j] = f32_to_bf16(p);
    }
}

}  // namespace epi

extern "C" size_t epi_adam_tensor_bytes(void) { return sizeof(epi::AdamTensor); }
extern "C" int epi_adam_chunk_elems(void) { return epi::ADAM_CHUNK; }

// table: device array of AdamTensor rows; chunks: device array of (tensor index, chunk index) pairs, nchunks entries.
extern "C" int epi_adam_step(const void* table, const void* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                             long long step, epi_stream_t stream) {
    if (!table || !chunks || nchunks <= 0 || step < 1) return EPI_ERR_INVALID_ARGUMENT;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(epi::adam_multi_kernel, dim3(nchunks), dim3(epi::ADAM_THREADS), 0, (hipStream_t)stream,
                       (const epi::AdamTensor*)table, (const int2*)chunks, beta1, beta2, eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
                       (const float*)nullptr, 0.f);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// The same with torch.nn.utils.clip_grad_norm_(parameters, max_norm) folded in (reference refiner/main.py:53-54): one launch
// reduces the squared total gradient norm into `norm_sq` (device scalar, ZERO on entry; it holds the squared norm afterwards),
// the Adam launch scales every gradient by min(1, max_norm / (norm + 1e-6)) as it reads it.  The gradients are not modified.
extern "C" int epi_adam_step_clipped(const void* table, const void* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                                     long long step, float max_norm, float* norm_sq, epi_stream_t stream) {
    if (!table || !chunks || !norm_sq || nchunks <= 0 || step < 1 || !(max_norm > 0.f)) return EPI_ERR_INVALID_ARGUMENT;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    float* part = epi::deterministic() ? epi::det_scratch((size_t)nchunks, (hipStream_t)stream) : nullptr;
    hipLaunchKernelGGL(epi::grad_sumsq_kernel, dim3(nchunks), dim3(epi::ADAM_THREADS), 0, (hipStream_t)stream,
                       (const epi::AdamTensor*)table, (const int2*)chunks, norm_sq, part);
    EPI_CHECK_LAUNCH();
    if (part) {
        hipLaunchKernelGGL(epi::sumsq_total_kernel, dim3(1), dim3(epi::ADAM_THREADS), 0, (hipStream_t)stream, (const float*)part, nchunks, norm_sq);
        EPI_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(epi::adam_multi_kernel, dim3(nchunks), dim3(epi::ADAM_THREADS), 0, (hipStream_t)stream,
                       (const epi::AdamTensor*)table, (const int2*)chunks, beta1, beta2, eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
                       (const float*)norm_sq, max_norm);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
