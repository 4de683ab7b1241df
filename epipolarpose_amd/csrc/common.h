// Shared device helpers for libepipolar_hip (gfx950 only: wave64, 256 CUs, 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "../../include/epipolar_hip.h"

#define EPI_WAVE 64
// float64 per-thread math that the CPU test-suite also compiles for the host (tests/hostcheck): never called on the host by the library
#define EPI_HD __host__ __device__
#define EPI_LOG2E 1.4426950408889634f

#define EPI_CHECK_LAUNCH()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return EPI_ERR_LAUNCH; \
    } while (0)

namespace epi {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

// Deterministic mode (epi_set_deterministic, csrc/capi.hip): every cross-workgroup floating-point sum runs in a FIXED order -- no fp32 atomics.
bool deterministic();
float* det_scratch(size_t floats, hipStream_t stream);      // the stream's device scratch for per-workgroup partial sums (null: not enabled / too small)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float(((unsigned int)v) << 16); }

// round-to-nearest-even float -> bf16 on the gfx950 conversion instruction (v_cvt_pk_bf16_f32: two values per issue;
// NaN stays a quiet NaN).  pack_bf16x2(lo, hi) = hi << 16 | lo.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
    f32x2_t v; v.x = lo; v.y = hi;
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) { return (unsigned short)(pack_bf16x2(f, 0.f) & 0xffffu); }

// Element-type traits for the streaming kernels: VEC elements == 16 bytes per lane per access.
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        float4v t = *reinterpret_cast<const float4v*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        float4v t; t.x = v[0]; t.y = v[1]; t.z = v[2]; t.w = v[3];
        *reinterpret_cast<float4v*>(p) = t;
    }
    __device__ static __forceinline__ float load1(const float* p) { return *p; }
    __device__ static __forceinline__ void store1(float* p, float v) { *p = v; }
};
template <> struct Elem<unsigned short> {   // bf16 storage
    static constexpr int VEC = 8;
    __device__ static __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
        uint4v t = *reinterpret_cast<const uint4v*>(p);
        unsigned int w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ void store(unsigned short* p, const float (&v)[8]) {
        unsigned int w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        uint4v t; t.x = w[0]; t.y = w[1]; t.z = w[2]; t.w = w[3];
        *reinterpret_cast<uint4v*>(p) = t;
    }
    __device__ static __forceinline__ float load1(const unsigned short* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store1(unsigned short* p, float v) { *p = f32_to_bf16(v); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (<= 16 waves); result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float t = (lane < nw) ? smem[lane] : 0.f;
    t = wave_sum(t);
    return t;
}

}  // namespace epi
