// Inverted dropout for the refiner MLP (reference refiner/model.py:26,47-67: nn.Dropout(p) after every Linear -> BatchNorm -> ReLU).
// Counter-based generator: the keep decision of element i is a hash of (seed, i) -- no state, reproducible, and the backward pass
// regenerates the same mask instead of storing it.  y = keep ? x / (1 - p) : 0.
#include "common.h"

namespace epi {

__device__ __forceinline__ unsigned int mix32(unsigned long long seed, unsigned long long i) {
    unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;          // splitmix64
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (unsigned int)(z >> 32);
}

// forward: y = mask * x * scale;  backward (same kernel): dx = mask * dy * scale
__global__ __launch_bounds__(256) void dropout_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, long long n,
                                                      unsigned long long seed, unsigned int threshold, float scale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool keep = mix32(seed, (unsigned long long)i) >= threshold;
    y[i] = keep ? f32_to_bf16(bf16_to_f32(x[i]) * scale) : (unsigned short)0;
}

}  // namespace epi

extern "C" int epi_dropout_bf16(const void* x, void* y, long long n, float p, unsigned long long seed, epi_stream_t stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f) || !(p < 1.f)) return EPI_ERR_INVALID_ARGUMENT;
    const unsigned int threshold = (unsigned int)((double)p * 4294967296.0);
    hipLaunchKernelGGL(epi::dropout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)x,
                       (unsigned short*)y, n, seed, threshold, 1.f / (1.f - p));
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
