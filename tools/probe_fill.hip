// Fill-rate probe (gfx950): how fast can ONE compute unit move a GEMM-style operand tile from global memory into LDS?
// Answers the question behind DESIGN.md section 4a ("~23 GB/s per CU of global->LDS fill"): is that the LDS-DMA path
// (global_load_lds_dwordx4), the L2 / MALL / HBM level the bytes come from, or the number of bytes in flight?
//
//   probe_fill                      prints a table: mode x source span x workgroups per CU x stages in flight
//
// Every workgroup (256 threads) loops `iters` times over "stage one 32 KiB tile": 8 pieces of 1 KiB per wave, rows of 128
// bytes picked from a span of `span` bytes (shared by all workgroups when the span is small, so that it lives in every XCD's L2).
//   mode 0  global_load_lds_dwordx4 (DMA, no VGPRs), a ring of NS LDS stages, counted vmcnt: NS-1 tiles in flight
//   mode 1  global_load_dwordx4 -> VGPR -> ds_write_b128, NS-1 tiles in flight in registers
// No MFMA, no fragment reads: the result is the ceiling of the fill path alone.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_fill_bin tools/probe_fill.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// tile t of workgroup w: 256 rows of 128 bytes; row r of the tile lives at ((w * 977 + t * 131 + r * 17) * 128) mod span
// (scattered 128-byte rows, like the rows of an activation matrix with a long row pitch)
template <int MODE, int NS>
__global__ void __launch_bounds__(256) fill_kernel(const char* __restrict__ src, size_t span_rows, int iters, unsigned int* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t w = blockIdx.x;
    unsigned int acc = 0;
    // piece p (0..7) of this wave covers tile rows (wave*8 + p)*8 .. +7, 8 lanes (16 B each) per row
    auto src_of = [&](int t, int p) -> const char* {
        const unsigned int r = (unsigned int)(wave * 8 + p) * 8 + (lane >> 3);
        const unsigned int row = ((unsigned int)w * 977u + (unsigned int)t * 131u + r * 17u) & (unsigned int)(span_rows - 1);      // span: a power of two
        return src + (size_t)row * 128 + (lane & 7) * 16;
    };
    if (MODE == 0) {
        // prologue: NS-1 tiles in flight
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
#pragma unroll
            for (int p = 0; p < 8; ++p) dma16(src_of(s, p), lds + s * 32768 + (wave * 8 + p) * 1024);
        for (int t = 0; t < iters; ++t) {
            const int nxt = t + NS - 1;
            {
                char* dst = lds + (nxt % NS) * 32768;
#pragma unroll
                for (int p = 0; p < 8; ++p) dma16(src_of(nxt, p), dst + (wave * 8 + p) * 1024);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * 8) : "memory");      // tile t has landed (loads return in order)
            __syncthreads();
            acc += *reinterpret_cast<const unsigned int*>(lds + (t % NS) * 32768 + threadIdx.x * 4);    // touch the tile
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        uint4v reg[NS - 1][8];
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
#pragma unroll
            for (int p = 0; p < 8; ++p) reg[s][p] = *reinterpret_cast<const uint4v*>(src_of(s, p));
        for (int t0 = 0; t0 < iters; t0 += NS - 1) {
#pragma unroll
            for (int s = 0; s < NS - 1; ++s) {
                const int t = t0 + s;
                char* dst = lds + (t & 1) * 32768;
#pragma unroll
                for (int p = 0; p < 8; ++p) *reinterpret_cast<uint4v*>(dst + (wave * 8 + p) * 1024 + lane * 16) = reg[s][p];
#pragma unroll
                for (int p = 0; p < 8; ++p) reg[s][p] = *reinterpret_cast<const uint4v*>(src_of(t + NS - 1, p));
                __syncthreads();
                acc += *reinterpret_cast<const unsigned int*>(lds + (t & 1) * 32768 + threadIdx.x * 4);
                __syncthreads();
            }
        }
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc += reg[s][p].x;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int NS>
static double run(const char* src, size_t span, int wgs, int iters, unsigned int* sink, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<MODE, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((fill_kernel<MODE, NS>), dim3(wgs), dim3(256), lds, 0, src, span / 128, 50, sink);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((fill_kernel<MODE, NS>), dim3(wgs), dim3(256), lds, 0, src, span / 128, iters, sink);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (hipGetLastError() != hipSuccess) return -1;
    return (double)wgs * iters * 32768.0 / (ms * 1e-3);      // bytes per second, whole chip
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("# %s, %d CUs, %.2f GHz; tile = 32 KiB per workgroup and step; GB/s per CU (B/clk per CU) | TB/s chip\n", prop.name, cus, clk * 1e-9);
    const size_t max_span = (size_t)2 << 30;
    char* src = nullptr;
    unsigned int* sink = nullptr;
    hipMalloc(&src, max_span + 4096);
    hipMalloc(&sink, 64);
    hipMemset(src, 1, max_span);
    const size_t spans[] = {(size_t)1 << 20, (size_t)32 << 20, (size_t)2 << 30};          // L2 of every XCD / MALL / HBM
    const char* span_name[] = {"1 MiB (L2)", "32 MiB (MALL)", "2 GiB (HBM)"};
    for (int si = 0; si < 3; ++si) {
        for (int wpc = 1; wpc <= 2; ++wpc) {
            const int wgs = cus * wpc;
            // the LDS allocation sets the residency: > 80 KiB = one workgroup per CU, <= 80 KiB = two
            const size_t one = 96 << 10, two = 64 << 10;
            struct R { const char* name; double v; } rows[] = {
                {"dma  2 stages (1 in flight)", run<0, 2>(src, spans[si], wgs, 2000, sink, wpc == 1 ? one : two)},
                {"dma  3 stages (2 in flight)", wpc == 1 ? run<0, 3>(src, spans[si], wgs, 2000, sink, one) : -1.0},
                {"dma  4 stages (3 in flight)", wpc == 1 ? run<0, 4>(src, spans[si], wgs, 2000, sink, 128 << 10) : -1.0},
                {"vgpr 1 tile in flight      ", run<1, 2>(src, spans[si], wgs, 2000, sink, wpc == 1 ? one : two)},
                {"vgpr 2 tiles in flight     ", run<1, 3>(src, spans[si], wgs, 2000, sink, wpc == 1 ? one : two)},
            };
            for (auto& r : rows)
                if (r.v > 0)
                    printf("%-14s %d wg/CU  %s  %7.1f GB/s/CU (%5.1f B/clk)  %6.2f TB/s\n", span_name[si], wpc, r.name, r.v / cus * 1e-9, r.v / cus / clk,
                           r.v * 1e-12);
        }
    }
    return 0;
}
