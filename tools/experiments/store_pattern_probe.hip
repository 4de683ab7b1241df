// Store-pattern probe (gfx950): how fast can 256 CUs WRITE a [M][N] bf16 matrix when the bytes leave in the pattern of the GEMM epilogues?
//   pattern 0  linear: every workgroup writes 32 KB contiguous, 16 B per lane, 1 KB per wave instruction
//   pattern 1  the 128 x 128 tile of head_gemm_kernel's coalesced epilogue: wave (wm, wn) owns 64 rows x 128 B; one instruction = 8 rows x 128 B (8 lanes x 16 B per row),
//              row pitch N * 2 bytes; tiles in the kernel's order (tile_n fastest, XCD-contiguous ids)
//   pattern 2  the same tile, but one instruction = 4 rows x 256 B (the full tile width: 16 lanes x 16 B per row)
//   pattern 3  a 64 x 256 tile (whole 512-byte rows at N = 256): one instruction = 2 rows x 512 B
// also: the matching LOAD patterns (rd = 1) into registers (sum kept), and load+store (copy A [M][K] -> C [M][N], pattern 1 shapes) to see the mix.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/store_pattern_probe_bin tools/experiments/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int lin, int total) {
    const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int PATTERN, bool RD>
__global__ __launch_bounds__(256) void probe(char* __restrict__ C, int M, int N, unsigned* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long pitch = (long long)N * 2;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    uint4v v; v.x = tid; v.y = lid; v.z = 7; v.w = 9;
    unsigned acc = 0;
    auto touch = [&](char* p) {
        if (RD) { const uint4v t = *reinterpret_cast<const uint4v*>(p); acc += t.x ^ t.y ^ t.z ^ t.w; }
        else *reinterpret_cast<uint4v*>(p) = v;
    };
    if (PATTERN == 0) {
        char* base = C + (long long)lid * 32768;
#pragma unroll
        for (int it = 0; it < 8; ++it) touch(base + (it * 256 + tid) * 16);
    } else if (PATTERN == 1 || PATTERN == 2) {
        const int tiles_n = N / 128, tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
        char* base = C + (long long)tile_m * 128 * pitch + tile_n * 256;
        if (PATTERN == 1) {
            const int wm = wid >> 1, wn = wid & 1, c8 = lane & 7, r8 = lane >> 3;
#pragma unroll
            for (int it = 0; it < 8; ++it) touch(base + (long long)(wm * 64 + it * 8 + r8) * pitch + wn * 128 + c8 * 16);
        } else {
            const int c16 = lane & 15, r4 = lane >> 4;
#pragma unroll
            for (int it = 0; it < 8; ++it) touch(base + (long long)(wid * 32 + it * 4 + r4) * pitch + c16 * 16);
        }
    } else {
        const int tiles_n = N / 256, tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
        char* base = C + (long long)tile_m * 64 * pitch + tile_n * 512;
        const int c32 = lane & 31, r2 = lane >> 5;
#pragma unroll
        for (int it = 0; it < 8; ++it) touch(base + (long long)(wid * 16 + it * 2 + r2) * pitch + c32 * 16);
    }
    if (RD && acc == 0x12345678u) sink[0] = acc;
}

// copy with the GEMM's shapes: read a [128][K] slab of A (K bf16 per row, contiguous rows), write the 128 x 128 tile pattern 1 (tiles_n workgroups read the same slab)
template <int K>
__global__ __launch_bounds__(256) void copy_probe(const char* __restrict__ A, char* __restrict__ C, int M, int N) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const long long pitch = (long long)N * 2;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_n = N / 128, tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    const char* a = A + (long long)tile_m * 128 * K * 2;
    uint4v s; s.x = s.y = s.z = s.w = 0;
    constexpr int VECS = 128 * K * 2 / 16 / 256;            // 16-byte vectors per thread
#pragma unroll
    for (int it = 0; it < VECS; ++it) { const uint4v t = *reinterpret_cast<const uint4v*>(a + (it * 256 + tid) * 16); s.x ^= t.x; s.y ^= t.y; s.z ^= t.z; s.w ^= t.w; }
    char* base = C + (long long)tile_m * 128 * pitch + tile_n * 256;
    const int wm = wid >> 1, wn = wid & 1, c8 = lane & 7, r8 = lane >> 3;
#pragma unroll
    for (int it = 0; it < 8; ++it) *reinterpret_cast<uint4v*>(base + (long long)(wm * 64 + it * 8 + r8) * pitch + wn * 128 + c8 * 16) = s;
}

int main() {
    const int M = 131072;
    const size_t arena = (size_t)3 << 30;
    char* buf; unsigned* sink;
    CK(hipMalloc(&buf, arena)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, arena));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](auto launch, size_t bytes_per, const char* name, double moved) {
        const int nrot = (int)(arena / bytes_per) < 1 ? 1 : (int)(arena / bytes_per), iters = 24;
        for (int i = 0; i < 4; ++i) launch(buf + (size_t)(i % nrot) * bytes_per);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < iters; ++i) launch(buf + (size_t)(i % nrot) * bytes_per);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-64s %7.2f us  %5.2f TB/s\n", name, ms / iters * 1e3, moved / (ms / iters * 1e-3) / 1e12);
    };
    for (int N : {256, 512, 128}) {
        const size_t bytes = (size_t)M * N * 2;
        char name[128];
#define RUN(P, RDV, label) do { snprintf(name, sizeof name, "N=%4d %s %s", N, RDV ? "LOAD " : "STORE", label); \
        timeit([&](char* p) { hipLaunchKernelGGL((probe<P, RDV>), dim3((unsigned)(bytes / 32768)), dim3(256), 0, 0, p, M, N, sink); }, bytes, name, (double)bytes); } while (0)
        RUN(0, false, "linear 32 KB per workgroup");
        RUN(1, false, "128x128 tile, 8 rows x 128 B per instruction (the epilogue)");
        RUN(2, false, "128x128 tile, 4 rows x 256 B per instruction");
        if (N >= 256) RUN(3, false, "64x256 tile, 2 rows x 512 B per instruction");
        RUN(0, true, "linear 32 KB per workgroup");
        RUN(1, true, "128x128 tile, 8 rows x 128 B per instruction");
    }
    {   // copy shapes: 64 -> 256 (K = 64, N = 256) and 256 -> 64 is linear, 128 -> 512
        const size_t abytes = (size_t)M * 64 * 2, cbytes = (size_t)M * 256 * 2;
        timeit([&](char* p) { hipLaunchKernelGGL((copy_probe<64>), dim3((unsigned)(cbytes / 32768)), dim3(256), 0, 0, p, p + abytes, M, 256); }, abytes + cbytes,
               "COPY  A [M][64] -> C [M][256], 128x128 tiles (64 -> 256 conv traffic)", (double)(abytes + cbytes));
        const int M2 = 32768;
        const size_t a2 = (size_t)M2 * 128 * 2, c2 = (size_t)M2 * 512 * 2;
        timeit([&](char* p) { hipLaunchKernelGGL((copy_probe<128>), dim3((unsigned)(c2 / 32768)), dim3(256), 0, 0, p, p + a2, M2, 512); }, a2 + c2,
               "COPY  A [32768][128] -> C [32768][512], 128x128 tiles (128 -> 512)", (double)(a2 + c2));
    }
    return 0;
}
