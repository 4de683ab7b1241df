// LDS read throughput per CU: ds_read_b64_tr_b16 (the transposing read both operands of a weight-gradient MFMA need) against plain ds_read_b64 / ds_read_b128,
// 4 waves per workgroup, one workgroup per CU, conflict-free addresses.  hipcc --offload-arch=gfx950 -O3 -o tools/lds_tr_probe_bin tools/lds_tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(int iters, unsigned int* sink, int shift, int swizzle) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned int*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned acc = 0;
    // 8 independent reads per iteration, addresses walk through 64 KB
    unsigned a = base + wid * 16384 + (KIND == 2 ? lane * 16 : lane * 8);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned addr = a + ((u * (KIND == 2 ? 1024 : 512)) & 8191);
            if (KIND == 0) { s16x4 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("" :: "v"(v)); acc += (unsigned)v.x; }
            if (KIND == 1) { s16x4 v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("" :: "v"(v)); acc += (unsigned)v.x; }
            if (KIND == 2) { u32x4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr)); asm volatile("" :: "v"(v)); acc += v.x; }
            if (KIND >= 3) {
                // csrc/head_gemm.hip tr_frag: lane_c = lane & 15, group = lane >> 4: rows row0 + (lane_c >> 2) (+ 8 for groups 2, 3), 16-byte chunk (col >> 3) + ((lane_c & 3) >> 1)
                // XOR-ed with the row's swizzle, + 8 bytes for odd lane_c; columns + 16 elements for odd groups.  row0 = 16 * (u & 3) + shift, col = 32 * (u >> 2)
                constexpr int ROWB = KIND == 3 ? 256 : 128;
                const int lane_c = lane & 15, grp = lane >> 4;
                const int row = 16 * (u & 3) + 8 * (grp >> 1) + (lane_c >> 2) + shift;
                const int col = 32 * (u >> 2) + 16 * (grp & 1);
                const int chunk = (col >> 3) + ((lane_c & 3) >> 1);
                const int swz = swizzle ? (ROWB >= 256 ? 4 * (row & 3) : 4 * ((row >> 1) & 1)) : 0;
                const unsigned ad = base + wid * 16384 + row * ROWB + ((chunk ^ swz) << 4) + 8 * (lane_c & 1);
                s16x4 v; asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(ad)); asm volatile("" :: "v"(v)); acc += (unsigned)v.x;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int KIND> static double run(const char* name, int bytes_per_lane, int shift = 0, int swizzle = 1) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    unsigned int* sink; CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int iters = 20000, grid = 256;
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 65536, 0, 100, sink, shift, swizzle);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(256), 65536, 0, iters, sink, shift, swizzle);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double reads_per_cu = (double)iters * 8 * 4;                 // wave-level read instructions per CU
    const double ns_per_read = ms * 1e6 / reads_per_cu;
    printf("%-22s %6.2f ns per wave-level read per CU  = %6.1f bytes/ns/CU (%d B per lane)\n", name, ns_per_read, 64.0 * bytes_per_lane / ns_per_read, bytes_per_lane);
    return ns_per_read;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    run<0>("ds_read_b64_tr_b16", 8);
    run<1>("ds_read_b64", 8);
    run<2>("ds_read_b128", 16);
    printf("# the weight-gradient kernels' operand pattern (tr_frag): 8 rows x 64 bytes per wave-level read\n");
    run<3>("tr, 256-B rows, swizzled", 8, 0, 1);
    run<3>("tr, 256-B rows, plain", 8, 0, 0);
    run<4>("tr, 128-B rows, swizzled", 8, 0, 1);
    run<4>("tr, 128-B rows, plain", 8, 0, 0);
    run<4>("tr, 128-B rows, shift 1", 8, 1, 1);
    run<4>("tr, 128-B rows, shift 3", 8, 3, 1);
    return 0;
}
