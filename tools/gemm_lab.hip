// GEMM laboratory (gfx950): answers, on the GPU box, WHERE the time of the small implicit-GEMM launches goes.
//
//   gemm_lab probe            synthetic fill probe with the GEMM's REAL addressing (rows of a power-of-two pitch, every workgroup walking the
//                             same K offsets in lock step) against padded pitches and against a per-workgroup K rotation
//   gemm_lab gemm             the real kernels through the C ABI (libepipolar_hip.so): ResNet-50 layer shapes at batch 32, padded pitches,
//                             tile / pipeline overrides, operands rotating through > 600 MB of buffers (nothing L2- or MALL-resident) or warm
//   gemm_lab conv             3x3 layers through epi_conv2d_fwd: patch kernel vs generic gather kernel
//   gemm_lab layers [batch]   the per-layer table of all 22 distinct ResNet-50 convolutions: forward / backward-data / backward-weight, us per launch
//
// Every figure is the mean over back-to-back launches on one stream between two HIP events (no host work between launches).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_lab_bin tools/gemm_lab.hip -Lepipolarpose_amd/_lib -lepipolar_hip -Wl,-rpath,'$ORIGIN/../epipolarpose_amd/_lib'
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <functional>
#include <algorithm>
#include "../include/epipolar_hip.h"

extern "C" int epi_gemm_tune(int tile, int pipe);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((int)(h & 0xffff) - 32768) * (1.0f / 65536.0f);      // [-0.5, 0.5)
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// The NT kernel's staging pattern without the MFMAs: workgroup w = (tile_m, tile_n) with tile_n fastest over `tiles_n`; per K step it stages
// rows [tile_m*128, +128) of A (pitch bytes apart) and rows [tile_n*128, +128) of B, 128 bytes of each row, at byte offset kt*128 of the row;
// 2 LDS stages, vmcnt(0) + barrier per step (the CfgSmall loop).  rot: M tile t starts at K tile (t * rot) % nk.
__global__ void __launch_bounds__(256) nt_fill_kernel(const char* __restrict__ A, const char* __restrict__ B, size_t pitch, int nk, int tiles_n, int rot,
                                                      unsigned int* sink) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order as in the real kernel: logical id contiguous per XCD
    const int total = gridDim.x, lin = blockIdx.x;
    const int q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    const char* a_row[4];
    const char* b_row[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (wave * 4 + p) * 8 + (lane >> 3);
        a_row[p] = A + ((size_t)tile_m * 128 + row) * pitch + (lane & 7) * 16;
        b_row[p] = B + ((size_t)tile_n * 128 + row) * pitch + (lane & 7) * 16;
    }
    int kt = rot ? (int)(((long long)tile_m * rot) % nk) : 0;
    unsigned int acc = 0;
    auto issue = [&](int buf) {
        char* dst = lds + buf * 32768 + wave * 4096;
#pragma unroll
        for (int p = 0; p < 4; ++p) dma16(a_row[p] + (size_t)kt * 128, dst + p * 1024);
#pragma unroll
        for (int p = 0; p < 4; ++p) dma16(b_row[p] + (size_t)kt * 128, dst + 16384 + p * 1024);
        kt = kt + 1 == nk ? 0 : kt + 1;
    };
    issue(0);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) issue((t + 1) & 1);
        acc += *reinterpret_cast<const unsigned int*>(lds + (t & 1) * 32768 + threadIdx.x * 4);
        __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

static double time_launches(int reps, const std::function<void(int)>& launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch(i + 3);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms * 1e3 / reps;      // us per launch
}

static char* g_arena = nullptr;
static size_t g_arena_bytes = 0;
static unsigned int* g_sink = nullptr;

static void probe() {
    printf("# probe: 128x128-tile staging pattern of the NT kernel without MFMAs, 2 stages, 32 KiB per workgroup and K step, nk = 16 K steps\n");
    printf("# pitch = row pitch in bytes of both operands (A: 8192+ rows, B: 256 rows); wgs = workgroups (tiles_n = 2); rot = K rotation multiplier\n");
    printf("# fresh = operands rotate through %zu MiB (no L2 / MALL reuse between launches); us per launch | GB/s per busy CU | TB/s\n", g_arena_bytes >> 20);
    const size_t pitches[] = {512, 1024, 2048, 4096, 2048 + 128, 4096 + 128, 2048 + 256};
    const int wgss[] = {128, 256, 512};
    for (size_t pitch : pitches) {
        for (int wgs : wgss) {
            const int nk = (int)std::min<size_t>(16, pitch / 128);
            const size_t a_bytes = (size_t)(wgs / 2) * 128 * pitch, b_bytes = 256 * pitch;
            const size_t set = (a_bytes + b_bytes + 4095) / 4096 * 4096;
            const int nset_fresh = (int)std::min<size_t>(64, g_arena_bytes / set);
            for (int fresh = 0; fresh < 2; ++fresh)
                for (int rot = 0; rot < 2; ++rot) {
                    const int nset = fresh ? nset_fresh : 1;
                    const double us = time_launches(40, [&](int i) {
                        const char* base = g_arena + (size_t)(i % nset) * set;
                        hipLaunchKernelGGL(nt_fill_kernel, dim3(wgs), dim3(256), 65536, 0, base, base + a_bytes, pitch, nk, 2, rot, g_sink);
                    });
                    const double bytes = (double)wgs * nk * 32768.0;
                    const int busy = std::min(wgs, 256);
                    printf("pitch %5zu  wgs %4d  %s  rot %d   %7.2f us   %6.1f GB/s/CU   %5.2f TB/s\n", pitch, wgs, fresh ? "fresh" : "warm ", rot, us,
                           bytes / us * 1e-3 / busy, bytes / us * 1e-6);
                }
        }
    }
}

struct Shape { const char* name; int M, N, K; };

static void gemm_battery(bool quick) {
    const Shape shapes[] = {
        {"l1.c1  256->64  ", 131072, 64, 256},   {"l1.c3  64->256  ", 131072, 256, 64},   {"l2.c1  512->128 ", 32768, 128, 512},
        {"l2.c3  128->512 ", 32768, 512, 128},   {"l3.0c1 512->256 ", 32768, 256, 512},   {"l3.c1  1024->256", 8192, 256, 1024},
        {"l3.c3  256->1024", 8192, 1024, 256},   {"l4.0c1 1024->512", 8192, 512, 1024},   {"l4.c1  2048->512", 2048, 512, 2048},
        {"l4.c3  512->2048", 2048, 2048, 512},   {"fin.dX 1088->256", 131072, 256, 1088},
    };
    printf("# gemm: C[M][N] = A[M][K] * Bt[N][K]^T bf16 through epi_gemm_bf16; us per launch, fresh operands (rotating sets) | warm (one set)\n");
    printf("# cfg: tile 0 = by shape, 3 half 64x128, 4 quarter 64x64; pipe 2 = 4-stage ring; pad = lda/ldb + 64 elements\n");
    for (const Shape& s : shapes) {
        struct Var { const char* name; int tile, pipe, pad; };
        std::vector<Var> vars = {{"default        ", 0, 0, 0}, {"pad64          ", 0, 0, 64}, {"pipe2          ", 0, 2, 0}};
        if (!quick) {
            vars.push_back({"half           ", 3, 0, 0});
            vars.push_back({"quarter        ", 4, 0, 0});
        }
        for (const Var& v : vars) {
            const int lda = s.K + v.pad, ldb = s.K + v.pad, ldc = s.N;
            const size_t a_bytes = (size_t)s.M * lda * 2, b_bytes = (size_t)s.N * ldb * 2, c_bytes = (size_t)s.M * ldc * 2;
            const size_t ws_bytes = epi_gemm_workspace_bytes(s.M, s.N, s.K, 1);
            const size_t set = ((a_bytes + b_bytes + c_bytes + 4095) / 4096) * 4096;
            if (ws_bytes + set > g_arena_bytes) { printf("%s %s  (arena too small)\n", s.name, v.name); continue; }
            char* ws = g_arena;
            char* sets = g_arena + (ws_bytes + 4095) / 4096 * 4096;
            const int nfresh = (int)std::max<size_t>(1, std::min<size_t>(48, (g_arena_bytes - (sets - g_arena)) / set));
            hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (unsigned short*)g_arena, g_arena_bytes / 2, 12345u);     // (results of earlier variants are operands now)
            epi_gemm_tune(v.tile, v.pipe);
            double us[2];
            int rc = 0;
            for (int fresh = 1; fresh >= 0; --fresh) {
                const int nset = fresh ? nfresh : 1;
                us[fresh] = time_launches(fresh ? 40 : 40, [&](int i) {
                    char* base = sets + (size_t)(i % nset) * set;
                    rc |= epi_gemm_bf16(base, lda, base + a_bytes, ldb, base + a_bytes + b_bytes, ldc, EPI_BF16, s.M, s.N, s.K, nullptr, ws, ws_bytes, 0);
                });
            }
            const double flop = 2.0 * s.M * s.N * s.K, bytes = (double)a_bytes + b_bytes + c_bytes;
            printf("%s %s  fresh %7.2f us (%6.1f TF, %5.2f TB/s)   warm %7.2f us (%6.1f TF)   rc %d  sets %d\n", s.name, v.name, us[1], flop / us[1] * 1e-6,
                   bytes / us[1] * 1e-6, us[0], flop / us[0] * 1e-6, rc, nfresh);
        }
        epi_gemm_tune(0, -1);
    }
}

static void conv_battery() {
    struct CS { const char* name; int B, H, W, Cin, Cout, k, stride; };
    const CS shapes[] = {{"l1.c2 64 3x3 ", 32, 64, 64, 64, 64, 3, 1},   {"l2.c2 128 3x3", 32, 32, 32, 128, 128, 3, 1}, {"l3.c2 256 3x3", 32, 16, 16, 256, 256, 3, 1},
                         {"l4.c2 512 3x3", 32, 8, 8, 512, 512, 3, 1},  {"l2.0c2 s2    ", 32, 64, 64, 128, 128, 3, 2}, {"l3.0c2 s2    ", 32, 32, 32, 256, 256, 3, 2},
                         {"l4.0c2 s2    ", 32, 16, 16, 512, 512, 3, 2}, {"l3.ds 1x1 s2 ", 32, 32, 32, 512, 1024, 1, 2}, {"l4.ds 1x1 s2 ", 32, 16, 16, 1024, 2048, 1, 2}};
    printf("# conv: epi_conv2d_fwd (no BatchNorm sums), fresh operands; patch = conv_patch_kernel (mode 2) / generic gather kernel (mode 0)\n");
    for (const CS& s : shapes) {
        const int pad = s.k / 2, Ho = (s.H + 2 * pad - s.k) / s.stride + 1, Wo = (s.W + 2 * pad - s.k) / s.stride + 1;
        const size_t x_bytes = (size_t)s.B * s.H * s.W * s.Cin * 2, w_bytes = (size_t)s.Cout * s.k * s.k * s.Cin * 2, y_bytes = (size_t)s.B * Ho * Wo * s.Cout * 2;
        const size_t ws_bytes = epi_conv2d_workspace_bytes(s.B, s.H, s.W, s.Cin, s.Cout, s.k, s.k, s.stride, pad);
        const size_t set = ((x_bytes + w_bytes + y_bytes + 4095) / 4096) * 4096;
        char* ws = g_arena;
        char* sets = g_arena + (ws_bytes + 4095) / 4096 * 4096;
        const int nfresh = (int)std::max<size_t>(1, std::min<size_t>(48, (g_arena_bytes - (sets - g_arena)) / set));
        for (int patch = 2; patch >= 0; patch -= 2) {
            {
                if (patch == 2 && (s.stride != 1 || s.k != 3)) continue;
                hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (unsigned short*)g_arena, g_arena_bytes / 2, 12345u);
                epi_conv3x3_patch_mode(patch);
                int rc = 0;
                const double us = time_launches(40, [&](int i) {
                    char* base = sets + (size_t)(i % nfresh) * set;
                    rc |= epi_conv2d_fwd(base, base + x_bytes, base + x_bytes + w_bytes, s.B, s.H, s.W, s.Cin, s.Cout, s.k, s.k, s.stride, pad, nullptr, nullptr, ws, ws_bytes, 0);
                });
                const double flop = 2.0 * s.B * Ho * Wo * s.Cout * s.k * s.k * s.Cin;
                printf("%s  %s   %7.2f us (%6.1f TF)  rc %d\n", s.name, patch == 2 ? "patch  " : "generic", us, flop / us * 1e-6, rc);
            }
        }
    }
    epi_conv3x3_patch_mode(2);
}

// Every distinct ResNet-50 convolution behind the stem at batch 32: forward (with the BatchNorm sums from its epilogue, as the step runs it), backward-data
// and backward-weight (one launch per layer; the step groups these per ResNet stage), back-to-back launches on operands that rotate through the arena.
// The C++ successor of tools/bench_conv.py, whose Python launch path put a ~12 us floor under every entry (profiles/r03_conv_layers_final.txt).
static void layers_table(int batch) {
    struct L { const char* name; int n, Cin, Cout, k, s, H; };
    const L layers[] = {{"l1.c1", 1, 64, 64, 1, 1, 64},    {"l1.c2", 3, 64, 64, 3, 1, 64},     {"l1.c3", 4, 64, 256, 1, 1, 64},   {"l1.c1", 2, 256, 64, 1, 1, 64},
                        {"l2.c1", 1, 256, 128, 1, 1, 64},  {"l2.c2", 1, 128, 128, 3, 2, 64},   {"l2.c3", 4, 128, 512, 1, 1, 32},  {"l2.ds", 1, 256, 512, 1, 2, 64},
                        {"l2.c1", 3, 512, 128, 1, 1, 32},  {"l2.c2", 3, 128, 128, 3, 1, 32},   {"l3.c1", 1, 512, 256, 1, 1, 32},  {"l3.c2", 1, 256, 256, 3, 2, 32},
                        {"l3.c3", 6, 256, 1024, 1, 1, 16}, {"l3.ds", 1, 512, 1024, 1, 2, 32},  {"l3.c1", 5, 1024, 256, 1, 1, 16}, {"l3.c2", 5, 256, 256, 3, 1, 16},
                        {"l4.c1", 1, 1024, 512, 1, 1, 16}, {"l4.c2", 1, 512, 512, 3, 2, 16},   {"l4.c3", 3, 512, 2048, 1, 1, 8},  {"l4.ds", 1, 1024, 2048, 1, 2, 16},
                        {"l4.c1", 2, 2048, 512, 1, 1, 8},  {"l4.c2", 2, 512, 512, 3, 1, 8}};
    printf("# ResNet-50 convolutions behind the stem, batch %d, 256x256 input; us per launch (HIP events around back-to-back launches, operands rotating through\n", batch);
    printf("# %zu MiB: nothing L2- or MALL-resident between launches); bound = max(2.5 PF, 8 TB/s on x + y + w); TF = algorithmic TFLOP/s of the forward\n", g_arena_bytes >> 20);
    printf("%-7s %2s %5s %5s %1s %1s %3s | %7s | %-13s | %-13s | %-13s\n", "layer", "n", "Cin", "Cout", "k", "s", "H", "GFLOP", "fwd us/bound", "dgrad us/bound", "wgrad us/bound");
    double tot[3] = {0, 0, 0}, tot_bound = 0, tot_flops = 0;
    for (const L& l : layers) {
        const int pad = l.k / 2, Ho = (l.H + 2 * pad - l.k) / l.s + 1;
        const size_t x_b = (size_t)batch * l.H * l.H * l.Cin * 2, w_b = (size_t)l.Cout * l.k * l.k * l.Cin * 2, y_b = (size_t)batch * Ho * Ho * l.Cout * 2;
        const size_t sums_b = 4 * 2 * (size_t)l.Cout * 4;
        const size_t set = ((x_b + 2 * w_b + 2 * y_b + x_b + sums_b + 4095) / 4096) * 4096;          // x | w | w_bwd | y | dy | dx | sums
        const size_t ws_b = std::max(epi_conv2d_workspace_bytes(batch, l.H, l.H, l.Cin, l.Cout, l.k, l.k, l.s, pad),
                                     epi_gemm_tn_workspace_bytes(batch * Ho * Ho, l.Cout, l.Cin, l.k * l.k));
        char* ws = g_arena;
        char* sets = g_arena + (ws_b + 4095) / 4096 * 4096;
        const int nset = (int)std::max<size_t>(1, std::min<size_t>(32, (g_arena_bytes - (sets - g_arena)) / set));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (unsigned short*)g_arena, g_arena_bytes / 2, 777u);
        int rc = 0;
        for (int i = 0; i < nset; ++i) {
            char* b = sets + (size_t)i * set;
            rc |= epi_conv2d_pack_weight_bwd(b + x_b, l.Cout, l.Cin, l.k, l.k, l.s, pad, b + x_b + w_b, 0);
            CK(hipMemsetAsync(b + 2 * x_b + 2 * w_b + 2 * y_b, 0, sums_b, 0));
        }
        double us[3];
        us[0] = time_launches(30, [&](int i) {
            char* b = sets + (size_t)(i % nset) * set;
            int done = 0;
            rc |= epi_conv2d_fwd(b, b + x_b, b + x_b + 2 * w_b, batch, l.H, l.H, l.Cin, l.Cout, l.k, l.k, l.s, pad, (float*)(b + 2 * x_b + 2 * w_b + 2 * y_b), &done, ws, ws_b, 0);
        });
        us[1] = time_launches(30, [&](int i) {
            char* b = sets + (size_t)(i % nset) * set;
            rc |= epi_conv2d_bwd_data(b + x_b + 2 * w_b + y_b, b + x_b + w_b, b + x_b + 2 * w_b + 2 * y_b, batch, l.H, l.H, l.Cin, l.Cout, l.k, l.k, l.s, pad, nullptr, ws, ws_b, 0);
        });
        us[2] = time_launches(30, [&](int i) {
            char* b = sets + (size_t)(i % nset) * set;
            rc |= epi_conv2d_bwd_weight(b, b + x_b + 2 * w_b + y_b, b + x_b, EPI_BF16, batch, l.H, l.H, l.Cin, l.Cout, l.k, l.k, l.s, pad, ws, ws_b, 0);
        });
        const double flops = 2.0 * batch * Ho * Ho * (double)l.Cout * l.Cin * l.k * l.k;
        const double bound = std::max(flops / 2.5e15, (double)(x_b + y_b + w_b) / 8e12) * 1e6;
        printf("%-7s %2d %5d %5d %1d %1d %3d | %7.2f | %6.1f /%5.1f | %6.1f /%5.1f | %6.1f /%5.1f   fwd %4.0f TF (%.2f of bound)%s\n", l.name, l.n, l.Cin, l.Cout, l.k, l.s, l.H,
               flops * 1e-9, us[0], bound, us[1], bound, us[2], bound, flops / us[0] * 1e-6, bound / us[0], rc ? "  rc != 0" : "");
        for (int k = 0; k < 3; ++k) tot[k] += l.n * us[k];
        tot_bound += l.n * bound;
        tot_flops += l.n * flops;
    }
    printf("# total ours   fwd %7.1f us  dgrad %7.1f us  wgrad %7.1f us  sum %7.1f us  -> %6.1f TFLOP/s over 3 x %.1f GFLOP\n", tot[0], tot[1], tot[2],
           tot[0] + tot[1] + tot[2], 3 * tot_flops / (tot[0] + tot[1] + tot[2]) * 1e-6, tot_flops * 1e-9);
    printf("# total bound  fwd %7.1f us  dgrad %7.1f us  wgrad %7.1f us  sum %7.1f us\n", tot_bound, tot_bound, tot_bound, 3 * tot_bound);
    printf("# (MIOpen / CK on the same shapes through PyTorch, round 3, profiles/r03_conv_layers_final.txt: fwd 1001 / dgrad 1544 / wgrad 1641 us with ~12 us of Python per call in it)\n");
}

// ---- round 5: the staggered 256 x 256 loop (tile 2) against the lock-step loop of rounds 1-4 (tile 5) and the 128 x 128 tile (tile 1) ----
// reference: one thread per output element, fp32 accumulation in k order
__global__ void ref_gemm_kernel(const unsigned short* __restrict__ A, int lda, const unsigned short* __restrict__ Bt, int ldb, float* __restrict__ C, int ldc,
                                int M, int N, int K, int m_lo, int n_lo, int m_cnt, int n_cnt) {
    const int n = n_lo + blockIdx.x * blockDim.x + threadIdx.x, m = m_lo + blockIdx.y;
    if (n >= n_lo + n_cnt || n >= N || m >= M) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += __uint_as_float((unsigned)A[(size_t)m * lda + k] << 16) * __uint_as_float((unsigned)Bt[(size_t)n * ldb + k] << 16);
    C[(size_t)(m - m_lo) * ldc + (n - n_lo)] = acc;
}
// err[0] = max |c - ref| (as uint bits of a non-negative float), err[1] = max |ref|, err[2] = count of |c - ref| > 0.01 * |ref| + 0.02
__global__ void cmp_kernel(const unsigned short* __restrict__ C, int ldc, const float* __restrict__ R, int ldr, int m_lo, int n_lo, int m_cnt, int n_cnt, unsigned int* err) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= n_cnt || m >= m_cnt) return;
    const float c = __uint_as_float((unsigned)C[(size_t)(m_lo + m) * ldc + n_lo + n] << 16), r = R[(size_t)m * ldr + n];
    const float d = fabsf(c - r);
    atomicMax(err, __float_as_uint(d));
    atomicMax(err + 1, __float_as_uint(fabsf(r)));
    if (d > 0.01f * fabsf(r) + 0.02f) atomicAdd(err + 2, 1u);
}
__global__ void checksum_kernel(const unsigned int* __restrict__ p, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (unsigned long long)p[i] * (i % 1021 + 1);
    atomicAdd(out, s);
}

static float bits_f(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
static void stag_battery() {
    const Shape shapes[] = {{"8192^3          ", 8192, 8192, 8192}, {"4096^3          ", 4096, 4096, 4096}, {"2048^3          ", 2048, 2048, 2048},
                            {"fin.dX 1088->256", 131072, 256, 1088}, {"dc1.dX 2048x2048x4096", 2048, 2048, 4096}, {"dc3 32768x256x1024", 32768, 256, 1024},
                            {"8192x256x2304   ", 8192, 256, 2304},  {"8192x1024x256   ", 8192, 1024, 256},   {"ragged 1000x520x456", 1000, 520, 456},
                            {"2048x512x2048   ", 2048, 512, 2048},  {"131072x256x64   ", 131072, 256, 64},    {"fin.fwd 256->1088", 131072, 1088, 256},
                            {"131072x64x256   ", 131072, 64, 256}};
    printf("# stag: epi_gemm_bf16, tile 2 = 256x256 staggered loop (round 5), tile 5 = 256x256 lock-step loop (rounds 1-4), tile 1 = 128x128, tile 0 = planner's choice\n");
    printf("# check: a 512 x 512 corner block + the LAST 256 x 256 block against an fp32 reference (bad = elements off by > 1 %% + 0.02); race screen: 6 reruns, checksum of C\n");
    unsigned int* err; CK(hipMalloc(&err, 16));
    unsigned long long* sum; CK(hipMalloc(&sum, 8));
    float* ref; CK(hipMalloc(&ref, 512 * 512 * 4));
    for (const Shape& s : shapes) {
        const int lda = s.K, ldb = s.K, ldc = s.N;
        const size_t a_bytes = (size_t)s.M * lda * 2, b_bytes = (size_t)s.N * ldb * 2, c_bytes = (size_t)s.M * ldc * 2;
        size_t ws_bytes = 0;
        for (int t : {0, 1, 2, 5}) { epi_gemm_tune(t, -1); ws_bytes = std::max(ws_bytes, epi_gemm_workspace_bytes(s.M, s.N, s.K, 1)); }
        const size_t set = ((a_bytes + b_bytes + c_bytes + 4095) / 4096) * 4096;
        char* ws = g_arena;
        char* sets = g_arena + (ws_bytes + 4095) / 4096 * 4096;
        if ((size_t)(sets - g_arena) + set > g_arena_bytes) { printf("%s (arena too small)\n", s.name); continue; }
        const int nfresh = (int)std::max<size_t>(1, std::min<size_t>(8, (g_arena_bytes - (sets - g_arena)) / set));
        hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (unsigned short*)g_arena, g_arena_bytes / 2, 4242u);
        for (int tile : {2, 5, 1, 0}) {
            epi_gemm_tune(tile, -1);
            int rc = 0;
            char* base = sets;
            unsigned short* A = (unsigned short*)base; unsigned short* B = (unsigned short*)(base + a_bytes); unsigned short* C = (unsigned short*)(base + a_bytes + b_bytes);
            // correctness on set 0
            CK(hipMemsetAsync(C, 0xff, c_bytes, 0));
            rc |= epi_gemm_bf16(A, lda, B, ldb, C, ldc, EPI_BF16, s.M, s.N, s.K, nullptr, ws, ws_bytes, 0);
            unsigned int herr[2][4] = {};
            for (int blk = 0; blk < 2; ++blk) {
                const int m_cnt = std::min(s.M, blk ? 256 : 512), n_cnt = std::min(s.N, blk ? 256 : 512);
                const int m_lo = blk ? s.M - m_cnt : 0, n_lo = blk ? s.N - n_cnt : 0;
                CK(hipMemsetAsync(err, 0, 16, 0));
                hipLaunchKernelGGL(ref_gemm_kernel, dim3((n_cnt + 63) / 64, m_cnt), dim3(64), 0, 0, A, lda, B, ldb, ref, 512, s.M, s.N, s.K, m_lo, n_lo, m_cnt, n_cnt);
                hipLaunchKernelGGL(cmp_kernel, dim3((n_cnt + 63) / 64, m_cnt), dim3(64), 0, 0, C, ldc, ref, 512, m_lo, n_lo, m_cnt, n_cnt, err);
                CK(hipMemcpy(herr[blk], err, 16, hipMemcpyDeviceToHost));
            }
            // race screen: identical reruns must give identical bits
            unsigned long long first = 0; int mismatches = 0;
            for (int rep = 0; rep < 6; ++rep) {
                if (rep) rc |= epi_gemm_bf16(A, lda, B, ldb, C, ldc, EPI_BF16, s.M, s.N, s.K, nullptr, ws, ws_bytes, 0);
                CK(hipMemsetAsync(sum, 0, 8, 0));
                hipLaunchKernelGGL(checksum_kernel, dim3(1024), dim3(256), 0, 0, (const unsigned int*)C, c_bytes / 4, sum);
                unsigned long long h; CK(hipMemcpy(&h, sum, 8, hipMemcpyDeviceToHost));
                if (!rep) first = h; else if (h != first) ++mismatches;
            }
            double us[2];
            for (int fresh = 1; fresh >= 0; --fresh) {
                const int nset = fresh ? nfresh : 1;
                us[fresh] = time_launches(s.M >= 8192 && s.N >= 8192 ? 10 : 30, [&](int i) {
                    char* b = sets + (size_t)(i % nset) * set;
                    rc |= epi_gemm_bf16(b, lda, b + a_bytes, ldb, b + a_bytes + b_bytes, ldc, EPI_BF16, s.M, s.N, s.K, nullptr, ws, ws_bytes, 0);
                });
            }
            const double flop = 2.0 * s.M * s.N * s.K;
            printf("%-22s tile %d  fresh %8.2f us (%7.1f TF)  warm %8.2f us (%7.1f TF)  err %.4f / %.4f of max|ref| %.2f  bad %u+%u  rerun mismatches %d  rc %d\n", s.name, tile, us[1],
                   flop / us[1] * 1e-6, us[0], flop / us[0] * 1e-6, bits_f(herr[0][0]), bits_f(herr[1][0]), bits_f(herr[0][1]), herr[0][2], herr[1][2],
                   mismatches, rc);
            fflush(stdout);
        }
        epi_gemm_tune(0, -1);
    }
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs\n", prop.name, prop.multiProcessorCount);
    g_arena_bytes = (size_t)1536 << 20;
    CK(hipMalloc(&g_arena, g_arena_bytes));
    CK(hipMalloc(&g_sink, 64));
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (unsigned short*)g_arena, g_arena_bytes / 2, 12345u);
    CK(hipDeviceSynchronize());
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&nt_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const char* what = argc > 1 ? argv[1] : "all";
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");
    if (!strcmp(what, "probe") || !strcmp(what, "all")) probe();
    if (!strcmp(what, "gemm") || !strcmp(what, "all")) gemm_battery(quick);
    if (!strcmp(what, "conv") || !strcmp(what, "all")) conv_battery();
    if (!strcmp(what, "layers")) layers_table(argc > 2 ? atoi(argv[2]) : 32);
    if (!strcmp(what, "stag")) stag_battery();
    return 0;
}
