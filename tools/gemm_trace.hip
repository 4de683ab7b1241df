// Phase trace of the NT implicit-GEMM kernel (gfx950): links the EPI_GEMM_TRACE build of the library (tools/build_trace_lib.sh), launches one
// ResNet-50 layer shape at a time on fresh operands and prints, over ALL workgroups of the last launch, how long each phase took:
//   prologue (entry -> staging roles computed) | first K tile landed | K loop | tile parked in LDS | stores issued | stores drained
// plus the dispatch picture: when workgroups started and ended relative to the first start (s_memrealtime, 100 MHz), how many CUs ran them.
// This is the measurement VERDICT r03 asked for ("locate the ~5 us of per-launch fixed cost").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <set>
#include <map>
#include <algorithm>
#include "../include/epipolar_hip.h"

extern "C" int epi_gemm_trace_read(unsigned long long* out, int clear);
extern "C" int epi_gemm_tune(int tile, int pipe);
extern "C" int epi_gemm_store_policy(int v);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const float v = ((int)(h & 0xffff) - 32768) * (1.0f / 65536.0f);
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

constexpr int WGS = 8192, SLOTS = 16;
static std::vector<unsigned long long> g_trace((size_t)WGS* SLOTS);

static double pct(std::vector<double> v, double q) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)(q * (v.size() - 1) + 0.5))];
}

static void report(const char* name, double us_event) {
    if (epi_gemm_trace_read(g_trace.data(), 1)) { printf("trace read failed\n"); return; }
    // workgroups that stamped
    std::vector<int> wgs;
    for (int w = 0; w < WGS; ++w) if (g_trace[(size_t)w * SLOTS + 0] && g_trace[(size_t)w * SLOTS + 8]) wgs.push_back(w);
    if (wgs.empty()) { printf("%s: no stamps (kernel without trace points, e.g. A-stationary / patch)\n", name); return; }
    unsigned long long rt0 = ~0ull, rt1 = 0;
    for (int w : wgs) { rt0 = std::min(rt0, g_trace[(size_t)w * SLOTS + 1]); rt1 = std::max(rt1, g_trace[(size_t)w * SLOTS + 8]); }
    // shader clock per 100 MHz tick from the longest-lived workgroup
    double clk_per_tick = 0; unsigned long long best = 0;
    for (int w : wgs) {
        const unsigned long long* t = &g_trace[(size_t)w * SLOTS];
        if (t[8] - t[1] > best && t[7] > t[0]) { best = t[8] - t[1]; clk_per_tick = (double)(t[7] - t[0]) / (double)(t[8] - t[1]); }
    }
    const double us_per_clk = clk_per_tick > 0 ? 0.01 / clk_per_tick : 1.0 / 2400.0;
    std::vector<double> ph[6], start, end, life;
    std::set<unsigned long long> cus;
    std::map<unsigned long long, int> per_cu;
    for (int w : wgs) {
        const unsigned long long* t = &g_trace[(size_t)w * SLOTS];
        const int idx[7] = {0, 2, 3, 4, 5, 6, 7};
        for (int k = 0; k < 6; ++k) ph[k].push_back(t[idx[k + 1]] >= t[idx[k]] && t[idx[k + 1]] ? (double)(t[idx[k + 1]] - t[idx[k]]) * us_per_clk : 0.0);
        start.push_back((double)(t[1] - rt0) * 0.01);
        end.push_back((double)(t[8] - rt0) * 0.01);
        life.push_back((double)(t[8] - t[1]) * 0.01);
        const unsigned long long cu = (t[10] << 32) | (t[9] & 0xffffff00ull);
        cus.insert(cu);
        per_cu[cu]++;
    }
    int max_per_cu = 0;
    for (auto& kv : per_cu) max_per_cu = std::max(max_per_cu, kv.second);
    int late = 0;
    for (double s : start) if (s > 2.0) ++late;
    printf("%s\n", name);
    printf("    event %.2f us/launch | first start -> last end %.2f us | %zu workgroups (K tiles %llu) on %zu CUs (max %d per CU), %d started > 2 us after the first | %.2f GHz\n",
           us_event, (double)(rt1 - rt0) * 0.01, wgs.size(), g_trace[(size_t)wgs[0] * SLOTS + 11], cus.size(), max_per_cu, late, 1e-3 / us_per_clk);
    printf("    start  p50 %.2f  p90 %.2f  max %.2f us | end  p10 %.2f  p50 %.2f  max %.2f us | lifetime  p50 %.2f  max %.2f us\n", pct(start, .5), pct(start, .9),
           pct(start, 1), pct(end, .1), pct(end, .5), pct(end, 1), pct(life, .5), pct(life, 1));
    const char* names[6] = {"prologue      ", "first tile    ", "K loop        ", "park in LDS   ", "stores issued ", "stores drained"};
    for (int k = 0; k < 6; ++k) printf("    %s  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us\n", names[k], pct(ph[k], .1), pct(ph[k], .5), pct(ph[k], .9), pct(ph[k], 1));
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs; phases in us (s_memtime scaled by s_memrealtime), per workgroup, wave 0\n", prop.name, prop.multiProcessorCount);
    const size_t arena_bytes = (size_t)1536 << 20;
    char* arena = nullptr;
    CK(hipMalloc(&arena, arena_bytes));
    struct Shape { const char* name; int M, N, K; };
    const Shape shapes[] = {{"l1.c1  256->64   M=131072", 131072, 64, 256}, {"l1.c3  64->256   M=131072", 131072, 256, 64}, {"l2.c1  512->128  M=32768", 32768, 128, 512},
                            {"l2.c3  128->512  M=32768", 32768, 512, 128},  {"l3.c1  1024->256 M=8192", 8192, 256, 1024},   {"l3.c3  256->1024 M=8192", 8192, 1024, 256},
                            {"l4.c1  2048->512 M=2048", 2048, 512, 2048},   {"l4.c3  512->2048 M=2048", 2048, 2048, 512}};
    struct Var { const char* name; int tile, pipe, store; };
    const Var vars[] = {{"default", 0, 0, 0}, {"small (forced 128^2)", 1, 0, 0}, {"pipe2", 0, 2, 0}, {"half", 3, 0, 0}, {"stores nt", 0, 0, 1}, {"stores sc1", 0, 0, 2}};
    for (const Shape& s : shapes)
        for (const Var& v : vars) {

            const int lda = s.K, ldb = s.K, ldc = s.N;
            const size_t a_bytes = (size_t)s.M * lda * 2, b_bytes = (size_t)s.N * ldb * 2, c_bytes = (size_t)s.M * ldc * 2;
            const size_t ws_bytes = epi_gemm_workspace_bytes(s.M, s.N, s.K, 1);
            const size_t set = ((a_bytes + b_bytes + c_bytes + 4095) / 4096) * 4096;
            char* ws = arena;
            char* sets = arena + (ws_bytes + 4095) / 4096 * 4096;
            const int nset = (int)std::max<size_t>(1, std::min<size_t>(48, (arena_bytes - (sets - arena)) / set));
            hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, (unsigned short*)arena, arena_bytes / 2, 12345u);
            epi_gemm_tune(v.tile, v.pipe);
            epi_gemm_store_policy(v.store);
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            int rc = 0;
            const int reps = 12;
            for (int i = 0; i < 3; ++i) {
                char* base = sets + (size_t)(i % nset) * set;
                rc |= epi_gemm_bf16(base, lda, base + a_bytes, ldb, base + a_bytes + b_bytes, ldc, EPI_BF16, s.M, s.N, s.K, nullptr, ws, ws_bytes, 0);
            }
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) {
                char* base = sets + (size_t)((i + 3) % nset) * set;
                rc |= epi_gemm_bf16(base, lda, base + a_bytes, ldb, base + a_bytes + b_bytes, ldc, EPI_BF16, s.M, s.N, s.K, nullptr, ws, ws_bytes, 0);
            }
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            // the stamps of the LAST launch of the run are what the trace buffer holds now (every launch overwrites them)
            char label[160];
            snprintf(label, sizeof label, "%s  [%s]  rc %d", s.name, v.name, rc);
            report(label, ms * 1e3 / reps);
            CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        }
    return 0;
}
