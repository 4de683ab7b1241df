// Phase timing of conv_patch_kernel (s_memtime stamps): build on the GPU box with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEPI_PATCH_TRACE -Iinclude tools/patch_trace.cpp epipolarpose_amd/csrc/*.hip -o /tmp/patch_trace
// usage: patch_trace B H W Cin Cout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "epipolar_hip.h"
extern "C" int epi_patch_trace_read(unsigned long long* out, int clear);
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, H = argc > 2 ? atoi(argv[2]) : 16, W = argc > 3 ? atoi(argv[3]) : 16;
    const int Cin = argc > 4 ? atoi(argv[4]) : 256, Cout = argc > 5 ? atoi(argv[5]) : 256;
    const size_t nx = (size_t)B * H * W * Cin, nw = (size_t)Cout * 9 * Cin, ny = (size_t)B * H * W * Cout;
    void *x, *w, *y, *ws;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2);
    std::vector<unsigned short> hx(nx), hw(nw);
    for (size_t i = 0; i < nx; ++i) hx[i] = 0x3c00 + (i * 7919 % 128);
    for (size_t i = 0; i < nw; ++i) hw[i] = 0x3a00 + (i * 104729 % 128);
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    const size_t wsb = epi_conv2d_workspace_bytes(B, H, W, Cin, Cout, 3, 3, 1, 1);
    hipMalloc(&ws, wsb ? wsb : 16);
    std::vector<unsigned long long> tr(64 * 96);
    for (int it = 0; it < 4; ++it) {
        epi_patch_trace_read(tr.data(), 1);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        const int rc = epi_conv2d_fwd(x, w, y, B, H, W, Cin, Cout, 3, 3, 1, 1, nullptr, nullptr, ws, wsb, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("rc %d\n", rc); return 1; }
        epi_patch_trace_read(tr.data(), 0);
        if (it < 3) continue;
        printf("B %d H %d W %d Cin %d Cout %d: launch(es) %.1f us; stamps in s_memtime ticks relative to the workgroup's start\n", B, H, W, Cin, Cout, ms * 1e3);
        unsigned long long first = ~0ULL;
        for (int g = 0; g < 64; ++g) if (tr[g * 96] && tr[g * 96] < first) first = tr[g * 96];
        for (int g = 0; g < 64; g += 5) {
            if (!tr[g * 96]) continue;
            printf("wg %4d start +%6llu |", g * 16, tr[g * 96] - first);
            for (int i = 1; i < 96 && tr[g * 96 + i]; ++i) printf(" %llu", tr[g * 96 + i] - tr[g * 96 + i - 1]);
            printf("\n");
        }
    }
    return 0;
}
