// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  lds[i] = i; lane l points at elements 4l..4l+3 (8 bytes).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    int l = threadIdx.x;
    int elem = (mode == 0) ? 4 * l : ((l & 15) * 64 + (l >> 4) * 4);   // mode 1: lane(l&15) -> row (l&15) of a [16][64] tile, 4 elems at col (l>>4)*4
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(&lds[elem]));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    return 0;
}
