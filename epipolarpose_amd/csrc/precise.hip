// fp32-grade verification mode (epipolarpose_amd/models/precise.py): the pieces that let the network run on fp32 activations through the
// SAME bf16 MFMA GEMM kernels as the training path.
//
// The training path is bf16 x bf16 -> fp32 on v_mfma_f32_32x32x16_bf16 with bf16 activations in HBM, so against the reference's fp32
// network (lib/models/pose3d_resnet.py:185-201) it can only be held to a bf16 yardstick.  This mode keeps every activation in fp32 and
// feeds each GEMM with SPLIT operands: x = hi + lo (+ lo2) with hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo) -- 8 significant
// bits each, so two pieces carry x to 2^-18 and three to 2^-26 relative -- laid out along the GEMM's reduction dimension:
//     conv(x, w) ~= conv([x_hi | x_lo | x_hi], [w_hi | w_hi | w_lo])          (channel blocks: the convolution sums over channels)
//     dW         ~= sum over the batch of [x_hi ; x_lo ; x_hi] (x) [dy_hi ; dy_hi ; dy_lo]   (row blocks: the weight gradient sums over rows)
// i.e. one call of the unchanged kernel on tensors with 3 (or 6) times the reduction extent, fp32 accumulation, fp32 result.
// Nothing here is on the training hot path.
#include "common.h"

namespace epi {

struct SplitPattern { int nblk; int piece[8]; };      // piece: 0 hi, 1 lo, 2 lo2

// x [rows][C] f32 -> out bf16:  row_concat == 0: [rows][nblk * C], block b of a row = piece[b] of that row's C values (channel blocks);
//                                row_concat == 1: [nblk][rows][C] (row blocks).  One thread: 4 consecutive values of one row.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, long long rows, int C, SplitPattern pat, int row_concat,
                                                         unsigned short* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq = C >> 2;
    if (t >= rows * cq) return;
    const long long r = t / cq;
    const int c = (int)(t - r * cq) * 4;
    const float4v v = *reinterpret_cast<const float4v*>(x + r * C + c);
    const float in[4] = {v.x, v.y, v.z, v.w};
    float pc[3][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float hi = bf16_to_f32(f32_to_bf16(in[k]));
        const float r1 = in[k] - hi;                       // exact: hi holds the leading 8 bits of x
        const float lo = bf16_to_f32(f32_to_bf16(r1));
        const float lo2 = bf16_to_f32(f32_to_bf16(r1 - lo));
        pc[0][k] = hi; pc[1][k] = lo; pc[2][k] = lo2;
    }
    for (int b = 0; b < pat.nblk; ++b) {
        const int p = pat.piece[b];
        uint2 o;
        o.x = pack_bf16x2(pc[p][0], pc[p][1]);
        o.y = pack_bf16x2(pc[p][2], pc[p][3]);
        const long long dst = row_concat ? ((long long)b * rows + r) * C + c : (r * pat.nblk + b) * (long long)C + c;
        *reinterpret_cast<uint2*>(out + dst) = o;
    }
}

// MaxPool2d(3, 2, 1) on fp32 NHWC activations: the selection rule and the one-byte window positions of csrc/pool.hip; one thread = 4 channels
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_f32_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ pos,
                                                                   int H, int W, int C4, int Ho, int Wo, long long total) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c4 = (int)(t % C4);
    long long r = t / C4;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho);
    const long long n = r / Ho;
    float best[4];
    unsigned int where[4];
    const unsigned int first = (unsigned int)((oh == 0 ? 1 : 0) * 3 + (ow == 0 ? 1 : 0));
#pragma unroll
    for (int k = 0; k < 4; ++k) { best[k] = -INFINITY; where[k] = first; }
    const float* xn = x + n * H * W * (long long)(C4 * 4) + c4 * 4;
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
        if ((unsigned)h >= (unsigned)H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * ow - 1 + kw;
            if ((unsigned)w >= (unsigned)W) continue;
            const float4v v = *reinterpret_cast<const float4v*>(xn + ((long long)h * W + w) * (C4 * 4));
            const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (f[k] > best[k] || f[k] != f[k]) { best[k] = f[k]; where[k] = kh * 3 + kw; }
        }
    }
    float4v o; o.x = best[0]; o.y = best[1]; o.z = best[2]; o.w = best[3];
    *reinterpret_cast<float4v*>(y + t * 4) = o;
    *reinterpret_cast<unsigned int*>(pos + t * 4) = where[0] | (where[1] << 8) | (where[2] << 16) | (where[3] << 24);
}

__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_f32_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ pos, float* __restrict__ dx,
                                                                   int H, int W, int C4, int Ho, int Wo, long long total) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c4 = (int)(t % C4);
    long long r = t / C4;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const long long n = r / H;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int oh0 = h >> 1, oh1 = (h + 1) >> 1, ow0 = w >> 1, ow1 = (w + 1) >> 1;
    for (int oh = oh0; oh <= oh1; ++oh) {
        if (oh >= Ho) continue;
        const int kh = h - (2 * oh - 1);
        for (int ow = ow0; ow <= ow1; ++ow) {
            if (ow >= Wo) continue;
            const unsigned int me = (unsigned int)(kh * 3 + (w - (2 * ow - 1)));
            const long long o = ((n * Ho + oh) * Wo + ow) * (long long)C4 + c4;
            const unsigned int p = *reinterpret_cast<const unsigned int*>(pos + o * 4);
            const float4v g = *reinterpret_cast<const float4v*>(dy + o * 4);
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (((p >> (8 * k)) & 0xffu) == me) acc[k] += gv[k];
        }
    }
    float4v o4; o4.x = acc[0]; o4.y = acc[1]; o4.z = acc[2]; o4.w = acc[3];
    *reinterpret_cast<float4v*>(dx + t * 4) = o4;
}

}  // namespace epi

// pieces: nblk entries in {0 hi, 1 lo, 2 lo2}; C % 4 == 0; out: nblk * rows * C bf16 values
extern "C" int epi_split_bf16(const float* x, long long rows, int C, const int* pieces, int nblk, int row_concat, void* out, epi_stream_t stream) {
    if (!x || !out || !pieces || rows <= 0 || C <= 0 || nblk <= 0 || nblk > 8) return EPI_ERR_INVALID_ARGUMENT;
    if (C % 4 || (reinterpret_cast<uintptr_t>(x) & 15u) || (reinterpret_cast<uintptr_t>(out) & 7u)) return EPI_ERR_UNSUPPORTED;
    epi::SplitPattern pat = {};
    pat.nblk = nblk;
    for (int b = 0; b < nblk; ++b) {
        if (pieces[b] < 0 || pieces[b] > 2) return EPI_ERR_INVALID_ARGUMENT;
        pat.piece[b] = pieces[b];
    }
    const long long total = rows * (C / 4);
    if ((total + 255) / 256 > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::split_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, rows, C, pat, row_concat ? 1 : 0,
                       (unsigned short*)out);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_maxpool3x3s2_fwd_f32(const void* x, void* y, void* pos, int B, int H, int W, int C, epi_stream_t stream) {
    if (!x || !y || !pos || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (C % 4 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) || (reinterpret_cast<uintptr_t>(pos) & 3u)) return EPI_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    if ((total + 255) / 256 > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::maxpool3x3s2_fwd_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y,
                       (unsigned char*)pos, H, W, C / 4, Ho, Wo, total);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_maxpool3x3s2_bwd_f32(const void* dy, const void* pos, void* dx, int B, int H, int W, int C, epi_stream_t stream) {
    if (!dy || !pos || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (C % 4 || ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15u) || (reinterpret_cast<uintptr_t>(pos) & 3u)) return EPI_ERR_UNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)B * H * W * (C / 4);
    if ((total + 255) / 256 > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::maxpool3x3s2_bwd_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)dy,
                       (const unsigned char*)pos, (float*)dx, H, W, C / 4, Ho, Wo, total);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
