// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the ResNet stem (reference pose3d_resnet.py:104,186), NHWC bf16.
//
// HBM-bound.  Forward: one read of x (B*H*W*C*2 bytes through L1/L2: every pixel belongs to at most 4 windows), one write of y and of a
// one-byte window position per output element (0 .. 8, scan order kh*3 + kw) -- instead of the int64 index tensor the library keeps
// (8 bytes per element) -- so that the backward pass never reads x again and needs no atomics and no zero fill: an input pixel (h, w)
// lies in <= 2 x 2 windows, it GATHERS dy from those whose recorded position is its own.  Backward traffic: dy + positions + one
// write of dx (batch 32, 64 x 128 x 128 -> 64 x 64: 16.8 + 8.4 + 67 MB against 67 MB zero fill + 67 MB indices + atomics in the library).
// Selection rule of the library kernel (max_pool_forward_nhwc): scan kh, kw ascending over the window clipped to the image, take a
// value when it is greater than the running maximum or NaN (so the first of equal maxima wins and a NaN sticks).
#include "common.h"

namespace epi {

// XCD-aware block order (8 XCDs with private L2s; the dispatcher places linear block b on XCD b % 8): every XCD gets a CONTIGUOUS range
// of logical blocks, so the input rows that neighbouring output rows share are fetched into one L2 instead of two (PMC: the forward
// read x 1.5 times and the backward dy + positions 2.1 times in linear block order, 1.02 and 1.00 times in this order;
// profiles/r02_pmc_traffic_summary.csv, _b_pool_xcd_order.csv)
__device__ __forceinline__ long long pool_block(long long lin, long long total) {
    const long long q = total >> 3, r = total & 7, xcd = lin & 7, idx = lin >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// one thread: 8 consecutive channels (16 bytes) of one output pixel.
// AFFINE: x is the RAW stem convolution output and every value read becomes bf16(relu(x * scale[c] + shift[c])) first -- the stem's BatchNorm +
// ReLU applied on the way into the pool (pose3d_resnet.py:186-188: `x = self.maxpool(self.relu(self.bn1(self.conv1(x))))`): the 67 MB
// normalised tensor, whose only reader is this pool, is never written.  The rounding to bf16 before the comparison makes the result the one the
// two-pass form gives, bit for bit.
template <bool AFFINE>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const unsigned short* __restrict__ x, const float* __restrict__ scale_shift,
                                                               unsigned short* __restrict__ y,
                                                               unsigned char* __restrict__ pos, int H, int W, int C8, int Ho, int Wo, long long total) {
    const long long t = pool_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c8 = (int)(t % C8);
    long long r = t / C8;
    const int ow = (int)(r % Wo); r /= Wo;
    const int oh = (int)(r % Ho);
    const long long n = r / Ho;
    float best[8];
    unsigned int bits[8], where[8];
    // the library starts from the first position inside the image (where a window of -inf values leaves its gradient)
    const unsigned int first = (unsigned int)((oh == 0 ? 1 : 0) * 3 + (ow == 0 ? 1 : 0));
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bits[k] = 0xff80u; where[k] = first; }
    float sc[8], sh[8];
    if (AFFINE) {
        const int C = C8 * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) { sc[k] = scale_shift[c8 * 8 + k]; sh[k] = scale_shift[C + c8 * 8 + k]; }
    }
    const unsigned short* xn = x + n * H * W * (long long)(C8 * 8) + c8 * 8;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * oh - 1 + kh;
        if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int w = 2 * ow - 1 + kw;
            if ((unsigned)w >= (unsigned)W) continue;
            const uint4v v = *reinterpret_cast<const uint4v*>(xn + ((long long)h * W + w) * (C8 * 8));
            const unsigned int wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned int b = (k & 1) ? (wd[k >> 1] >> 16) : (wd[k >> 1] & 0xffffu);
                float f = __uint_as_float(b << 16);
                if (AFFINE) {
                    b = f32_to_bf16(fmaxf(f * sc[k] + sh[k], 0.f));
                    f = __uint_as_float(b << 16);
                }
                if (f > best[k] || f != f) { best[k] = f; bits[k] = b; where[k] = kh * 3 + kw; }
            }
        }
    }
    uint4v o;
    o.x = bits[0] | (bits[1] << 16); o.y = bits[2] | (bits[3] << 16); o.z = bits[4] | (bits[5] << 16); o.w = bits[6] | (bits[7] << 16);
    *reinterpret_cast<uint4v*>(y + t * 8) = o;
    uint2 p;
    p.x = where[0] | (where[1] << 8) | (where[2] << 16) | (where[3] << 24);
    p.y = where[4] | (where[5] << 8) | (where[6] << 16) | (where[7] << 24);
    *reinterpret_cast<uint2*>(pos + t * 8) = p;
}

// one thread: 8 consecutive channels of one INPUT pixel; sums (fp32, rounded once) dy of the windows that selected it
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ pos,
                                                               unsigned short* __restrict__ dx, int H, int W, int C8, int Ho, int Wo, long long total) {
    const long long t = pool_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c8 = (int)(t % C8);
    long long r = t / C8;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const long long n = r / H;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    // windows oh with 2*oh - 1 <= h <= 2*oh + 1: h even -> oh = h/2; h odd -> (h-1)/2 and (h+1)/2
    const int oh0 = h >> 1, oh1 = (h + 1) >> 1, ow0 = w >> 1, ow1 = (w + 1) >> 1;
    for (int oh = oh0; oh <= oh1; ++oh) {
        if (oh >= Ho) continue;
        const int kh = h - (2 * oh - 1);
        for (int ow = ow0; ow <= ow1; ++ow) {
            if (ow >= Wo) continue;
            const unsigned int me = (unsigned int)(kh * 3 + (w - (2 * ow - 1)));
            const long long o = ((n * Ho + oh) * Wo + ow) * (long long)C8 + c8;
            const uint2 p = *reinterpret_cast<const uint2*>(pos + o * 8);
            // any of the 8 channels selected here?  (most windows select another pixel: skip the dy load)
            const unsigned int m4 = me * 0x01010101u;
            const unsigned int e0 = p.x ^ m4, e1 = p.y ^ m4;
            const bool any = (((e0 - 0x01010101u) & ~e0) | ((e1 - 0x01010101u) & ~e1)) & 0x80808080u;       // a zero byte in e0 / e1
            if (!any) continue;
            const uint4v g = *reinterpret_cast<const uint4v*>(dy + o * 8);
            const unsigned int gd[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned int pk = ((k < 4 ? p.x : p.y) >> (8 * (k & 3))) & 0xffu;
                const unsigned int b = (k & 1) ? (gd[k >> 1] & 0xffff0000u) : (gd[k >> 1] << 16);
                if (pk == me) acc[k] += __uint_as_float(b);
            }
        }
    }
    uint4v o4;
    o4.x = pack_bf16x2(acc[0], acc[1]); o4.y = pack_bf16x2(acc[2], acc[3]); o4.z = pack_bf16x2(acc[4], acc[5]); o4.w = pack_bf16x2(acc[6], acc[7]);
    *reinterpret_cast<uint4v*>(dx + t * 8) = o4;
}

}  // namespace epi

static inline int pool_out(int H) { return (H + 2 - 3) / 2 + 1; }

extern "C" int epi_maxpool3x3s2_fwd(const void* x, void* y, void* pos, int B, int H, int W, int C, epi_stream_t stream) {
    if (!x || !y || !pos || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (C % 8 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) || (reinterpret_cast<uintptr_t>(pos) & 7u)) return EPI_ERR_UNSUPPORTED;
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long long total = (long long)B * Ho * Wo * (C / 8);
    if ((total + 255) / 256 > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::maxpool3x3s2_fwd_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, (const float*)nullptr, (unsigned short*)y, (unsigned char*)pos, H, W, C / 8, Ho, Wo, total);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

// y = maxpool(bf16(relu(x * scale + shift))): scale_shift [2C] f32 = the `scale_shift` epi_bn_finalize (or epi_bn_act_fwd) wrote
extern "C" int epi_maxpool3x3s2_bn_relu_fwd(const void* x, const float* scale_shift, void* y, void* pos, int B, int H, int W, int C, epi_stream_t stream) {
    if (!x || !scale_shift || !y || !pos || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (C % 8 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) || (reinterpret_cast<uintptr_t>(pos) & 7u)) return EPI_ERR_UNSUPPORTED;
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long long total = (long long)B * Ho * Wo * (C / 8);
    if ((total + 255) / 256 > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::maxpool3x3s2_fwd_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, scale_shift, (unsigned short*)y, (unsigned char*)pos, H, W, C / 8, Ho, Wo, total);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_maxpool3x3s2_bwd(const void* dy, const void* pos, void* dx, int B, int H, int W, int C, epi_stream_t stream) {
    if (!dy || !pos || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (C % 8 || ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15u) || (reinterpret_cast<uintptr_t>(pos) & 7u)) return EPI_ERR_UNSUPPORTED;
    const int Ho = pool_out(H), Wo = pool_out(W);
    const long long total = (long long)B * H * W * (C / 8);
    if ((total + 255) / 256 > 0x7fffffffLL) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::maxpool3x3s2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)dy, (const unsigned char*)pos, (unsigned short*)dx, H, W, C / 8, Ho, Wo, total);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
