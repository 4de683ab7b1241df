// Input pipeline on the GPU: crop + scale + rotate a person patch out of a decoded camera frame and
// normalise it -- one byte-streaming kernel for what the reference does per sample on the host with OpenCV and NumPy
// (lib/utils/img_utils.py:114-127 generate_patch_image_cv -> cv2.warpAffine(INTER_LINEAR), :265-279 BGR -> RGB, colour
// scaling, clip, mean / std normalisation; lib/dataset/JointIntegralDataset.py:67-68 the ImageNet mean / std in 0..255 units).
//
// cv2.warpAffine semantics restated (OpenCV 4.1 imgwarp.cpp, the pin of the reference's environment.yml; cv2 itself is not
// installed -> parity unpinned): the forward map M is inverted in double; destination pixel (x, y) reads the
// source position in FIXED POINT, X = (round((M10*y + M12) * 1024) + 16 + round(M00*x * 1024)) >> 5 (5 fractional bits), samples
// the 2 x 2 neighbourhood with integer weights of scale 2^15 (rounded products of the two linear weights, corrected so that the
// four sum to 2^15), adds 2^14 and shifts by 15; taps outside the frame contribute 0 (BORDER_CONSTANT).
#include "common.h"

namespace epi {

struct PatchArgs {
    const unsigned char* frames;       // BGR bytes
    const long long* frame_offset;     // [B] byte offset of sample b's frame
    const int* frame_hw;               // [B][2] (height, width)
    const double* trans;               // [B][2][3] forward affine (frame -> patch), as gen_trans_from_patch_cv returns it
    const int* do_flip;                // [B] or null: mirror the frame first (generate_patch_image_cv:120-122)
    const float* color_scale;          // [B][3] (per RGB output channel) or null
    float mean[3], inv_std_is_std[3];  // per RGB channel; std (the kernel divides, as the reference does)
    int normalize;
    void* out;
    int out_bf16, out_nhwc;
    int B, PH, PW;
    // synthetic occlusion (lib/utils/augmentation.py:61-114), applied to the uint8 RGB patch before the colour stage (img_utils.py:271-272)
    const unsigned char* occ_bank;     // RGBA bytes of every occluder image, or null
    const long long* occ_offset;       // [N] byte offset of occluder n
    const int* occ_hw;                 // [N][2] (height, width) at native size
    const int* occ_place;              // [B][max_occ][5]: occluder index (-1: end of the sample's list), pasted width, height, x0, y0 (top-left in the patch)
    int max_occ;
};

// One pixel of occluder image `src` [sh][sw][4] resized DOWN to (dw, dh) <= (sw, sh) by box-filter averaging (cv2.resize INTER_AREA restated with
// exact integer arithmetic; the test oracle restates the same rule): destination pixel (px, py) covers [px*sw, (px+1)*sw) x [py*sh, (py+1)*sh) in
// units of 1/dw x 1/dh source pixels.
__device__ __forceinline__ void occluder_pixel(const unsigned char* __restrict__ src, int sh, int sw, int dh, int dw, int px, int py, int (&rgba)[4]) {
    if (dw == sw && dh == sh) {
        const unsigned char* q = src + ((long long)py * sw + px) * 4;
        rgba[0] = q[0]; rgba[1] = q[1]; rgba[2] = q[2]; rgba[3] = q[3];
        return;
    }
    long long acc[4] = {0, 0, 0, 0};
    const int ylo = py * sh, yhi = ylo + sh, xlo = px * sw, xhi = xlo + sw;
    for (int ky = ylo / dh; ky <= (yhi - 1) / dh; ++ky) {
        const int wy = min(yhi, (ky + 1) * dh) - max(ylo, ky * dh);
        for (int kx = xlo / dw; kx <= (xhi - 1) / dw; ++kx) {
            const int w = wy * (min(xhi, (kx + 1) * dw) - max(xlo, kx * dw));
            const unsigned char* q = src + ((long long)ky * sw + kx) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] += (long long)w * q[c];
        }
    }
    const long long den = (long long)sh * sw;
#pragma unroll
    for (int c = 0; c < 4; ++c) rgba[c] = (int)((2 * acc[c] + den) / (2 * den));
}

// One pixel of `src` resized UP to (dw, dh) >= (sw, sh): cv2.resize INTER_LINEAR on uint8 restated (OpenCV 4.1 resize.cpp: the coordinate tables of
// cv::resize, HResizeLinear<uchar, int, short, 2048>, the 8-bit VResizeLinear).  x: fx = float((px + 0.5) * (1 / (dw / sw)) - 0.5), sx = floor(fx), fx -= sx,
// (sx, fx) = (0, 0) left of the image and (sw - 1, 0) from its last column on; y: no border rule, the two rows are clamped into the image instead;
// weights short(round-half-even(w * 2048)) of the float32 values; ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
// augmentation.py:122 -- patches larger than 256 px (im_scale_factor > 1: the 384 px configuration).
__device__ __forceinline__ void occluder_pixel_linear(const unsigned char* __restrict__ src, int sh, int sw, int dh, int dw, int px, int py, int (&rgba)[4]) {
    float fx = (float)(((double)px + 0.5) * (1.0 / ((double)dw / (double)sw)) - 0.5);     // scale = 1. / inv_scale as cv::resize computes it
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= sw - 1) { sx = sw - 1; fx = 0.f; }
    float fy = (float)(((double)py + 0.5) * (1.0 / ((double)dh / (double)sh)) - 0.5);
    const int sy = (int)floorf(fy);
    fy -= (float)sy;
    const int a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, fx), 2048.f)), a1 = (int)rintf(__fmul_rn(fx, 2048.f));
    const int b0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, fy), 2048.f)), b1 = (int)rintf(__fmul_rn(fy, 2048.f));
    const int x1 = min(sx + 1, sw - 1), y0 = min(max(sy, 0), sh - 1), y1 = min(max(sy + 1, 0), sh - 1);
    const unsigned char* q00 = src + ((long long)y0 * sw + sx) * 4;
    const unsigned char* q01 = src + ((long long)y0 * sw + x1) * 4;
    const unsigned char* q10 = src + ((long long)y1 * sw + sx) * 4;
    const unsigned char* q11 = src + ((long long)y1 * sw + x1) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r0 = q00[c] * a0 + q01[c] * a1, r1 = q10[c] * a0 + q11[c] * a1;
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        rgba[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
}

__device__ __forceinline__ int round_half_even(double v) { return (int)__double2ll_rn(v); }   // cvRound / saturate_cast<int>(double)

// integer bilinear weights of OpenCV's table (initInterTab2D, INTER_LINEAR): products of the float32 1-D weights scaled by 2^15,
// rounded to nearest-even, the largest (smallest) entry corrected when the sum misses 2^15
__device__ __forceinline__ void bilinear_itab(int ax, int ay, int (&w)[4]) {
    const float fx = (float)ax * (1.f / 32.f), fy = (float)ay * (1.f / 32.f);
    const float tx[2] = {1.f - fx, fx}, ty[2] = {1.f - fy, fy};
    int isum = 0;
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const float v = ty[k1] * tx[k2];
            int iv = (int)__float2ll_rn(v * 32768.f);
            iv = iv > 32767 ? 32767 : (iv < -32768 ? -32768 : iv);       // saturate_cast<short>
            w[k1 * 2 + k2] = iv;
            isum += iv;
        }
    if (isum != 32768) {
        const int diff = isum - 32768;
        int mk = 0;      // search window of OpenCV: ksize2 = 1 -> rows/cols [1, 2) -> only entry (1,1)? no: for ksize 2 the window is k in [ksize/2, ksize/2+... ) = the 2x2 block itself
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            if (diff < 0 ? w[k] > w[mk] : w[k] < w[mk]) mk = k;
        }
        w[mk] -= diff;
    }
}

__global__ __launch_bounds__(256) void patch_crop_kernel(PatchArgs p) {
    const int b = blockIdx.y;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= p.PH * p.PW) return;
    const int y = pix / p.PW, x = pix - y * p.PW;
    const double* m = p.trans + (long long)b * 6;
    // invert the forward map exactly as cv::warpAffine does (double)
    double M0 = m[0], M1 = m[1], M2 = m[2], M3 = m[3], M4 = m[4], M5 = m[5];
    double D = M0 * M4 - M1 * M3;
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M4 * D, A22 = M0 * D;
    M0 = A11; M1 *= -D; M3 *= -D; M4 = A22;
    const double b1 = -M0 * M2 - M1 * M5, b2 = -M3 * M2 - M4 * M5;
    M2 = b1; M5 = b2;
    const int adelta = round_half_even(M0 * x * 1024.0), bdelta = round_half_even(M3 * x * 1024.0);
    const int X0 = round_half_even((M1 * y + M2) * 1024.0) + 16, Y0 = round_half_even((M4 * y + M5) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = X >> 5, sy = Y >> 5, ax = X & 31, ay = Y & 31;
    int w[4];
    bilinear_itab(ax, ay, w);
    const int H = p.frame_hw[2 * b], W = p.frame_hw[2 * b + 1];
    const unsigned char* img = p.frames + p.frame_offset[b];
    const bool flip = p.do_flip && p.do_flip[b];
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int k1 = 0; k1 < 2; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int yy = sy + k1, xx = sx + k2;
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                const int xs = flip ? W - 1 - xx : xx;
                const unsigned char* px = img + ((long long)yy * W + xs) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += (int)px[c] * w[k1 * 2 + k2];
            }
        }
    int rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {            // output channel c = RGB; source channel 2 - c (frames are BGR: img_utils.py:269)
        const int v = (acc[2 - c] + (1 << 14)) >> 15;
        rgb[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
    if (p.occ_bank) {
        // paste_over (augmentation.py:84-114), occluder after occluder: float32 alpha blend, the assignment into the uint8 image truncates.
        // Separate roundings for every product and sum, as NumPy evaluates them (no fused multiply-add).
        const int* pl = p.occ_place + (long long)b * p.max_occ * 5;
        for (int k = 0; k < p.max_occ; ++k) {
            const int idx = pl[5 * k];
            if (idx < 0) break;
            const int dw = pl[5 * k + 1], dh = pl[5 * k + 2], ox = x - pl[5 * k + 3], oy = y - pl[5 * k + 4];
            if ((unsigned)ox >= (unsigned)dw || (unsigned)oy >= (unsigned)dh) continue;
            int rgba[4];
            const int sh = p.occ_hw[2 * idx], sw = p.occ_hw[2 * idx + 1];
            if (dw > sw || dh > sh) occluder_pixel_linear(p.occ_bank + p.occ_offset[idx], sh, sw, dh, dw, ox, oy, rgba);          // factor > 1: INTER_LINEAR (:122)
            else occluder_pixel(p.occ_bank + p.occ_offset[idx], sh, sw, dh, dw, ox, oy, rgba);                                     // INTER_AREA
            const float alpha = __fdiv_rn((float)rgba[3], 255.f), rest = __fsub_rn(1.f, alpha);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t1 = alpha * (float)rgba[c], t2 = rest * (float)rgb[c];
                asm volatile("" : "+v"(t1), "+v"(t2));              // two rounded products, then a rounded sum: keep the compiler from fusing them
                rgb[c] = (int)(t1 + t2);
            }
        }
    }
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float f = (float)rgb[c];
        if (p.color_scale) f = fminf(fmaxf(f * p.color_scale[3 * b + c], 0.f), 255.f);      // img_utils.py:276
        if (p.normalize) f = (f - p.mean[c]) / p.inv_std_is_std[c];                          // :277-278
        o[c] = f;
    }
    const long long plane = (long long)p.PH * p.PW;
    if (p.out_nhwc) {
        const long long base = ((long long)b * plane + pix) * 3;
        if (p.out_bf16) { unsigned short* q = (unsigned short*)p.out + base; q[0] = f32_to_bf16(o[0]); q[1] = f32_to_bf16(o[1]); q[2] = f32_to_bf16(o[2]); }
        else { float* q = (float*)p.out + base; q[0] = o[0]; q[1] = o[1]; q[2] = o[2]; }
    } else {
        const long long base = (long long)b * 3 * plane + pix;
        if (p.out_bf16) { unsigned short* q = (unsigned short*)p.out + base; q[0] = f32_to_bf16(o[0]); q[plane] = f32_to_bf16(o[1]); q[2 * plane] = f32_to_bf16(o[2]); }
        else { float* q = (float*)p.out + base; q[0] = o[0]; q[plane] = o[1]; q[2 * plane] = o[2]; }
    }
}

}  // namespace epi

extern "C" int epi_crop_patches(const void* frames, const long long* frame_offset, const int* frame_hw, const double* trans,
                                const int* do_flip, const float* color_scale, const float* mean_host, const float* std_host, int B,
                                int patch_h, int patch_w, void* out, int out_dtype, int out_layout, epi_stream_t stream) {
    return epi_crop_patches_occluded(frames, frame_offset, frame_hw, trans, do_flip, color_scale, mean_host, std_host, B, patch_h, patch_w, nullptr, nullptr,
                                     nullptr, nullptr, 0, out, out_dtype, out_layout, stream);
}

extern "C" int epi_crop_patches_occluded(const void* frames, const long long* frame_offset, const int* frame_hw, const double* trans,
                                         const int* do_flip, const float* color_scale, const float* mean_host, const float* std_host, int B,
                                         int patch_h, int patch_w, const void* occ_bank, const long long* occ_offset, const int* occ_hw,
                                         const int* occ_place, int max_occ, void* out, int out_dtype, int out_layout, epi_stream_t stream) {
    if (!frames || !frame_offset || !frame_hw || !trans || !out || B <= 0 || patch_h <= 0 || patch_w <= 0) return EPI_ERR_INVALID_ARGUMENT;
    if (occ_bank && (!occ_offset || !occ_hw || !occ_place || max_occ <= 0)) return EPI_ERR_INVALID_ARGUMENT;
    if ((out_dtype != EPI_F32 && out_dtype != EPI_BF16) || (out_layout != EPI_NCHW && out_layout != EPI_NHWC)) return EPI_ERR_UNSUPPORTED;
    if ((mean_host == nullptr) != (std_host == nullptr)) return EPI_ERR_INVALID_ARGUMENT;
    epi::PatchArgs a = {};
    a.frames = (const unsigned char*)frames; a.frame_offset = frame_offset; a.frame_hw = frame_hw; a.trans = trans; a.do_flip = do_flip;
    a.color_scale = color_scale; a.normalize = mean_host ? 1 : 0;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean_host ? mean_host[c] : 0.f; a.inv_std_is_std[c] = std_host ? std_host[c] : 1.f; }
    a.out = out; a.out_bf16 = out_dtype == EPI_BF16; a.out_nhwc = out_layout == EPI_NHWC; a.B = B; a.PH = patch_h; a.PW = patch_w;
    a.occ_bank = (const unsigned char*)occ_bank; a.occ_offset = occ_offset; a.occ_hw = occ_hw; a.occ_place = occ_place; a.max_occ = occ_bank ? max_occ : 0;
    const dim3 grid((unsigned)((patch_h * patch_w + 255) / 256), (unsigned)B);
    hipLaunchKernelGGL(epi::patch_crop_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
