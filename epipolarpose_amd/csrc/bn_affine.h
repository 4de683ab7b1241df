// BatchNorm affine (scale, shift) of one layer call, derived on the device from the batch sums -- shared by the BatchNorm kernels
// (csrc/bn_nhwc.hip) and the GEMM kernels that normalise their A operand on the fly (csrc/head_gemm.hip, GemmArgs::bn_in).
#pragma once
#include "common.h"

namespace epi {

// Per-channel affine of one BatchNorm call, derived inside the apply kernel.
// Training (sums != nullptr): mean / rstd from the batch sums (float64: E[x^2] - m^2 cancels), running statistics
// (momentum, unbiased variance) and num_batches_tracked updated by workgroup 0.  Inference: running statistics.
struct BnAffine {
    const float* sums;            // [ncopies][2C] partial batch sums (sum x | sum x^2), added up here, or nullptr
    int ncopies;
    long long R;
    double inv_r;                 // 1 / R (host-computed: no float64 divide on the device)
    const float *gamma, *beta;
    float eps, momentum;
    float *running_mean, *running_var;
    long long* num_batches;
    float *mean, *rstd, *scale, *shift;   // saved for the backward pass (mean/rstd may be nullptr in inference)
    float* bwd_sums;              // [2C] or nullptr: accumulator of this layer's NEXT backward pass, cleared here
};

// One channel's affine of one BatchNorm call (BnAffine) with the lead workgroup's side effects -- the derivation of bn_apply2d_kernel
__device__ __forceinline__ void bn_derive(const BnAffine& a, int C, int c, bool lead, float& sc, float& sh) {
    double m, var;
    if (a.sums) {
        float s1 = a.sums[c], s2 = a.sums[C + c];
        for (int k = 1; k < a.ncopies; ++k) { s1 += a.sums[2 * k * C + c]; s2 += a.sums[(2 * k + 1) * C + c]; }
        m = (double)s1 * a.inv_r;
        var = fma(-m, m, (double)s2 * a.inv_r);
        if (var < 0) var = 0;
    } else {
        m = a.running_mean[c];
        var = a.running_var[c];
    }
    const float rs = 1.0f / sqrtf((float)var + a.eps);
    sc = a.gamma[c] * rs;
    sh = a.beta[c] - (float)m * sc;
    if (lead) {
        if (a.mean) { a.mean[c] = (float)m; a.rstd[c] = rs; }
        a.scale[c] = sc;
        a.shift[c] = sh;
        if (a.sums && a.running_mean) {
            const double unbiased = (a.R > 1) ? var * (double)a.R / (double)(a.R - 1) : var;
            a.running_mean[c] = (1.f - a.momentum) * a.running_mean[c] + a.momentum * (float)m;
            a.running_var[c] = (1.f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
        }
        if (a.bwd_sums) { a.bwd_sums[c] = 0.f; a.bwd_sums[C + c] = 0.f; }
    }
}


}  // namespace epi
