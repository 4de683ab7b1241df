// Scoring kernels of the least-median-of-squares fundamental-matrix estimate behind cameras.py:136-143
// (`cv2.findFundamentalMat(u1, u2, cv2.FM_LMEDS)`; OpenCV 4.1.0, modules/calib3d/src/ptsetreg.cpp LMeDSPointSetRegistrator::run and
// fundam.cpp FMEstimatorCallback::computeError -- the dependency is absent from /root/reference, its published algorithm is restated).
// The host draws the 7-point samples with OpenCV's generator and solves them (utils/triangulation.py find_fundamental_mat_lmeds); what
// scales with the number of correspondences runs here: for every candidate matrix the symmetric epipolar error of EVERY point pair and
// the MEDIAN of those errors.
//
// epi_fundamental_lmeds_medians: one workgroup per candidate.  err_i = max(d1^2 / |l1|^2, d2^2 / |l2|^2) as float32 (the reference
// implementation stores float), median = the middle element of the sorted errors (odd N) or the mean of the two middle ones (even N).
// No sort: errors are non-negative floats, so their bit patterns order like unsigned integers, and the k-th smallest is found by a 4-pass
// radix SELECT (8 bits per pass: histogram of the next byte among the values that match the prefix found so far, in LDS); the errors are
// recomputed in every pass (≈40 flops per point) instead of being stored, so N is unbounded.
#include "common.h"

namespace epi {

__host__ __device__ __forceinline__ float fm_error(const double* __restrict__ F, double x1, double y1, double x2, double y2) {      // (host: tests/hostcheck)
    double a = F[0] * x1 + F[1] * y1 + F[2];
    double b = F[3] * x1 + F[4] * y1 + F[5];
    double c = F[6] * x1 + F[7] * y1 + F[8];
    const double s2 = 1. / (a * a + b * b);
    const double d2 = x2 * a + y2 * b + c;
    a = F[0] * x2 + F[3] * y2 + F[6];
    b = F[1] * x2 + F[4] * y2 + F[7];
    c = F[2] * x2 + F[5] * y2 + F[8];
    const double s1 = 1. / (a * a + b * b);
    const double d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);         // (std::max(a, b): b only when a < b -- NaN handling as the reference's)
}

constexpr int FM_THREADS = 256;

// k-th smallest (0-based) of the N error bit patterns of candidate F: 4 passes over the points
__device__ unsigned int fm_select(const double* __restrict__ F, const double* __restrict__ u1, const double* __restrict__ u2, int N, int k,
                                  unsigned int* hist /* [256] LDS */, unsigned int* shared /* [2] LDS */) {
    unsigned int prefix = 0u, mask = 0u;
    int rank = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int t = threadIdx.x; t < 256; t += FM_THREADS) hist[t] = 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < N; i += FM_THREADS) {
            const unsigned int bits = __float_as_uint(fm_error(F, u1[2 * i], u1[2 * i + 1], u2[2 * i], u2[2 * i + 1]));
            if ((bits & mask) == prefix) atomicAdd(&hist[(bits >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int r = rank, b = 0;
            for (; b < 255; ++b) {
                if (r < (int)hist[b]) break;
                r -= (int)hist[b];
            }
            shared[0] = (unsigned int)b;
            shared[1] = (unsigned int)r;
        }
        __syncthreads();
        prefix |= shared[0] << shift;
        mask |= 255u << shift;
        rank = (int)shared[1];
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(FM_THREADS) void fm_lmeds_medians_kernel(const double* __restrict__ Fs, const double* __restrict__ u1,
                                                                      const double* __restrict__ u2, int N, double* __restrict__ medians) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned int shared[2];
    __shared__ double Fl[9];
    if (threadIdx.x < 9) Fl[threadIdx.x] = Fs[(long long)blockIdx.x * 9 + threadIdx.x];
    __syncthreads();
    const unsigned int hi = fm_select(Fl, u1, u2, N, N / 2, hist, shared);
    double med = (double)__uint_as_float(hi);
    if ((N & 1) == 0) {
        const unsigned int lo = fm_select(Fl, u1, u2, N, N / 2 - 1, hist, shared);
        med = (double)(__uint_as_float(lo) + __uint_as_float(hi)) * 0.5;          // float sum, then * 0.5 in double (as the reference)
    }
    if (threadIdx.x == 0) medians[blockIdx.x] = med;
}

__global__ void fm_errors_kernel(const double* __restrict__ F, const double* __restrict__ u1, const double* __restrict__ u2, int N,
                                 float* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) err[i] = fm_error(F, u1[2 * i], u1[2 * i + 1], u2[2 * i], u2[2 * i + 1]);
}

}  // namespace epi

using namespace epi;

extern "C" int epi_fundamental_lmeds_medians(const double* F, int H, const double* u1, const double* u2, int N, double* medians, epi_stream_t stream) {
    if (!F || !u1 || !u2 || !medians || H <= 0 || N <= 0) return EPI_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(fm_lmeds_medians_kernel, dim3((unsigned)H), dim3(FM_THREADS), 0, (hipStream_t)stream, F, u1, u2, N, medians);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_fundamental_errors(const double* F, const double* u1, const double* u2, int N, float* err, epi_stream_t stream) {
    if (!F || !u1 || !u2 || !err || N <= 0) return EPI_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(fm_errors_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, F, u1, u2, N, err);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
