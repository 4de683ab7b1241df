// H36M pose evaluation on the device: MPJPE, PA-MPJPE (Procrustes), N-MPJPE, 14-joint variants and per-axis errors.
//
// Replaces the per-sample / per-joint Python loop of H36M_Integral.evaluate (lib/dataset/h36m.py:168-378) and
// compute_similarity_transform (lib/utils/prep_h36m.py:108-168, np.linalg.svd of a 3x3) with one thread per sample,
// float64 throughout (the reference computes in float64).  Launch-latency bound (a few thousand samples).
#include "common.h"
#include "linalg3.h"

namespace epi {

constexpr int EV_MAXJ = 32;

__global__ void evaluate_poses_kernel(const double* __restrict__ pred, const double* __restrict__ gt, const double* __restrict__ pelvis_z,
                                      const double* __restrict__ fl, const double* __restrict__ cp, int N, int J, int root,
                                      const int* __restrict__ j14, int n14, double* __restrict__ metrics, double* __restrict__ per_joint) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double P[EV_MAXJ][3], G[EV_MAXJ][3];
    const double fx = fl[2 * n], fy = fl[2 * n + 1], cx = cp[2 * n], cy = cp[2 * n + 1], pz = pelvis_z[n];
    double muP[3] = {0, 0, 0}, muG[3] = {0, 0, 0};
    for (int j = 0; j < J; ++j) {                                       // h36m.py:222-237 (CamBackProj, prep_h36m.py:85-89)
        const double* p = pred + ((long long)n * J + j) * 3;
        const double* g = gt + ((long long)n * J + j) * 3;
        const double dp = p[2] + pz, dg = g[2] + pz;
        P[j][0] = (p[0] - cx) / fx * dp; P[j][1] = (p[1] - cy) / fy * dp; P[j][2] = dp;
        G[j][0] = (g[0] - cx) / fx * dg; G[j][1] = (g[1] - cy) / fy * dg; G[j][2] = dg;
        for (int k = 0; k < 3; ++k) { muP[k] += P[j][k]; muG[k] += G[j][k]; }
    }
    for (int k = 0; k < 3; ++k) { muP[k] /= J; muG[k] /= J; }
    // compute_similarity_transform(X = gt, Y = pred, compute_optimal_scale=True), prep_h36m.py:126-166
    double ssX = 0, ssY = 0, A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int j = 0; j < J; ++j)
        for (int k = 0; k < 3; ++k) {
            const double x0 = G[j][k] - muG[k], y0 = P[j][k] - muP[k];
            ssX += x0 * x0; ssY += y0 * y0;
        }
    const double normX = sqrt(ssX), normY = sqrt(ssY);
    for (int j = 0; j < J; ++j)
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) A[a][b] += (G[j][a] - muG[a]) / normX * ((P[j][b] - muP[b]) / normY);
    double U[3][3], s[3], V[3][3], T[3][3];
    svd3(A, U, s, V);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) T[a][b] = V[a][0] * U[b][0] + V[a][1] * U[b][1] + V[a][2] * U[b][2];       // T = V U^T
    const double d = det3(T);
    const double sg = (d > 0) ? 1.0 : ((d < 0) ? -1.0 : 0.0);                                                    // :150-153
    for (int a = 0; a < 3; ++a) V[a][2] *= sg;
    s[2] *= sg;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) T[a][b] = V[a][0] * U[b][0] + V[a][1] * U[b][1] + V[a][2] * U[b][2];
    const double bsc = (s[0] + s[1] + s[2]) * normX / normY;                                                     // :158
    double cvec[3];
    for (int b = 0; b < 3; ++b) cvec[b] = muG[b] - bsc * (muP[0] * T[0][b] + muP[1] * T[1][b] + muP[2] * T[2][b]);   // :166
    // aligned / normalised predictions, all root-centred (h36m.py:241-248)
    double pr[3], gr[3], ar[3], nr[3];
    for (int b = 0; b < 3; ++b) {
        pr[b] = P[root][b]; gr[b] = G[root][b];
        ar[b] = bsc * (P[root][0] * T[0][b] + P[root][1] * T[1][b] + P[root][2] * T[2][b]) + cvec[b];
        nr[b] = bsc * P[root][b];
    }
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < J; ++j) {
        double e = 0, ea = 0, en = 0, ax[3];
        for (int b = 0; b < 3; ++b) {
            const double gg = G[j][b] - gr[b];
            const double dpp = gg - (P[j][b] - pr[b]);
            const double al = bsc * (P[j][0] * T[0][b] + P[j][1] * T[1][b] + P[j][2] * T[2][b]) + cvec[b] - ar[b];
            const double nm = bsc * P[j][b] - nr[b];
            e += dpp * dpp; ea += (gg - al) * (gg - al); en += (gg - nm) * (gg - nm);
            ax[b] = fabs(dpp);
        }
        e = sqrt(e); ea = sqrt(ea); en = sqrt(en);
        per_joint[(long long)n * J + j] = e;
        acc[0] += e; acc[1] += ea; acc[2] += en; acc[6] += ax[0]; acc[7] += ax[1]; acc[8] += ax[2];
        bool in14 = false;
        for (int k = 0; k < n14; ++k) in14 = in14 || (j14[k] == j);
        if (in14) { acc[3] += e; acc[4] += ea; acc[5] += en; }
    }
    double* m = metrics + (long long)n * 9;
    m[0] = acc[0] / J; m[1] = acc[1] / J; m[2] = acc[2] / J;
    m[3] = acc[3] / n14; m[4] = acc[4] / n14; m[5] = acc[5] / n14;
    m[6] = acc[6] / J; m[7] = acc[7] / J; m[8] = acc[8] / J;
}

}  // namespace epi

extern "C" int epi_evaluate_poses(const double* pred_img, const double* gt_img, const double* pelvis_z, const double* fl, const double* c_p,
                                  int N, int J, int root, const int32_t* j14, int n14, double* metrics, double* per_joint,
                                  epi_stream_t stream) {
    if (!pred_img || !gt_img || !pelvis_z || !fl || !c_p || !j14 || !metrics || !per_joint || N <= 0 || J <= 0 || n14 <= 0)
        return EPI_ERR_INVALID_ARGUMENT;
    if (J > epi::EV_MAXJ || root < 0 || root >= J) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(epi::evaluate_poses_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, pred_img, gt_img, pelvis_z, fl, c_p,
                       N, J, root, j14, n14, metrics, per_joint);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}
