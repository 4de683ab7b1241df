// Small float64 linear algebra shared by the evaluation and triangulation kernels (one thread per problem).
#pragma once
#include "common.h"

namespace epi {

// One-sided Jacobi SVD of a 3x3:  A = U diag(s) V^T, singular values sorted descending (np.linalg.svd convention).
EPI_HD inline void svd3(const double (&A)[3][3], double (&U)[3][3], double (&s)[3], double (&V)[3][3]) {
    double M[3][3], W[3][3];                      // M[c] = column c of A*V ; W[c] = column c of V
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) { M[c][r] = A[r][c]; W[c][r] = (r == c) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += M[p][i] * M[p][i]; be += M[q][i] * M[q][i]; ga += M[p][i] * M[q][i]; }
                if (fabs(ga) > 1e-15 * sqrt(al * be)) {
                    rotated = true;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                    for (int i = 0; i < 3; ++i) {
                        const double mp = M[p][i], mq = M[q][i];
                        M[p][i] = c * mp - sn * mq; M[q][i] = sn * mp + c * mq;
                        const double wp = W[p][i], wq = W[q][i];
                        W[p][i] = c * wp - sn * wq; W[q][i] = sn * wp + c * wq;
                    }
                }
            }
        if (!rotated) break;
    }
    double n[3];
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; ++c) n[c] = sqrt(M[c][0] * M[c][0] + M[c][1] * M[c][1] + M[c][2] * M[c][2]);
    for (int a = 0; a < 2; ++a)
        for (int b = a + 1; b < 3; ++b)
            if (n[ord[b]] > n[ord[a]]) { const int t = ord[a]; ord[a] = ord[b]; ord[b] = t; }
    for (int k = 0; k < 3; ++k) {
        const int c = ord[k];
        s[k] = n[c];
        for (int r = 0; r < 3; ++r) { V[r][k] = W[c][r]; U[r][k] = (n[c] > 0) ? M[c][r] / n[c] : 0.0; }
    }
    if (!(s[2] > 1e-300)) {        // rank-deficient: complete U's last column so that U stays orthonormal
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
}

EPI_HD __forceinline__ double det3(const double (&T)[3][3]) {
    return T[0][0] * (T[1][1] * T[2][2] - T[1][2] * T[2][1]) - T[0][1] * (T[1][0] * T[2][2] - T[1][2] * T[2][0]) +
           T[0][2] * (T[1][0] * T[2][1] - T[1][1] * T[2][0]);
}

}  // namespace epi
