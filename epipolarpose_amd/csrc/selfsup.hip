// Self-supervision geometry on the device: crop affine, patch->image decode, multi-view triangulation
// (iterative LS / linear LS / homogeneous DLT) and re-projection into per-view pseudo labels.
//
// Replaces the host-side Python loops of the reference (four nested loops: pair x joint x <=10
// iterations x cv2.solve, after a blocking D2H copy):
//   lib/utils/img_utils.py:63-111,141-243, lib/utils/triangulation.py:8-181, lib/utils/prep_h36m.py:170-204.
// One thread owns one (group, joint); the fused kernel keeps the whole SS step in ONE launch.
#include "common.h"
#include <type_traits>
#include "linalg3.h"

namespace epi {

struct Aff { double a00, a01, a02, a10, a11, a12; };

// img_utils.py:72-105 (gen_trans_from_patch_cv) with its float32 roundings (:82-83,87-99), then the exact
// affine through the three point pairs (cv2.getAffineTransform) in closed form: the patch-side triple is
// axis aligned, so  org = s0 + (px-dcx)/dhw * (s2-s0) + (py-dcy)/dhh * (s1-s0).
EPI_HD __forceinline__ void patch_affines(double cx, double cy, double bw, double bh, double scale, double rot,
                                              double pw, double ph, Aff* inv, Aff* fwd) {
    const double rot_rad = 3.141592653589793 * rot / 180.0;
    const double sn = sin(rot_rad), cs = cos(rot_rad);
    const double hh = (double)(float)(bh * scale * 0.5);
    const double hw = (double)(float)(bw * scale * 0.5);
    const double dnx = (double)(float)(0.0 * cs - hh * sn), dny = (double)(float)(0.0 * sn + hh * cs);
    const double rtx = (double)(float)(hw * cs - 0.0 * sn), rty = (double)(float)(hw * sn + 0.0 * cs);
    const double s0x = (double)(float)cx, s0y = (double)(float)cy;
    const double s1x = (double)(float)(cx + dnx), s1y = (double)(float)(cy + dny);
    const double s2x = (double)(float)(cx + rtx), s2y = (double)(float)(cy + rty);
    const float dcx = (float)(pw * 0.5), dcy = (float)(ph * 0.5);
    const double dhw = (double)((dcx + (float)(pw * 0.5)) - dcx);
    const double dhh = (double)((dcy + (float)(ph * 0.5)) - dcy);
    const double exx = (s2x - s0x) / dhw, exy = (s2y - s0y) / dhw;     // image of the patch x axis
    const double eyx = (s1x - s0x) / dhh, eyy = (s1y - s0y) / dhh;     // image of the patch y axis
    Aff a;
    a.a00 = exx; a.a01 = eyx; a.a02 = s0x - exx * (double)dcx - eyx * (double)dcy;
    a.a10 = exy; a.a11 = eyy; a.a12 = s0y - exy * (double)dcx - eyy * (double)dcy;
    if (inv) *inv = a;
    if (fwd) {
        const double det = a.a00 * a.a11 - a.a01 * a.a10;
        const double i00 = a.a11 / det, i01 = -a.a01 / det, i10 = -a.a10 / det, i11 = a.a00 / det;
        fwd->a00 = i00; fwd->a01 = i01; fwd->a02 = -(i00 * a.a02 + i01 * a.a12);
        fwd->a10 = i10; fwd->a11 = i11; fwd->a12 = -(i10 * a.a02 + i11 * a.a12);
    }
}

// Least squares of an (R x 3) system by Householder QR, in place (stand-in for cv2.solve(DECOMP_SVD),
// triangulation.py:95,155 -- identical for full column rank).  Rows beyond the used views are zero.
template <typename T, int R>
EPI_HD __forceinline__ void qr_solve3(T (&A)[R][3], T (&b)[R], T (&x)[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        T tail2 = 0;                                     // sub-diagonal part kept separate: no cancellation
#pragma unroll
        for (int i = k + 1; i < R; ++i) tail2 += A[i][k] * A[i][k];
        const T akk = A[k][k];
        const T nrm = sqrt(akk * akk + tail2);
        const T alpha = (akk > 0) ? -nrm : nrm;
        const T vk = akk - alpha;                       // same sign as akk: no cancellation
        const T vn2 = vk * vk + tail2;                  // |v|^2
        const T inv = (vn2 > 0) ? (T)2 / vn2 : (T)0;
#pragma unroll
        for (int j = k + 1; j < 3; ++j) {
            T dot = vk * A[k][j];
#pragma unroll
            for (int i = k + 1; i < R; ++i) dot += A[i][k] * A[i][j];
            const T f = dot * inv;
            A[k][j] -= f * vk;
#pragma unroll
            for (int i = k + 1; i < R; ++i) A[i][j] -= f * A[i][k];
        }
        {
            T dot = vk * b[k];
#pragma unroll
            for (int i = k + 1; i < R; ++i) dot += A[i][k] * b[i];
            const T f = dot * inv;
            b[k] -= f * vk;
#pragma unroll
            for (int i = k + 1; i < R; ++i) b[i] -= f * A[i][k];
        }
        A[k][k] = alpha;
    }
    x[2] = b[2] / A[2][2];
    x[1] = (b[1] - A[1][2] * x[2]) / A[1][1];
    x[0] = (b[0] - A[0][1] * x[1] - A[0][2] * x[2]) / A[0][0];
}

// rows of the inhomogeneous system (triangulation.py:138-148): A = C P[:, :3], b = -(C P[:, 3])
template <typename T, int NV>
EPI_HD __forceinline__ void build_ls(const T (&u)[NV][2], const T (&P)[NV][12], int nv, T (&A)[2 * NV][3], T (&b)[2 * NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool on = v < nv;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) A[2 * v + r][c] = on ? (u[v][r] * P[v][8 + c] - P[v][4 * r + c]) : (T)0;
            b[2 * v + r] = on ? (P[v][4 * r + 3] - u[v][r] * P[v][11]) : (T)0;
        }
    }
}

template <typename T, int NV>
EPI_HD __forceinline__ int tri_linear_ls(const T (&u)[NV][2], const T (&P)[NV][12], int nv, T (&x)[3]) {
    T A[2 * NV][3], b[2 * NV];
    build_ls<T, NV>(u, P, nv, A, b);
    qr_solve3<T, 2 * NV>(A, b, x);
    return 1;                                            // triangulation.py:97: status all True
}

// triangulation.py:104-181
template <typename T, int NV>
EPI_HD __forceinline__ int tri_iterative_ls(const T (&u)[NV][2], const T (&P)[NV][12], int nv, T tol, int max_iter, T (&x)[3]) {
    T A[2 * NV][3], b[2 * NV], d[NV], dn[NV];
    build_ls<T, NV>(u, P, nv, A, b);
#pragma unroll
    for (int v = 0; v < NV; ++v) { d[v] = 1; dn[v] = 1; }                         // :151
    x[0] = x[1] = x[2] = 0;
    for (int it = 0; it < max_iter; ++it) {                                     // :153
        T Aw[2 * NV][3], bw[2 * NV];
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) { Aw[i][0] = A[i][0]; Aw[i][1] = A[i][1]; Aw[i][2] = A[i][2]; bw[i] = b[i]; }
        qr_solve3<T, 2 * NV>(Aw, bw, x);                                         // :155
        bool conv = true;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv) {
                dn[v] = P[v][8] * x[0] + P[v][9] * x[1] + P[v][10] * x[2] + P[v][11];   // :158-159
                if (!(fabs(dn[v] - d[v]) <= tol)) conv = false;                 // :161-162
            }
        }
        if (conv) break;                                                         // :163
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv) {
                const T w = (T)1 / dn[v];                                        // :166-169 (cumulative)
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    A[2 * v + r][0] *= w; A[2 * v + r][1] *= w; A[2 * v + r][2] *= w; b[2 * v + r] *= w;
                }
                d[v] = dn[v];                                                    // :172-173
            }
        }
    }
    bool all_front = true;
    int code = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < nv) {
            if (!(dn[v] > 0)) all_front = false;
            if (dn[v] <= 0) code -= (1 << v);                                    // :178-179
        }
    }
    return all_front ? 1 : code;                                                 // :176-177
}

// triangulation.py:8-27 == cv2.triangulatePoints: homogeneous 2Vx4 system, right-singular vector of the
// smallest singular value.  One-sided (Hestenes) Jacobi SVD in float64 on the columns of M.
template <int NV>
EPI_HD __forceinline__ int tri_dlt(const double (&u)[NV][2], const double (&P)[NV][12], int nv, double (&x)[3]) {
    double M[4][2 * NV];      // column-major: M[c][row]
    double Vm[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bool on = v < nv;
            M[c][2 * v] = on ? (u[v][0] * P[v][8 + c] - P[v][c]) : 0.0;
            M[c][2 * v + 1] = on ? (u[v][1] * P[v][8 + c] - P[v][4 + c]) : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) Vm[c][r] = (r == c) ? 1.0 : 0.0;     // Vm[c] = c-th column of V
    }
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double al = 0, be = 0, ga = 0;
#pragma unroll
                for (int i = 0; i < 2 * NV; ++i) { al += M[p][i] * M[p][i]; be += M[q][i] * M[q][i]; ga += M[p][i] * M[q][i]; }
                if (fabs(ga) > 4e-16 * sqrt(al * be)) {
                    rotated = true;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double t = ((zeta >= 0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                    for (int i = 0; i < 2 * NV; ++i) {
                        const double mp = M[p][i], mq = M[q][i];
                        M[p][i] = c * mp - s * mq;
                        M[q][i] = s * mp + c * mq;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const double vp = Vm[p][i], vq = Vm[q][i];
                        Vm[p][i] = c * vp - s * vq;
                        Vm[q][i] = s * vp + c * vq;
                    }
                }
            }
        }
        if (!rotated) break;
    }
    int best = 0;
    double bn = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double n2 = 0;
#pragma unroll
        for (int i = 0; i < 2 * NV; ++i) n2 += M[c][i] * M[c][i];
        if (c == 0 || n2 < bn) { bn = n2; best = c; }
    }
    double h[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c == best) { h[0] = Vm[c][0]; h[1] = Vm[c][1]; h[2] = Vm[c][2]; h[3] = Vm[c][3]; }
    x[0] = h[0] / h[3]; x[1] = h[1] / h[3]; x[2] = h[2] / h[3];                 // :24
    const double mx = fmax(fabs(x[0]), fmax(fabs(x[1]), fabs(x[2])));
    return (mx <= 1.0e16) ? 1 : 0;                                               // :25 (NaN -> 0)
}

// ---------------------------------------------------------------------------------------------------------------------
// Polynomial ("optimal") two-view triangulation, triangulation.py:184-220: F from the two projection matrices,
// cv2.correctMatches (Hartley & Sturm; HZ Algorithm 12.1), then the linear-eigen solve on the corrected matches.
// ---------------------------------------------------------------------------------------------------------------------

// triangulation.py:196-204.  P_full = [P; 0 0 0 1];  P_canon = P2_full inv(P1_full) = [M2 M1^-1 | p2 - M2 M1^-1 p1];
// F = [t]_x R with R = P_canon[:3,:3], t = P_canon[:3,3].
EPI_HD __forceinline__ void fundamental_from_P(const double (&P1)[12], const double (&P2)[12], double (&F)[3][3]) {
    const double m00 = P1[0], m01 = P1[1], m02 = P1[2], m10 = P1[4], m11 = P1[5], m12 = P1[6], m20 = P1[8], m21 = P1[9], m22 = P1[10];
    double inv[3][3];
    inv[0][0] = m11 * m22 - m12 * m21; inv[0][1] = m02 * m21 - m01 * m22; inv[0][2] = m01 * m12 - m02 * m11;
    inv[1][0] = m12 * m20 - m10 * m22; inv[1][1] = m00 * m22 - m02 * m20; inv[1][2] = m02 * m10 - m00 * m12;
    inv[2][0] = m10 * m21 - m11 * m20; inv[2][1] = m01 * m20 - m00 * m21; inv[2][2] = m00 * m11 - m01 * m10;
    const double det = m00 * inv[0][0] + m01 * inv[1][0] + m02 * inv[2][0];
    const double rd = 1.0 / det;
    double R[3][3], t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[r][c] = (P2[4 * r] * inv[0][c] + P2[4 * r + 1] * inv[1][c] + P2[4 * r + 2] * inv[2][c]) * rd;
        t[r] = P2[4 * r + 3] - (R[r][0] * P1[3] + R[r][1] * P1[7] + R[r][2] * P1[11]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                      // F[:, c] = t x R[:, c]
        F[0][c] = t[1] * R[2][c] - t[2] * R[1][c];
        F[1][c] = t[2] * R[0][c] - t[0] * R[2][c];
        F[2][c] = t[0] * R[1][c] - t[1] * R[0][c];
    }
}

EPI_HD __forceinline__ double horner(const double* c, int n, double x) {
    double v = c[n];
    for (int k = n - 1; k >= 0; --k) v = fma(v, x, c[k]);
    return v;
}

// Real roots in [-1, 1] of the polynomial c[0..N] (ascending powers), by the derivative chain: the roots of p^(k+1) cut
// [-1, 1] into pieces on which p^(k) is monotonic, each piece with a sign change is bisected (then Newton-polished).
// Needs no bound on the roots and tolerates vanishing leading coefficients.  `out` also receives the break points that
// were examined on the way (roots of p'), which callers may use as extra candidates.
template <int N>
EPI_HD int real_roots_unit(const double (&c)[N + 1], double (&out)[N]) {
    double D[N][N + 1];                          // D[k] = coefficients of the k-th derivative (degree N-k)
    for (int i = 0; i <= N; ++i) D[0][i] = c[i];
    for (int k = 1; k < N; ++k)
        for (int i = 0; i <= N - k; ++i) D[k][i] = D[k - 1][i + 1] * (double)(i + 1);
    double prev[N + 2], cur[N + 2];
    int np_ = 0;
    for (int k = N - 1; k >= 0; --k) {           // roots of D[k] (degree N-k) from the roots of D[k+1] in `prev`
        const int deg = N - k;
        int nc = 0;
        double lo = -1.0, flo = horner(D[k], deg, lo);
        for (int s = 0; s <= np_; ++s) {
            const double hi = (s < np_) ? prev[s] : 1.0;
            const double fhi = horner(D[k], deg, hi);
            if (hi > lo) {
                if (flo == 0.0) { if (s == 0) cur[nc++] = lo; }
                else if ((flo < 0) != (fhi < 0) && fhi != 0.0) {
                    double a = lo, b = hi;
                    const bool up = flo < 0;
                    for (int it = 0; it < 56; ++it) {
                        const double m = 0.5 * (a + b);
                        const double fm = horner(D[k], deg, m);
                        if ((fm < 0) == up) a = m; else b = m;
                    }
                    double r = 0.5 * (a + b);
                    if (k + 1 < N) {                                   // Newton polish for relative accuracy near 0
                        for (int it = 0; it < 2; ++it) {
                            const double d = horner(D[k + 1], deg - 1, r);
                            if (d != 0.0) {
                                const double r2 = r - horner(D[k], deg, r) / d;
                                if (r2 > lo && r2 < hi) r = r2;
                            }
                        }
                    }
                    cur[nc++] = r;
                }
                if (fhi == 0.0) cur[nc++] = hi;
            }
            lo = hi; flo = fhi;
        }
        for (int i = 0; i < nc && i < N; ++i) prev[i] = cur[i];
        np_ = nc < N ? nc : N;
        if (k == 0) for (int i = 0; i < np_; ++i) out[i] = prev[i];
    }
    return np_;
}

// cv2.correctMatches for one pair: (x1, y1) <-> (x2, y2) moved onto the closest pair with x2^T F x1 = 0.
EPI_HD void correct_match(const double (&F)[3][3], double (&p1)[2], double (&p2)[2]) {
    double Ft[3][3];                                                    // (ii) T2^T F T1,  T = [[1,0,x],[0,1,y],[0,0,1]]
    {
        double FT[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { FT[r][0] = F[r][0]; FT[r][1] = F[r][1]; FT[r][2] = F[r][0] * p1[0] + F[r][1] * p1[1] + F[r][2]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) { Ft[0][c] = FT[0][c]; Ft[1][c] = FT[1][c]; Ft[2][c] = p2[0] * FT[0][c] + p2[1] * FT[1][c] + FT[2][c]; }
    }
    double U[3][3], sv[3], V[3][3];
    svd3(Ft, U, sv, V);
    double e1[3] = {V[0][2], V[1][2], V[2][2]};                         // (iii) F e1 = 0
    double e2[3] = {U[1][0] * U[2][1] - U[2][0] * U[1][1],              //       e2^T F = 0: e2 = u0 x u1
                    U[2][0] * U[0][1] - U[0][0] * U[2][1],
                    U[0][0] * U[1][1] - U[1][0] * U[0][1]};
    const double n1 = 1.0 / sqrt(e1[0] * e1[0] + e1[1] * e1[1]), n2 = 1.0 / sqrt(e2[0] * e2[0] + e2[1] * e2[1]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { e1[i] *= n1; e2[i] *= n2; }
    // (iv)-(v)  R = [[ex, ey, 0], [-ey, ex, 0], [0, 0, 1]];  Fr = R2 Ft R1^T -- only the lower-right 2x2 block is needed
    double G1[3][3];                                                    // Ft R1^T
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        G1[r][0] = Ft[r][0] * e1[0] + Ft[r][1] * e1[1];
        G1[r][1] = -Ft[r][0] * e1[1] + Ft[r][1] * e1[0];
        G1[r][2] = Ft[r][2];
    }
    const double a = -e2[1] * G1[0][1] + e2[0] * G1[1][1], b = -e2[1] * G1[0][2] + e2[0] * G1[1][2];
    const double c = G1[2][1], d = G1[2][2], f1 = e1[2], f2 = e2[2];   // (vi)
    // (vii) g(t) = t((at+b)^2 + f2^2 (ct+d)^2)^2 - (ad-bc)(1+f1^2 t^2)^2 (at+b)(ct+d), ascending coefficients
    const double q0 = b * b + f2 * f2 * d * d, q1 = 2.0 * (a * b + f2 * f2 * c * d), q2 = a * a + f2 * f2 * c * c;
    const double k = a * d - b * c, w = f1 * f1;
    const double r0 = b * d, r1 = a * d + b * c, r2 = a * c;           // (at+b)(ct+d)
    double g[7];
    g[0] = -k * r0;
    g[1] = q0 * q0 - k * r1;
    g[2] = 2.0 * q0 * q1 - k * (r2 + 2.0 * w * r0);
    g[3] = q1 * q1 + 2.0 * q0 * q2 - k * 2.0 * w * r1;
    g[4] = 2.0 * q1 * q2 - k * (2.0 * w * r2 + w * w * r0);
    g[5] = q2 * q2 - k * w * w * r1;
    g[6] = -k * w * w * r2;
    double gmax = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) gmax = fmax(gmax, fabs(g[i]));
    // (viii) candidates: real roots with |t| <= 1, and with |t| >= 1 through the reversed polynomial in 1/t; t = infinity.
    double best_t = 0.0, best_cost = 1.0e300;
    bool at_inf = false;
    auto cost = [&](double t) {
        const double n = c * t + d, m = a * t + b;
        return t * t / (1.0 + w * t * t) + n * n / (m * m + f2 * f2 * n * n);
    };
    if (gmax > 0) {
        double roots[6];
        int nr = real_roots_unit<6>(g, roots);
        for (int i = 0; i < nr; ++i) {
            const double cs = cost(roots[i]);
            if (cs < best_cost) { best_cost = cs; best_t = roots[i]; }
        }
        double gr[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) gr[i] = g[6 - i];
        nr = real_roots_unit<6>(gr, roots);
        for (int i = 0; i < nr; ++i) {
            if (roots[i] == 0.0) continue;
            const double t = 1.0 / roots[i];
            const double cs = cost(t);
            if (cs < best_cost) { best_cost = cs; best_t = t; }
        }
    } else {
        best_cost = cost(0.0);                                          // g == 0: every t is stationary
    }
    if (w > 0) {
        const double c_inf = 1.0 / w + c * c / (a * a + f2 * f2 * c * c);
        if (c_inf < best_cost) at_inf = true;
    }
    double l1[3], l2[3];                                                // (ix)
    if (at_inf) { l1[0] = f1; l1[1] = 0.0; l1[2] = -1.0; l2[0] = -f2 * c; l2[1] = a; l2[2] = c; }
    else {
        const double t = best_t;
        l1[0] = t * f1; l1[1] = 1.0; l1[2] = -t;
        l2[0] = -f2 * (c * t + d); l2[1] = a * t + b; l2[2] = c * t + d;
    }
    // closest point of each line to the origin, then (x) back through R^T and T
    const double x1[3] = {-l1[0] * l1[2], -l1[1] * l1[2], l1[0] * l1[0] + l1[1] * l1[1]};
    const double x2[3] = {-l2[0] * l2[2], -l2[1] * l2[2], l2[0] * l2[0] + l2[1] * l2[1]};
    const double y1[3] = {e1[0] * x1[0] - e1[1] * x1[1], e1[1] * x1[0] + e1[0] * x1[1], x1[2]};
    const double y2[3] = {e2[0] * x2[0] - e2[1] * x2[1], e2[1] * x2[0] + e2[0] * x2[1], x2[2]};
    p1[0] = (y1[0] + p1[0] * y1[2]) / y1[2]; p1[1] = (y1[1] + p1[1] * y1[2]) / y1[2];
    p2[0] = (y2[0] + p2[0] * y2[2]) / y2[2]; p2[1] = (y2[1] + p2[1] * y2[2]) / y2[2];
}

// ---------------------------------------------------------------------------------------------------------------------
// Fundamental matrix from 2-D matches -- cv2.findFundamentalMat(u1, u2, FM_8POINT) (triangulation.py:216; cameras.py:136-143
// asks for FM_LMEDS, whose random sampling is not reproducible: on outlier-free matches it converges to the same F).
// Normalised 8-point algorithm as OpenCV's run8Point runs it: isotropic normalisation (centroid, mean distance sqrt 2),
// eigenvector of the 9x9 A^T A with the smallest eigenvalue (cyclic Jacobi, float64), rank 2 through the 3x3 SVD,
// de-normalisation, F[2][2] = 1.  x2^T F x1 = 0.  Returns 0 for degenerate input (fewer than 8 matches, coincident points).
// ---------------------------------------------------------------------------------------------------------------------
EPI_HD inline int fundamental_8point_one(const double* u1, const double* u2, int n, double (&F)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) F[i][j] = 0.0;
    if (n < 8) return 0;
    double c1[2] = {0, 0}, c2[2] = {0, 0};
    for (int i = 0; i < n; ++i) { c1[0] += u1[2 * i]; c1[1] += u1[2 * i + 1]; c2[0] += u2[2 * i]; c2[1] += u2[2 * i + 1]; }
    const double rn = 1.0 / (double)n;
    c1[0] *= rn; c1[1] *= rn; c2[0] *= rn; c2[1] *= rn;
    double d1 = 0, d2 = 0;
    for (int i = 0; i < n; ++i) {
        const double ax = u1[2 * i] - c1[0], ay = u1[2 * i + 1] - c1[1], bx = u2[2 * i] - c2[0], by = u2[2 * i + 1] - c2[1];
        d1 += sqrt(ax * ax + ay * ay);
        d2 += sqrt(bx * bx + by * by);
    }
    d1 *= rn; d2 *= rn;
    if (!(d1 >= 1.1920928955078125e-07) || !(d2 >= 1.1920928955078125e-07)) return 0;      // FLT_EPSILON on the mean distance
    const double s1 = 1.4142135623730951 / d1, s2 = 1.4142135623730951 / d2;
    double A[9][9], V[9][9];
    for (int i = 0; i < 9; ++i)
        for (int j = 0; j < 9; ++j) { A[i][j] = 0.0; V[i][j] = (i == j) ? 1.0 : 0.0; }
    for (int i = 0; i < n; ++i) {
        const double x1 = (u1[2 * i] - c1[0]) * s1, y1 = (u1[2 * i + 1] - c1[1]) * s1;
        const double x2 = (u2[2 * i] - c2[0]) * s2, y2 = (u2[2 * i + 1] - c2[1]) * s2;
        const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int a = 0; a < 9; ++a)
            for (int b = a; b < 9; ++b) A[a][b] += r[a] * r[b];
    }
    for (int a = 0; a < 9; ++a)
        for (int b = 0; b < a; ++b) A[a][b] = A[b][a];
    // cyclic Jacobi on the symmetric 9x9: A <- J^T A J, V <- V J
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0, diag = 0;
        for (int a = 0; a < 9; ++a) {
            diag += A[a][a] * A[a][a];
            for (int b = a + 1; b < 9; ++b) off += A[a][b] * A[a][b];
        }
        if (off <= 1e-30 * diag) break;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = A[p][q];
                if (fabs(apq) <= 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = ((theta >= 0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 9; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - sn * akq;
                    A[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 9; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - sn * aqk;
                    A[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 9; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq;
                    V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    int best = 0;
    for (int a = 1; a < 9; ++a)
        if (A[a][a] < A[best][best]) best = a;
    double F0[3][3], U[3][3], sv[3], W[3][3];
    for (int i = 0; i < 9; ++i) F0[i / 3][i % 3] = V[i][best];
    svd3(F0, U, sv, W);
    // rank 2: drop the smallest singular value.  F0 = U diag(s0, s1, 0) W^T
    double G[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[i][j] = U[i][0] * sv[0] * W[j][0] + U[i][1] * sv[1] * W[j][1];
    // F = T2^T G T1,  T = [[s, 0, -s cx], [0, s, -s cy], [0, 0, 1]]
    double GT[3][3];
    for (int i = 0; i < 3; ++i) {
        GT[i][0] = G[i][0] * s1;
        GT[i][1] = G[i][1] * s1;
        GT[i][2] = -G[i][0] * s1 * c1[0] - G[i][1] * s1 * c1[1] + G[i][2];
    }
    for (int j = 0; j < 3; ++j) {
        F[0][j] = s2 * GT[0][j];
        F[1][j] = s2 * GT[1][j];
        F[2][j] = -s2 * c2[0] * GT[0][j] - s2 * c2[1] * GT[1][j] + GT[2][j];
    }
    if (fabs(F[2][2]) > 1.1920928955078125e-07) {
        const double inv = 1.0 / F[2][2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) F[i][j] *= inv;
    }
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Bulk (fp32-storage) variants.  The drop-in float64 path above follows the reference operation by operation (Householder QR of
// the 2V x 3 system per re-weighting round, one-sided Jacobi SVD of the 2V x 4 system) and is fp64-VALU-bound: ~3000 float64
// operations and ~200 live registers per joint.  When the inputs are float32 anyway (tolerance 1e-2 mm, include/epipolar_hip.h)
// the same iteration runs on 3 x 3 normal equations:  every re-weighting round scales BOTH rows of view v by 1/d_v, so
// A^T W^2 A = sum_v s_v G_v with G_v = a_v0 a_v0^T + a_v1 a_v1^T (6 numbers per view) fixed and only s_v = prod 1/d_v^2 changing:
// ~25 fused multiply-adds to rebuild, a closed-form symmetric 3 x 3 solve, V depths -- an order of magnitude fewer float64
// operations and a quarter of the registers.  cond(A)^2 ~ 1e4 costs ~1e-12 relative in float64: far inside the float32 envelope.
// ---------------------------------------------------------------------------------------------------------------------
// 1 / x to float64 round-off without the IEEE division sequence (~30 instructions on gfx950): hardware estimate + two Newton steps
EPI_HD __forceinline__ double fast_rcp(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
#else
    return 1.0 / v;
#endif
}

// 1 / sqrt(x): hardware estimate + two Newton steps
EPI_HD __forceinline__ double fast_rsq(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(v);
    r = r * fma(-0.5 * v * r, r, 1.5);
    r = r * fma(-0.5 * v * r, r, 1.5);
    return r;
#else
    return 1.0 / sqrt(v);
#endif
}

EPI_HD __forceinline__ void solve_sym3(const double (&n)[6], const double (&r)[3], double (&x)[3]) {
    // n = (n00, n01, n02, n11, n12, n22); adjugate / determinant
    const double c00 = n[3] * n[5] - n[4] * n[4], c01 = n[2] * n[4] - n[1] * n[5], c02 = n[1] * n[4] - n[2] * n[3];
    const double c11 = n[0] * n[5] - n[2] * n[2], c12 = n[1] * n[2] - n[0] * n[4], c22 = n[0] * n[3] - n[1] * n[1];
    const double inv = fast_rcp(n[0] * c00 + n[1] * c01 + n[2] * c02);
    x[0] = (c00 * r[0] + c01 * r[1] + c02 * r[2]) * inv;
    x[1] = (c01 * r[0] + c11 * r[1] + c12 * r[2]) * inv;
    x[2] = (c02 * r[0] + c12 * r[1] + c22 * r[2]) * inv;
}

template <int NV>
EPI_HD __forceinline__ int tri_iterative_ne(const double (&u)[NV][2], const double (&P)[NV][12], int nv, double tol, int max_iter,
                                            double (&x)[3]) {
    double G[NV][6], h[NV][3], s[NV], d[NV], dn[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool on = v < nv;
        double a[2][3], b[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) a[r][c] = on ? (u[v][r] * P[v][8 + c] - P[v][4 * r + c]) : 0.0;     // triangulation.py:138-148
            b[r] = on ? (P[v][4 * r + 3] - u[v][r] * P[v][11]) : 0.0;
        }
        G[v][0] = a[0][0] * a[0][0] + a[1][0] * a[1][0]; G[v][1] = a[0][0] * a[0][1] + a[1][0] * a[1][1];
        G[v][2] = a[0][0] * a[0][2] + a[1][0] * a[1][2]; G[v][3] = a[0][1] * a[0][1] + a[1][1] * a[1][1];
        G[v][4] = a[0][1] * a[0][2] + a[1][1] * a[1][2]; G[v][5] = a[0][2] * a[0][2] + a[1][2] * a[1][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) h[v][c] = a[0][c] * b[0] + a[1][c] * b[1];
        s[v] = 1.0; d[v] = 1.0; dn[v] = 1.0;                                                               // :151
    }
    x[0] = x[1] = x[2] = 0;
    for (int it = 0; it < max_iter; ++it) {                                                                // :153
        double n[6] = {0, 0, 0, 0, 0, 0}, r[3] = {0, 0, 0};
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
            for (int k = 0; k < 6; ++k) n[k] = fma(s[v], G[v][k], n[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) r[k] = fma(s[v], h[v][k], r[k]);
        }
        solve_sym3(n, r, x);                                                                               // :155
        bool conv = true;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv) {
                dn[v] = P[v][8] * x[0] + P[v][9] * x[1] + P[v][10] * x[2] + P[v][11];                      // :158-159
                if (!(fabs(dn[v] - d[v]) <= tol)) conv = false;                                            // :161-162
            }
        }
        if (conv) break;                                                                                   // :163
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv) {
                const double w = fast_rcp(dn[v]);                                                          // :166-169 (cumulative)
                s[v] *= w * w;
                d[v] = dn[v];
            }
        }
    }
    bool all_front = true;
    int code = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < nv) {
            if (!(dn[v] > 0)) all_front = false;
            if (dn[v] <= 0) code -= (1 << v);                                                              // :178-179
        }
    }
    return all_front ? 1 : code;
}

// Linear least squares (triangulation.py:34-97) for float32 STORAGE (round 5): the normal equations A^T A x = A^T b accumulated and solved in float64 from
// the float32 inputs.  cond(A)^2 ~ 1e8 against float64's 1e-16 leaves ~1e-8 relative -- below the float32 rounding of the result -- at a third of the
// instructions of the float32 Householder QR (whose ~730 dependent instructions per item, not HBM, bounded the bulk launch at 0.44 of peak).  The
// float64-storage path keeps the QR (1e-6 mm against cv2.solve(DECOMP_SVD) needs cond, not cond^2).
template <int NV, typename TI>
EPI_HD __forceinline__ int tri_ls_ne(const TI (&uu)[NV][2], const TI (&PP)[NV][12], int nv, double (&x)[3]) {
    double n[6] = {0, 0, 0, 0, 0, 0}, r[3] = {0, 0, 0};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < nv) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double a0, a1, a2, b;
                if constexpr (std::is_same<TI, float>::value) {
                    // float32 storage (round 6): the rows themselves in float32 with ONE rounding each (fma) -- a perturbation of a row by 2^-24 of its size,
                    // i.e. by less than the storage rounding of the key point it is built from (measured against float64 rows on 4 352 noisy 4-view joints:
                    // 7e-5 mm) -- and only the sums of their products, where cond(A)^2 bites, in float64 (a product of two floats is exact there).
                    // 32 conversions and 72 float64 FMAs per item instead of 56 and 104.
                    const float uq = uu[v][q];
                    a0 = (double)fmaf(uq, PP[v][8], -PP[v][4 * q]);
                    a1 = (double)fmaf(uq, PP[v][9], -PP[v][4 * q + 1]);
                    a2 = (double)fmaf(uq, PP[v][10], -PP[v][4 * q + 2]);
                    b = (double)fmaf(-uq, PP[v][11], PP[v][4 * q + 3]);                                        // triangulation.py:138-148
                } else {
                    const double uq = (double)uu[v][q];
                    a0 = uq * (double)PP[v][8] - (double)PP[v][4 * q]; a1 = uq * (double)PP[v][9] - (double)PP[v][4 * q + 1];
                    a2 = uq * (double)PP[v][10] - (double)PP[v][4 * q + 2];
                    b = (double)PP[v][4 * q + 3] - uq * (double)PP[v][11];                                     // triangulation.py:138-148
                }
                n[0] = fma(a0, a0, n[0]); n[1] = fma(a0, a1, n[1]); n[2] = fma(a0, a2, n[2]);
                n[3] = fma(a1, a1, n[3]); n[4] = fma(a1, a2, n[4]); n[5] = fma(a2, a2, n[5]);
                r[0] = fma(a0, b, r[0]); r[1] = fma(a1, b, r[1]); r[2] = fma(a2, b, r[2]);
            }
        }
    }
    solve_sym3(n, r, x);
    return 1;                                                                                                  // triangulation.py:97: status all True
}

// Mixed precision (round 3): the same iteration with float64 where it is needed and float32 everywhere else.
//   * round 1 (all weights 1) is the plain least-squares solve: normal equations and solve in float64 -> x0, depths d0;
//   * every later round only MOVES the solution by a few millimetres (the cumulative re-weighting shifts it between the views'
//     individually consistent points), so it is solved for the correction:  N dx = sum_v s_v g_v  with  g_v = A_v^T (b_v - A_v x0)
//     (computed once in float64: the cancellation happens there) and N = sum_v s_v G_v in float32.  cond(N) ~ 1e4 costs 1e-3 of |dx|
//     ~ mm, i.e. ~1e-3 mm, inside the 1e-2 mm envelope of the fp32-storage path;
//   * the reference's stopping rule |d_new - d_old| <= 3e-5 mm on depths of ~5000 mm (triangulation.py:161) is evaluated on the depth
//     DIFFERENCE p_v . (dx_new - dx_old), which float32 resolves to ~1e-7 mm -- the number of rounds, and with it the result of the
//     cumulative re-weighting, is the reference's.
// ~300 float64 + ~110 float32 operations per further round instead of ~150 float64 per round.
EPI_HD __forceinline__ float fast_rcpf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(v);
    return fmaf(fmaf(-v, r, 1.0f), r, r);
#else
    return 1.0f / v;
#endif
}

// (TI: the type the inputs are held in -- float for the fp32-storage kernel, so that no float64 copy of the 4 x 14 inputs occupies
//  112 registers for the whole solve; every use converts)
template <int NV, typename TI>
EPI_HD __forceinline__ int tri_iterative_mixed(const TI (&uu)[NV][2], const TI (&PP)[NV][12], int nv, double tol, int max_iter,
                                               double (&x)[3]) {
    struct { const TI (&a)[NV][2]; EPI_HD __forceinline__ double operator()(int v, int q) const { return (double)a[v][q]; } } uw{uu};
    struct { const TI (&a)[NV][12]; EPI_HD __forceinline__ double operator()(int v, int k) const { return (double)a[v][k]; } } Pw{PP};
    x[0] = x[1] = x[2] = 0;
    if (max_iter <= 0) return 1;
    double n[6] = {0, 0, 0, 0, 0, 0}, r[3] = {0, 0, 0};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < nv) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double a0 = uw(v, q) * Pw(v, 8) - Pw(v, 4 * q), a1 = uw(v, q) * Pw(v, 9) - Pw(v, 4 * q + 1), a2 = uw(v, q) * Pw(v, 10) - Pw(v, 4 * q + 2);
                const double b = Pw(v, 4 * q + 3) - uw(v, q) * Pw(v, 11);                                     // triangulation.py:138-148
                n[0] = fma(a0, a0, n[0]); n[1] = fma(a0, a1, n[1]); n[2] = fma(a0, a2, n[2]);
                n[3] = fma(a1, a1, n[3]); n[4] = fma(a1, a2, n[4]); n[5] = fma(a2, a2, n[5]);
                r[0] = fma(a0, b, r[0]); r[1] = fma(a1, b, r[1]); r[2] = fma(a2, b, r[2]);
            }
        }
    }
    solve_sym3(n, r, x);                                                                                   // round 1, :155 with unit weights
    // per view: Gram block and residual right-hand side at x0 (float32 from here on), depth row, depth at x0
    float G[NV][6], g[NV][3], pz[NV][3], d[NV], dn[NV], s[NV];
    bool conv = true;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const bool on = v < nv;
#pragma unroll
        for (int k = 0; k < 6; ++k) G[v][k] = 0.f;
        g[v][0] = g[v][1] = g[v][2] = 0.f;
        s[v] = on ? 1.f : 0.f;
        double d0 = 1.0;
        if (on) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const double a0 = uw(v, q) * Pw(v, 8) - Pw(v, 4 * q), a1 = uw(v, q) * Pw(v, 9) - Pw(v, 4 * q + 1), a2 = uw(v, q) * Pw(v, 10) - Pw(v, 4 * q + 2);
                const double e = (Pw(v, 4 * q + 3) - uw(v, q) * Pw(v, 11)) - (a0 * x[0] + a1 * x[1] + a2 * x[2]);
                const float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, fe = (float)e;
                G[v][0] = fmaf(f0, f0, G[v][0]); G[v][1] = fmaf(f0, f1, G[v][1]); G[v][2] = fmaf(f0, f2, G[v][2]);
                G[v][3] = fmaf(f1, f1, G[v][3]); G[v][4] = fmaf(f1, f2, G[v][4]); G[v][5] = fmaf(f2, f2, G[v][5]);
                g[v][0] = fmaf(f0, fe, g[v][0]); g[v][1] = fmaf(f1, fe, g[v][1]); g[v][2] = fmaf(f2, fe, g[v][2]);
            }
            d0 = Pw(v, 8) * x[0] + Pw(v, 9) * x[1] + Pw(v, 10) * x[2] + Pw(v, 11);                             // :158-159
            if (!(fabs(d0 - 1.0) <= tol)) conv = false;                                                    // :161 against the initial d = 1 (:151)
        }
        pz[v][0] = on ? (float)Pw(v, 8) : 0.f; pz[v][1] = on ? (float)Pw(v, 9) : 0.f; pz[v][2] = on ? (float)Pw(v, 10) : 0.f;
        d[v] = dn[v] = (float)d0;
    }
    float dx[3] = {0.f, 0.f, 0.f};
    if (!conv) {
        const float dref = d[0];                                                                           // keeps the cumulative weights O(1)
        for (int it = 1; it < max_iter; ++it) {                                                            // :153
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v < nv) {
                    const float w = dref * fast_rcpf(dn[v]);                                               // :166-169 (cumulative), common factor dref
                    s[v] *= w * w;
                    d[v] = dn[v];
                }
            }
            float nn[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rr[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int v = 0; v < NV; ++v) {
#pragma unroll
                for (int k = 0; k < 6; ++k) nn[k] = fmaf(s[v], G[v][k], nn[k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) rr[k] = fmaf(s[v], g[v][k], rr[k]);
            }
            const float c00 = nn[3] * nn[5] - nn[4] * nn[4], c01 = nn[2] * nn[4] - nn[1] * nn[5], c02 = nn[1] * nn[4] - nn[2] * nn[3];
            const float c11 = nn[0] * nn[5] - nn[2] * nn[2], c12 = nn[1] * nn[2] - nn[0] * nn[4], c22 = nn[0] * nn[3] - nn[1] * nn[1];
            const float inv = fast_rcpf(nn[0] * c00 + nn[1] * c01 + nn[2] * c02);
            const float nx0 = (c00 * rr[0] + c01 * rr[1] + c02 * rr[2]) * inv, nx1 = (c01 * rr[0] + c11 * rr[1] + c12 * rr[2]) * inv,
                        nx2 = (c02 * rr[0] + c12 * rr[1] + c22 * rr[2]) * inv;                             // :155
            const float m0 = nx0 - dx[0], m1 = nx1 - dx[1], m2 = nx2 - dx[2];
            dx[0] = nx0; dx[1] = nx1; dx[2] = nx2;
            conv = true;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (v < nv) {
                    const float step = pz[v][0] * m0 + pz[v][1] * m1 + pz[v][2] * m2;                      // d_new - d_old, :158-162
                    dn[v] = d[v] + step;
                    if (!(fabsf(step) <= (float)tol)) conv = false;
                }
            }
            if (conv) break;                                                                               // :163
        }
    }
    x[0] += (double)dx[0]; x[1] += (double)dx[1]; x[2] += (double)dx[2];
    bool all_front = true;
    int code = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < nv) {
            if (!(dn[v] > 0)) all_front = false;
            if (dn[v] <= 0) code -= (1 << v);                                                              // :178-179
        }
    }
    return all_front ? 1 : code;
}

// Homogeneous DLT for fp32-storage bulk batches: the right-singular vector of the smallest singular value of the 2V x 4 system
// M is the eigenvector of the smallest eigenvalue of the 4 x 4 Gram matrix M^T M -- found by inverse iteration on its (shifted)
// LDL^T factorisation instead of a Jacobi SVD of M.  The eigenvalue gap is enormous (lambda_4 / lambda_3 = noise^2), three
// solves converge to float64 round-off; squaring the condition number (1e4 -> 1e8) costs 1e-8 relative, inside the fp32 envelope.
template <int NV, typename TI = double>
EPI_HD __forceinline__ int tri_dlt_gram(const TI (&u)[NV][2], const TI (&P)[NV][12], int nv, double (&x)[3]) {
    double g[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) g[a][b] = 0.0;
    double scale2 = 0.0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < nv) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                double m[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {                                                                         // cv2.triangulatePoints rows
                    if constexpr (std::is_same<TI, float>::value) m[c] = (double)fmaf(u[v][r], P[v][8 + c], -P[v][4 * r + c]);   // (float32 storage: see tri_ls_ne)
                    else m[c] = (double)u[v][r] * (double)P[v][8 + c] - (double)P[v][4 * r + c];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = a; b < 4; ++b) g[a][b] = fma(m[a], m[b], g[a][b]);
            }
        }
    }
    scale2 = g[0][0] + g[1][1] + g[2][2] + g[3][3];
    const double shift = 1e-14 * scale2;                 // keeps the factorisation regular when the matches are exact (lambda_4 = 0)
    // LDL^T of (G + shift I), unit lower triangular L stored in the lower part
    double L[4][4], D[4], Dv[4];                     // Dv: the pivots, D: their reciprocals
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double dj = g[j][j] + shift;
#pragma unroll
        for (int k = 0; k < j; ++k) dj -= L[j][k] * L[j][k] * Dv[k];
        Dv[j] = dj;
        D[j] = fast_rcp(dj);                             // (stored inverted: every later use divides)
        const double inv = D[j];
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double t = g[j][i];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[i][k] * L[j][k] * Dv[k];
            L[i][j] = t * inv;
        }
    }
    double hv[4] = {0.5, 0.5, 0.5, 0.5};
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        double y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {                   // L y = h
            double t = hv[i];
#pragma unroll
            for (int k = 0; k < i; ++k) t -= L[i][k] * y[k];
            y[i] = t;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] *= D[i];
#pragma unroll
        for (int i = 3; i >= 0; --i) {                  // L^T z = y
            double t = y[i];
#pragma unroll
            for (int k = i + 1; k < 4; ++k) t -= L[k][i] * y[k];
            y[i] = t;
        }
        const double nrm = fast_rsq(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) hv[i] = y[i] * nrm;
    }
    const double iw = fast_rcp(hv[3]);
    x[0] = hv[0] * iw; x[1] = hv[1] * iw; x[2] = hv[2] * iw;                                               // triangulation.py:24
    const double mx = fmax(fabs(x[0]), fmax(fabs(x[1]), fabs(x[2])));
    return (mx <= 1.0e16) ? 1 : 0;                                                                         // :25
}

enum { TRI_ITER = 0, TRI_LS = 1, TRI_DLT = 2, TRI_POLY = 3 };

template <typename T, int NV, int METHOD>
__device__ __forceinline__ int triangulate_one(const T (&u)[NV][2], const T (&P)[NV][12], int nv, T tol, int max_iter, T (&x)[3]) {
    if (METHOD == TRI_ITER) return tri_iterative_ls<T, NV>(u, P, nv, tol, max_iter, x);
    if (METHOD == TRI_LS) return tri_linear_ls<T, NV>(u, P, nv, x);
    return -100;
}
template <int NV>
__device__ __forceinline__ int triangulate_one_dlt(const double (&u)[NV][2], const double (&P)[NV][12], int nv, double (&x)[3]) {
    return tri_dlt<NV>(u, P, nv, x);
}

// Arithmetic type per method for storage type S: float64 everywhere (the inputs stay in the storage type and are converted at use).  The
// iterative solver must: the reference's stopping rule compares depths (~5000 mm) with an
// absolute 3e-5 mm tolerance (triangulation.py:161), below fp32 resolution, and its result depends on running
// exactly as many re-weighting rounds as the reference does.  The homogeneous DLT needs float64 (sigma_max /
// sigma_3 ~ 1e4).
template <typename S, int METHOD> struct TriCompute { typedef double type; };
// (round 5: the float32-storage LS accumulates its normal equations in float64 too -- tri_ls_ne)

template <typename S, int NV, int METHOD>
__global__ __launch_bounds__(256, (std::is_same<S, float>::value && METHOD == TRI_DLT) ? 4 : 1) void triangulate_kernel(const S* __restrict__ kps, int kstride, const S* __restrict__ Pm,
                                                          int G, int V, int J, double tol, int max_iter, S* __restrict__ X,
                                                          int* __restrict__ status) {
    typedef typename TriCompute<S, METHOD>::type T;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)G * J;
    if (t >= total) return;
    // (round 6: a 64-bit division per item was ~100 instructions of the ~600 a linear solve takes; item counts below 2^31 divide in 32 bits)
    const int g = total < (1ll << 31) ? (int)((unsigned)t / (unsigned)J) : (int)(t / J);
    const long long view_items = total;                                 // items of one view: sample s = v * G + g, img_utils.py:197-202
    // inputs stay in the storage type for the mixed-precision iterative solver (it converts at every use: 56 registers instead of 112)
    typedef typename std::conditional<std::is_same<S, float>::value && METHOD != TRI_POLY, float, T>::type TIN;
    TIN u[NV][2], P[NV][12];
    T x[3];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < V) {
            const long long s = (long long)v * G + g;                  // img_utils.py:197-202
            const S* kp = kps + (v * view_items + t) * kstride;         // == (s * J + j) * kstride
            const S* pp = Pm + s * 12;
            u[v][0] = (TIN)kp[0]; u[v][1] = (TIN)kp[1];
#pragma unroll
            for (int k = 0; k < 12; ++k) P[v][k] = (TIN)pp[k];
        } else {
            u[v][0] = u[v][1] = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) P[v][k] = 0;
        }
    }
    int st;
    if constexpr (METHOD == TRI_POLY) {
        double F[3][3];
        fundamental_from_P(P[0], P[1], F);
        correct_match(F, u[0], u[1]);
        st = tri_dlt<NV>(u, P, V, x);
    } else if constexpr (METHOD == TRI_DLT) {
        if constexpr (std::is_same<S, float>::value) st = tri_dlt_gram<NV, TIN>(u, P, V, x);  // bulk fp32 storage: Gram + inverse iteration
        else st = tri_dlt<NV>(u, P, V, x);
    } else if constexpr (METHOD == TRI_ITER && std::is_same<S, float>::value) {
        st = tri_iterative_mixed<NV, TIN>(u, P, V, tol, max_iter, x);                           // bulk fp32 storage: float64 first round, float32 corrections
    } else if constexpr (METHOD == TRI_LS && std::is_same<S, float>::value) {
        st = tri_ls_ne<NV, TIN>(u, P, V, x);                                                    // fp32 storage: float64 normal equations
    } else st = triangulate_one<T, NV, METHOD>(u, P, V, (T)tol, max_iter, x);
    X[3 * t] = (S)x[0]; X[3 * t + 1] = (S)x[1]; X[3 * t + 2] = (S)x[2];
    if (status) status[t] = st;
}

// Bulk form of triangulate_kernel (round 5).  The per-item kernel above reads its inputs the way the reference indexes them -- 12 scalar loads of P per view
// (the J items of a group all fetch the same 48 bytes), 2 per key point, 3 scalar stores per result: ~60 vector-memory instructions per wave for 3.5 KB, which
// the address path, not HBM, bounds (LS 3.25 TB/s = 0.41 of peak, profiles/r03_microbench_tri.txt).  Here a 256-thread workgroup owns 256 consecutive items
// (group, joint) = the groups g_lo .. g_hi and
//   * stages the projection matrices of those groups, per view ONE contiguous block [g_lo .. g_hi][12], through LDS with 16-byte loads (every thread then reads
//     its group's rows as broadcast 16-byte LDS reads);
//   * reads a key point as ONE (x, y) vector load per view (kstride == 2: consecutive items are consecutive in memory, 512 bytes per wave instruction);
//   * parks the results in LDS and writes them out as 16-byte stores.
// ~7 vector-memory instructions per wave.  Same arithmetic, same results as the per-item kernel (tests/test_hip_selfsup.py runs both).
template <typename S, int NV, int METHOD>
__global__ __launch_bounds__(256, (std::is_same<S, float>::value && METHOD != TRI_POLY) ? 3 : 1)
void triangulate_staged_kernel(const S* __restrict__ kps, int kstride, const S* __restrict__ Pm, int G, int V, int J, int ng_max, double tol, int max_iter,
                               S* __restrict__ X, int* __restrict__ status) {
    typedef typename TriCompute<S, METHOD>::type T;
    typedef typename std::conditional<std::is_same<S, float>::value && METHOD != TRI_POLY, float, T>::type TIN;
    extern __shared__ __attribute__((aligned(16))) char tri_lds[];
    S* Pl = reinterpret_cast<S*>(tri_lds);                                  // [V][ng_max][12]
    S* Xl = Pl + (size_t)V * ng_max * 12;                                   // [256][3]
    const int tid = threadIdx.x;
    const long long total = (long long)G * J, t0 = (long long)blockIdx.x * 256, t = t0 + tid;
    const int n_items = (int)(total - t0 < 256 ? total - t0 : 256);
    const int g_lo = (int)(t0 / J), g_hi = (int)((t0 + n_items - 1) / J), ng = g_hi - g_lo + 1;
    constexpr int VEC = 16 / (int)sizeof(S);                                // elements per 16-byte access (12 is a multiple of it)
    const bool live = tid < n_items;
    // (group, joint) of this thread's item without a 64-bit division (round 6: ~100 of the ~600 instructions of a linear solve): the item is
    // rel = r0 + tid past the first item of group g_lo, rel < 256 + J, and rel / J = (rel * ceil(2^16 / J)) >> 16 exactly while rel * J < 2^16
    const int r0 = (int)(t0 - (long long)g_lo * J), rel = live ? r0 + tid : r0;
    const int gl = (256 + J) * J < 65536 ? (int)(((unsigned)rel * (unsigned)((65536 + J - 1) / J)) >> 16) : rel / J;
    const int g = g_lo + gl;
    const long long t_item = live ? t : t0;
    // the key points first: their latency overlaps the staging of the projection matrices
    TIN u[NV][2], P[NV][12];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < V) {
            const S* kp = kps + ((long long)v * total + t_item) * kstride;          // == (((long long)v * G + g) * J + j) * kstride
            if (kstride == 2) {
                typedef S vec2 __attribute__((ext_vector_type(2)));
                const vec2 q = *reinterpret_cast<const vec2*>(kp);
                u[v][0] = (TIN)q.x; u[v][1] = (TIN)q.y;
            } else { u[v][0] = (TIN)kp[0]; u[v][1] = (TIN)kp[1]; }
        } else u[v][0] = u[v][1] = 0;
    }
    {
        const int per_view = ng * 12 / VEC;                                 // 16-byte vectors per view block
        // one wave per view (views 4 apart share a wave): no per-element division, 64 x 16 contiguous bytes per instruction
        for (int v = tid >> 6; v < V; v += 4) {
            const S* src = Pm + ((long long)v * G + g_lo) * 12;
            S* dst = Pl + (size_t)v * ng_max * 12;
            for (int e = tid & 63; e < per_view; e += 64) *reinterpret_cast<uint4v*>(dst + (size_t)e * VEC) = *reinterpret_cast<const uint4v*>(src + (size_t)e * VEC);
        }
    }
    __syncthreads();
    T x[3] = {0, 0, 0};
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        if (v < V) {
            const S* pp = Pl + ((size_t)v * ng_max + (g - g_lo)) * 12;
#pragma unroll
            for (int k = 0; k < 12; k += VEC) {
                const uint4v w = *reinterpret_cast<const uint4v*>(pp + k);
                S tmp[VEC];
                __builtin_memcpy(tmp, &w, 16);
#pragma unroll
                for (int e = 0; e < VEC; ++e) P[v][k + e] = (TIN)tmp[e];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) P[v][k] = 0;
        }
    }
    int st;
    if constexpr (METHOD == TRI_POLY) {
        double F[3][3];
        fundamental_from_P(P[0], P[1], F);
        correct_match(F, u[0], u[1]);
        st = tri_dlt<NV>(u, P, V, x);
    } else if constexpr (METHOD == TRI_DLT) {
        if constexpr (std::is_same<S, float>::value) st = tri_dlt_gram<NV, TIN>(u, P, V, x);   // bulk fp32 storage: Gram + inverse iteration
        else st = tri_dlt<NV>(u, P, V, x);
    } else if constexpr (METHOD == TRI_ITER && std::is_same<S, float>::value) {
        st = tri_iterative_mixed<NV, TIN>(u, P, V, tol, max_iter, x);                           // bulk fp32 storage: float64 first round, float32 corrections
    } else if constexpr (METHOD == TRI_LS && std::is_same<S, float>::value) {
        st = tri_ls_ne<NV, TIN>(u, P, V, x);
    } else st = triangulate_one<T, NV, METHOD>(u, P, V, (T)tol, max_iter, x);
    Xl[3 * tid] = (S)x[0]; Xl[3 * tid + 1] = (S)x[1]; Xl[3 * tid + 2] = (S)x[2];
    if (status && live) status[t] = st;
    __syncthreads();
    {
        const int n_el = n_items * 3, n_vec = n_el / VEC;
        S* out = X + t0 * 3;                                                // (t0 * 3 * sizeof(S) is a multiple of 3072: 16-byte aligned)
        for (int i = tid; i < n_vec; i += 256) *reinterpret_cast<uint4v*>(out + (size_t)i * VEC) = *reinterpret_cast<const uint4v*>(Xl + (size_t)i * VEC);
        for (int i = n_vec * VEC + tid; i < n_el; i += 256) out[i] = Xl[i];
    }
}

// cv2.correctMatches over G fundamental matrices x J pairs each (float64).
__global__ __launch_bounds__(256) void correct_matches_kernel(const double* __restrict__ Fm, const double* __restrict__ u1,
                                                              const double* __restrict__ u2, int G, int J,
                                                              double* __restrict__ o1, double* __restrict__ o2) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)G * J) return;
    const double* f = Fm + (t / J) * 9;
    double F[3][3], a[2] = {u1[2 * t], u1[2 * t + 1]}, b[2] = {u2[2 * t], u2[2 * t + 1]};
#pragma unroll
    for (int i = 0; i < 9; ++i) F[i / 3][i % 3] = f[i];
    correct_match(F, a, b);
    o1[2 * t] = a[0]; o1[2 * t + 1] = a[1];
    o2[2 * t] = b[0]; o2[2 * t + 1] = b[1];
}

// One fundamental matrix per group from its J matches (float64, one thread per group; the 9x9 Jacobi lives in scratch:
// a few thousand groups at most, launch-latency bound).
__global__ __launch_bounds__(64) void fundamental_8point_kernel(const double* __restrict__ u1, const double* __restrict__ u2, int G, int J,
                                                                 double* __restrict__ Fm, int* __restrict__ status) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    double F[3][3];
    const int ok = fundamental_8point_one(u1 + (long long)g * J * 2, u2 + (long long)g * J * 2, J, F);
    for (int i = 0; i < 9; ++i) Fm[(long long)g * 9 + i] = F[i / 3][i % 3];
    if (status) status[g] = ok;
}

struct MetaDev {
    const double *cx, *cy, *w, *h, *scale, *rot, *R, *T, *f, *c, *P;
};

__global__ void decode_kernel(const float* __restrict__ xyz, int B, int J, MetaDev m, double pw, double ph, double rect3d,
                              double* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * J) return;
    const int n = (int)(t / J);
    Aff inv;
    patch_affines(m.cx[n], m.cy[n], m.w[n], m.h[n], m.scale[n], m.rot[n], pw, ph, &inv, nullptr);
    // integral_loss.py:196-201: float32 coords -> float64, (x+.5)*pw, (y+.5)*ph, z*pw
    const double px = ((double)xyz[3 * t] + 0.5) * pw, py = ((double)xyz[3 * t + 1] + 0.5) * ph, pz = (double)xyz[3 * t + 2] * pw;
    out[3 * t] = inv.a00 * px + inv.a01 * py + inv.a02;                         // img_utils.py:108-111
    out[3 * t + 1] = inv.a10 * px + inv.a11 * py + inv.a12;
    out[3 * t + 2] = pz / pw * rect3d;                                          // img_utils.py:154
}

// prep_h36m.py:177-204 + img_utils.py:232-238 + integral_loss.py:170-177 for one (sample, joint)
EPI_HD __forceinline__ void reproject_one(const double (&X)[3], const double (&Xroot)[3], const double* R, const double* T,
                                              const double* f, const double* c, const Aff& fwd, double scale, double pw,
                                              double ph, double rect3d, float* lab) {
    const double dx = X[0] - T[0], dy = X[1] - T[1], dz = X[2] - T[2];
    const double camx = R[0] * dx + R[1] * dy + R[2] * dz;
    const double camy = R[3] * dx + R[4] * dy + R[5] * dz;
    const double camz = R[6] * dx + R[7] * dy + R[8] * dz;
    const double rz = R[6] * (Xroot[0] - T[0]) + R[7] * (Xroot[1] - T[1]) + R[8] * (Xroot[2] - T[2]);
    const double iu = camx / camz * f[0] + c[0];                                // CamProj, prep_h36m.py:170-175
    const double iv = camy / camz * f[1] + c[1];
    const double z = camz - rz;                                                 // :200
    const double pu = fwd.a00 * iu + fwd.a01 * iv + fwd.a02;                    // img_utils.py:235
    const double pv = fwd.a10 * iu + fwd.a11 * iv + fwd.a12;
    const double pz = z / (rect3d * scale) * pw;                                // img_utils.py:236
    lab[0] = (float)(pu / pw - 0.5);                                            // integral_loss.py:171-173
    lab[1] = (float)(pv / ph - 0.5);
    lab[2] = (float)(pz / pw);
}

__global__ void reproject_kernel(const double* __restrict__ X, int G, int V, int J, MetaDev m, double pw, double ph,
                                 double rect3d, int root, float* __restrict__ label, float* __restrict__ weight) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)G * V * J) return;
    const int n = (int)(t / J), j = (int)(t - (long long)n * J);
    const int g = n % G;
    Aff fwd;
    patch_affines(m.cx[n], m.cy[n], m.w[n], m.h[n], m.scale[n], m.rot[n], pw, ph, nullptr, &fwd);
    double x[3], xr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { x[k] = X[((long long)g * J + j) * 3 + k]; xr[k] = X[((long long)g * J + root) * 3 + k]; }
    float lab[3];
    reproject_one(x, xr, m.R + 9LL * n, m.T + 3LL * n, m.f + 2LL * n, m.c + 2LL * n, fwd, m.scale[n], pw, ph, rect3d, lab);
#pragma unroll
    for (int k = 0; k < 3; ++k) { label[3 * t + k] = lab[k]; weight[3 * t + k] = 1.f; }   // img_utils.py:238 (vis = ones)
}

// img_utils.py:166-190 in one launch.  A workgroup owns GPB consecutive groups; LDS holds both crop
// affines of every sample of those groups and the triangulated joints (the root joint is needed by all).
constexpr int SS_THREADS = 256;

template <int NV, int METHOD>
__global__ __launch_bounds__(SS_THREADS) void self_supervision_kernel(const float* __restrict__ xyz, int G, int V, int J, int GPB,
                                                                      MetaDev m, double pw, double ph, double rect3d, int root,
                                                                      double tol, int max_iter, float* __restrict__ label,
                                                                      float* __restrict__ weight, double* __restrict__ Xout) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Aff* inv_s = reinterpret_cast<Aff*>(smem_raw);                  // [GPB*V]
    Aff* fwd_s = inv_s + GPB * V;                                   // [GPB*V]
    double* Xs = reinterpret_cast<double*>(fwd_s + GPB * V);        // [GPB*J*3]
    const int g0 = blockIdx.x * GPB;
    const int ng = min(GPB, G - g0);
    for (int s = threadIdx.x; s < ng * V; s += SS_THREADS) {
        const int gl = s % ng, v = s / ng;
        const long long n = (long long)v * G + g0 + gl;
        patch_affines(m.cx[n], m.cy[n], m.w[n], m.h[n], m.scale[n], m.rot[n], pw, ph, &inv_s[gl * V + v], &fwd_s[gl * V + v]);
    }
    __syncthreads();
    const int gl = threadIdx.x / J, j = threadIdx.x - gl * J;
    const bool active = gl < ng;
    if (active) {
        double u[NV][2], P[NV][12], x[3];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < V) {
                const long long n = (long long)v * G + g0 + gl;
                const float* q = xyz + (n * J + j) * 3;
                const double px = ((double)q[0] + 0.5) * pw, py = ((double)q[1] + 0.5) * ph;
                const Aff a = inv_s[gl * V + v];
                u[v][0] = a.a00 * px + a.a01 * py + a.a02;
                u[v][1] = a.a10 * px + a.a11 * py + a.a12;
                const double* pp = m.P + n * 12;
#pragma unroll
                for (int k = 0; k < 12; ++k) P[v][k] = pp[k];
            } else {
                u[v][0] = u[v][1] = 0;
#pragma unroll
                for (int k = 0; k < 12; ++k) P[v][k] = 0;
            }
        }
        if constexpr (METHOD == TRI_POLY) {
            double F[3][3];
            fundamental_from_P(P[0], P[1], F);
            correct_match(F, u[0], u[1]);
            tri_dlt<NV>(u, P, V, x);
        } else if constexpr (METHOD == TRI_DLT) tri_dlt<NV>(u, P, V, x);
        else triangulate_one<double, NV, METHOD>(u, P, V, tol, max_iter, x);
        double* xs = Xs + (gl * J + j) * 3;
        xs[0] = x[0]; xs[1] = x[1]; xs[2] = x[2];
        if (Xout) {
            double* xo = Xout + ((long long)(g0 + gl) * J + j) * 3;
            xo[0] = x[0]; xo[1] = x[1]; xo[2] = x[2];
        }
    }
    __syncthreads();
    if (active) {
        double x[3], xr[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { x[k] = Xs[(gl * J + j) * 3 + k]; xr[k] = Xs[(gl * J + root) * 3 + k]; }
        for (int v = 0; v < V; ++v) {
            const long long n = (long long)v * G + g0 + gl;
            float lab[3];
            reproject_one(x, xr, m.R + 9 * n, m.T + 3 * n, m.f + 2 * n, m.c + 2 * n, fwd_s[gl * V + v], m.scale[n], pw, ph, rect3d, lab);
            float* lo = label + (n * J + j) * 3;
            float* wo = weight + (n * J + j) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) { lo[k] = lab[k]; wo[k] = 1.f; }
        }
    }
}

static inline bool meta_ok(const epi_view_meta* m, bool need_cam, bool need_p) {
    if (!m || !m->center_x || !m->center_y || !m->width || !m->height || !m->scale || !m->rot) return false;
    if (need_cam && (!m->R || !m->T || !m->f || !m->c)) return false;
    if (need_p && !m->P) return false;
    return true;
}
static inline MetaDev to_dev(const epi_view_meta* m) {
    MetaDev d;
    d.cx = m->center_x; d.cy = m->center_y; d.w = m->width; d.h = m->height; d.scale = m->scale; d.rot = m->rot;
    d.R = m->R; d.T = m->T; d.f = m->f; d.c = m->c; d.P = m->P;
    return d;
}

static int g_tri_staged = 1;

template <typename T, int METHOD>
static int launch_tri(const void* kps, int kstride, const void* P, int G, int V, int J, double tol, int max_iter, void* X,
                      int32_t* status, hipStream_t st) {
    const long long total = (long long)G * J;
    const unsigned grid = (unsigned)((total + 255) / 256);
    // the staged (bulk) kernel: its LDS block must fit, operands must allow 16-byte accesses; g_tri_staged = 0 forces the per-item kernel (tests run both)
    const int ng_max = 255 / J + 2;
    const size_t lds = ((size_t)V * ng_max * 12 + 768) * sizeof(T);
    const bool aligned = ((reinterpret_cast<uintptr_t>(kps) | reinterpret_cast<uintptr_t>(P) | reinterpret_cast<uintptr_t>(X)) & 15u) == 0;
    // measured on MI355X, 2^20 groups x 4 views x 17 joints, float32 storage (profiles/r06_microbench_tri.txt): LS 0.217 ms staged / 0.244 per item, DLT 0.286 /
    // 0.256, iterative 0.890 / 0.753 -- all three are bound by their VALU work (+560 VALU cycles per wave from LS to DLT = +0.069 ms, exactly the chip's rate);
    // the staged form's two barriers and LDS round trips pay only for the shortest solve
    const bool want_staged = g_tri_staged == 2 || (g_tri_staged == 1 && METHOD == TRI_LS);
    if (want_staged && aligned && lds <= 65536 && total >= 256) {
#define EPI_TRI_LAUNCH(NVV)                                                                                               \
    hipLaunchKernelGGL((triangulate_staged_kernel<T, NVV, METHOD>), dim3(grid), dim3(256), lds, st, (const T*)kps, kstride, (const T*)P, \
                       G, V, J, ng_max, tol, max_iter, (T*)X, (int*)status)
        if (V == 2) EPI_TRI_LAUNCH(2);
        else if (V <= 4) EPI_TRI_LAUNCH(4);
        else EPI_TRI_LAUNCH(8);
#undef EPI_TRI_LAUNCH
        EPI_CHECK_LAUNCH();
        return EPI_OK;
    }
#define EPI_TRI_LAUNCH(NVV)                                                                                               \
    hipLaunchKernelGGL((triangulate_kernel<T, NVV, METHOD>), dim3(grid), dim3(256), 0, st, (const T*)kps, kstride, (const T*)P, \
                       G, V, J, tol, max_iter, (T*)X, (int*)status)
    if (V == 2) EPI_TRI_LAUNCH(2);
    else if (V <= 4) EPI_TRI_LAUNCH(4);
    else EPI_TRI_LAUNCH(8);
#undef EPI_TRI_LAUNCH
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

template <int METHOD>
static int tri_entry(const void* kps, int kstride, const void* P, int dtype, int G, int V, int J, double tol, int max_iter,
                     void* X, int32_t* status, epi_stream_t stream) {
    if (!kps || !P || !X || G <= 0 || J <= 0 || kstride < 2) return EPI_ERR_INVALID_ARGUMENT;
    if (V < 2 || V > 8) return EPI_ERR_UNSUPPORTED;
    if ((long long)G * J > 0x7fffffffLL * 128) return EPI_ERR_UNSUPPORTED;
    if (dtype == EPI_F64) return launch_tri<double, METHOD>(kps, kstride, P, G, V, J, tol, max_iter, X, status, (hipStream_t)stream);
    if (dtype == EPI_F32) return launch_tri<float, METHOD>(kps, kstride, P, G, V, J, tol, max_iter, X, status, (hipStream_t)stream);
    return EPI_ERR_UNSUPPORTED;
}

}  // namespace epi

using namespace epi;

// 1 (default): bulk launches (>= 256 items, 16-byte aligned operands) of the linear solve (ls) take triangulate_staged_kernel; 2: of every method;
// 0: always the per-item kernel.  Returns the previous value.
extern "C" int epi_triangulate_staged(int on) {
    const int before = g_tri_staged;
    if (on >= 0) g_tri_staged = on > 2 ? 1 : on;
    return before;
}

extern "C" int epi_triangulate_iterls(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J,
                                      double tolerance, int max_iter, void* X, int32_t* status, epi_stream_t stream) {
    if (max_iter < 1) return EPI_ERR_INVALID_ARGUMENT;
    return tri_entry<TRI_ITER>(kps, kps_stride, P, dtype, G, V, J, tolerance, max_iter, X, status, stream);
}
extern "C" int epi_triangulate_ls(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J, void* X,
                                  int32_t* status, epi_stream_t stream) {
    return tri_entry<TRI_LS>(kps, kps_stride, P, dtype, G, V, J, 0.0, 1, X, status, stream);
}
extern "C" int epi_triangulate_dlt(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J, void* X,
                                   int32_t* status, epi_stream_t stream) {
    return tri_entry<TRI_DLT>(kps, kps_stride, P, dtype, G, V, J, 0.0, 1, X, status, stream);
}

extern "C" int epi_triangulate_poly(const void* kps, int kps_stride, const void* P, int dtype, int G, int V, int J, void* X,
                                    int32_t* status, epi_stream_t stream) {
    if (V != 2) return EPI_ERR_UNSUPPORTED;                             // a two-view method (triangulation.py:184)
    return tri_entry<TRI_POLY>(kps, kps_stride, P, dtype, G, V, J, 0.0, 1, X, status, stream);
}
extern "C" int epi_correct_matches(const double* F, const double* u1, const double* u2, int G, int J, double* out1, double* out2,
                                   epi_stream_t stream) {
    if (!F || !u1 || !u2 || !out1 || !out2 || G <= 0 || J <= 0) return EPI_ERR_INVALID_ARGUMENT;
    const long long total = (long long)G * J;
    if (total > 0x7fffffffLL * 128) return EPI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(correct_matches_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, F, u1, u2,
                       G, J, out1, out2);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_fundamental_8point(const double* u1, const double* u2, int G, int J, double* F, int32_t* status, epi_stream_t stream) {
    if (!u1 || !u2 || !F || G <= 0 || J <= 0) return EPI_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(fundamental_8point_kernel, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, (hipStream_t)stream, u1, u2, G, J, F,
                       (int*)status);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_decode_to_image(const float* xyz, int B, int J, const epi_view_meta* meta_host, double patch_w,
                                   double patch_h, double rect3d, double* kps_img, epi_stream_t stream) {
    if (!xyz || !kps_img || B <= 0 || J <= 0 || !meta_ok(meta_host, false, false)) return EPI_ERR_INVALID_ARGUMENT;
    const long long total = (long long)B * J;
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, B, J,
                       to_dev(meta_host), patch_w, patch_h, rect3d, kps_img);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_reproject_labels(const double* X, int G, int V, int J, const epi_view_meta* meta_host, double patch_w,
                                    double patch_h, double rect3d, int root_joint, float* label, float* weight,
                                    epi_stream_t stream) {
    if (!X || !label || !weight || G <= 0 || V <= 0 || J <= 0 || root_joint < 0 || root_joint >= J ||
        !meta_ok(meta_host, true, false))
        return EPI_ERR_INVALID_ARGUMENT;
    const long long total = (long long)G * V * J;
    hipLaunchKernelGGL(reproject_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, G, V, J,
                       to_dev(meta_host), patch_w, patch_h, rect3d, root_joint, label, weight);
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

template <int NV>
static int launch_ss(const float* xyz, int G, int V, int J, MetaDev m, double pw, double ph, double rect3d, int root, int method,
                     double tol, int max_iter, float* label, float* weight, double* Xout, hipStream_t st) {
    const int GPB = SS_THREADS / J;
    const unsigned grid = (unsigned)((G + GPB - 1) / GPB);
    const size_t lds = (size_t)GPB * V * 2 * sizeof(Aff) + (size_t)GPB * J * 3 * sizeof(double);
#define EPI_SS_LAUNCH(M)                                                                                                   \
    hipLaunchKernelGGL((self_supervision_kernel<NV, M>), dim3(grid), dim3(SS_THREADS), lds, st, xyz, G, V, J, GPB, m, pw, ph, \
                       rect3d, root, tol, max_iter, label, weight, Xout)
    if (method == TRI_ITER) EPI_SS_LAUNCH(TRI_ITER);
    else if (method == TRI_LS) EPI_SS_LAUNCH(TRI_LS);
    else if (method == TRI_DLT) EPI_SS_LAUNCH(TRI_DLT);
    else if constexpr (NV == 2) EPI_SS_LAUNCH(TRI_POLY);
    else return EPI_ERR_UNSUPPORTED;
#undef EPI_SS_LAUNCH
    EPI_CHECK_LAUNCH();
    return EPI_OK;
}

extern "C" int epi_self_supervision(const float* xyz, int G, int V, int J, const epi_view_meta* meta_host, double patch_w,
                                    double patch_h, double rect3d, int root_joint, int method, double tolerance, int max_iter,
                                    float* label, float* weight, double* X_out, epi_stream_t stream) {
    if (!xyz || !label || !weight || G <= 0 || J <= 0 || root_joint < 0 || root_joint >= J || !meta_ok(meta_host, true, true))
        return EPI_ERR_INVALID_ARGUMENT;
    if (method < 0 || method > 3 || max_iter < 1) return EPI_ERR_INVALID_ARGUMENT;
    if (V < 2 || V > 8 || J > SS_THREADS || (method == TRI_POLY && V != 2)) return EPI_ERR_UNSUPPORTED;
    const MetaDev m = to_dev(meta_host);
    hipStream_t st = (hipStream_t)stream;
    if (V == 2) return launch_ss<2>(xyz, G, V, J, m, patch_w, patch_h, rect3d, root_joint, method, tolerance, max_iter, label, weight, X_out, st);
    if (V <= 4) return launch_ss<4>(xyz, G, V, J, m, patch_w, patch_h, rect3d, root_joint, method, tolerance, max_iter, label, weight, X_out, st);
    return launch_ss<8>(xyz, G, V, J, m, patch_w, patch_h, rect3d, root_joint, method, tolerance, max_iter, label, weight, X_out, st);
}
