// TEST INFRASTRUCTURE (CPU suite): compiles the float64 per-thread device math of csrc/selfsup.hip for the HOST so that
// `pytest -m "not gpu"` can compare it with the oracle without a GPU.  Built by tests/hostcheck/build.py into
// tests/hostcheck/_build/libhostcheck.so; never linked into, loaded by, or shipped with the product library.
#include "../../epipolarpose_amd/csrc/selfsup.hip"
#include "../../epipolarpose_amd/csrc/fundamental.hip"

// csrc/fundamental.hip: the symmetric epipolar error of the LMedS scoring kernels, one pair at a time
extern "C" void hostcheck_fm_errors(const double* F, const double* u1, const double* u2, int n, float* err) {
    for (int i = 0; i < n; ++i) err[i] = epi::fm_error(F, u1[2 * i], u1[2 * i + 1], u2[2 * i], u2[2 * i + 1]);
}

extern "C" void hostcheck_correct_matches(const double* F, const double* u1, const double* u2, int n, double* o1, double* o2) {
    double Fm[3][3];
    for (int i = 0; i < 9; ++i) Fm[i / 3][i % 3] = F[i];
    for (int i = 0; i < n; ++i) {
        double a[2] = {u1[2 * i], u1[2 * i + 1]}, b[2] = {u2[2 * i], u2[2 * i + 1]};
        epi::correct_match(Fm, a, b);
        o1[2 * i] = a[0]; o1[2 * i + 1] = a[1]; o2[2 * i] = b[0]; o2[2 * i + 1] = b[1];
    }
}

extern "C" void hostcheck_fundamental(const double* P1, const double* P2, double* F) {
    double a[12], b[12], Fm[3][3];
    for (int i = 0; i < 12; ++i) { a[i] = P1[i]; b[i] = P2[i]; }
    epi::fundamental_from_P(a, b, Fm);
    for (int i = 0; i < 9; ++i) F[i] = Fm[i / 3][i % 3];
}

extern "C" void hostcheck_poly_triangulate(const double* u1, const double* u2, const double* P1, const double* P2, int n, double* X,
                                           int* status) {
    double P[2][12], Fm[3][3];
    for (int i = 0; i < 12; ++i) { P[0][i] = P1[i]; P[1][i] = P2[i]; }
    epi::fundamental_from_P(P[0], P[1], Fm);
    for (int i = 0; i < n; ++i) {
        double u[2][2] = {{u1[2 * i], u1[2 * i + 1]}, {u2[2 * i], u2[2 * i + 1]}}, x[3];
        epi::correct_match(Fm, u[0], u[1]);
        status[i] = epi::tri_dlt<2>(u, P, 2, x);
        X[3 * i] = x[0]; X[3 * i + 1] = x[1]; X[3 * i + 2] = x[2];
    }
}

extern "C" int hostcheck_real_roots6(const double* c, double* roots) {
    double cc[7], r[6];
    for (int i = 0; i < 7; ++i) cc[i] = c[i];
    const int n = epi::real_roots_unit<6>(cc, r);
    for (int i = 0; i < n; ++i) roots[i] = r[i];
    return n;
}

extern "C" void hostcheck_svd3(const double* A, double* U, double* s, double* V) {
    double a[3][3], u[3][3], v[3][3], sv[3];
    for (int i = 0; i < 9; ++i) a[i / 3][i % 3] = A[i];
    epi::svd3(a, u, sv, v);
    for (int i = 0; i < 9; ++i) { U[i] = u[i / 3][i % 3]; V[i] = v[i / 3][i % 3]; }
    for (int i = 0; i < 3; ++i) s[i] = sv[i];
}

extern "C" int hostcheck_fundamental_8point(const double* u1, const double* u2, int n, double* F) {
    double Fm[3][3];
    const int ok = epi::fundamental_8point_one(u1, u2, n, Fm);
    for (int i = 0; i < 9; ++i) F[i] = Fm[i / 3][i % 3];
    return ok;
}

// ---- two-view triangulators (the reference's V = 2 case): method 0 iterative LS, 1 linear LS, 2 DLT; float64 ----
extern "C" void hostcheck_triangulate2(int method, const double* u1, const double* u2, const double* P1, const double* P2, int n,
                                       double tol, int max_iter, double* X, int* status) {
    double P[2][12];
    for (int i = 0; i < 12; ++i) { P[0][i] = P1[i]; P[1][i] = P2[i]; }
    for (int i = 0; i < n; ++i) {
        double u[2][2] = {{u1[2 * i], u1[2 * i + 1]}, {u2[2 * i], u2[2 * i + 1]}}, x[3];
        int st;
        if (method == 0) st = epi::tri_iterative_ls<double, 2>(u, P, 2, tol, max_iter, x);
        else if (method == 1) st = epi::tri_linear_ls<double, 2>(u, P, 2, x);
        else if (method == 2) st = epi::tri_dlt<2>(u, P, 2, x);
        else if (method == 3) st = epi::tri_iterative_mixed<2, double>(u, P, 2, tol, max_iter, x);      // the bulk (fp32-storage) variants, fed float64 inputs here
        else if (method == 4) st = epi::tri_dlt_gram<2>(u, P, 2, x);
        else if (method == 6) st = epi::tri_ls_ne<2, double>(u, P, 2, x);                                  // round 5: the float32-storage LS (float64 normal equations)
        else if (method == 7 || method == 8) {
            // round 6: the float32-storage instantiations themselves (rows built in float32, sums of products in float64) on the inputs ROUNDED to float32,
            // as the bulk kernels hold them
            float uf[2][2], Pf[2][12];
            for (int v = 0; v < 2; ++v) {
                uf[v][0] = (float)u[v][0]; uf[v][1] = (float)u[v][1];
                for (int k = 0; k < 12; ++k) Pf[v][k] = (float)P[v][k];
            }
            st = method == 7 ? epi::tri_ls_ne<2, float>(uf, Pf, 2, x) : epi::tri_dlt_gram<2, float>(uf, Pf, 2, x);
        }
        else st = epi::tri_iterative_ne<2>(u, P, 2, tol, max_iter, x);
        X[3 * i] = x[0]; X[3 * i + 1] = x[1]; X[3 * i + 2] = x[2];
        status[i] = st;
    }
}

// ---- patch <-> image affines, decode (patch -> image) and re-projection (world -> label) for one sample ----
extern "C" void hostcheck_patch_affines(double cx, double cy, double bw, double bh, double scale, double rot, double pw, double ph,
                                        double* inv6, double* fwd6) {
    epi::Aff inv, fwd;
    epi::patch_affines(cx, cy, bw, bh, scale, rot, pw, ph, &inv, &fwd);
    const double a[6] = {inv.a00, inv.a01, inv.a02, inv.a10, inv.a11, inv.a12}, b[6] = {fwd.a00, fwd.a01, fwd.a02, fwd.a10, fwd.a11, fwd.a12};
    for (int i = 0; i < 6; ++i) { inv6[i] = a[i]; fwd6[i] = b[i]; }
}

extern "C" void hostcheck_reproject(const double* X, int J, int root, const double* R, const double* T, const double* f, const double* c,
                                    double cx, double cy, double bw, double bh, double scale, double rot, double pw, double ph,
                                    double rect3d, float* label) {
    epi::Aff fwd;
    epi::patch_affines(cx, cy, bw, bh, scale, rot, pw, ph, nullptr, &fwd);
    const double Xr[3] = {X[3 * root], X[3 * root + 1], X[3 * root + 2]};
    for (int j = 0; j < J; ++j) {
        const double Xj[3] = {X[3 * j], X[3 * j + 1], X[3 * j + 2]};
        epi::reproject_one(Xj, Xr, R, T, f, c, fwd, scale, pw, ph, rect3d, label + 3 * j);
    }
}
