"""Oracle (test infrastructure): multi-view triangulation in float64 NumPy.

Restates ``/root/reference/lib/utils/triangulation.py``.  The reference delegates its linear
algebra to OpenCV (not vendored; conda pin opencv=4.1.0, environment.yml:96):

* ``cv2.solve(A, b, x, cv2.DECOMP_SVD)`` (triangulation.py:95,155) -- minimum-norm least squares
  through the SVD (OpenCV ``cv::solve`` -> ``SVD::backSubst``).  Restated with ``np.linalg.lstsq``.
* ``cv2.triangulatePoints`` (triangulation.py:22) -- per point the 4x4 matrix with rows
  ``x*P[2]-P[0], y*P[2]-P[1]`` for both cameras, SVD, right-singular vector of the smallest singular
  value (OpenCV ``cvTriangulatePoints``).  Restated with ``np.linalg.svd`` and generalised to V views.

Not imported by the product.
"""
import numpy as np


def lstsq_svd(a, b):
    """Stand-in for cv2.solve(A, b, dst, DECOMP_SVD): min-norm LS via SVD, float64."""
    return np.linalg.lstsq(a, b, rcond=None)[0]


def _ls_rows(u, p):
    """Rows of the inhomogeneous system for one camera.

    triangulation.py:138-148: C = [[-1,0,u],[0,-1,v]];  A = C @ P[:, :3];  b = -(C @ P[:, 3]).
    """
    c = np.array([[-1.0, 0.0, u[0]], [0.0, -1.0, u[1]]])
    return c @ p[:, :3], -(c @ p[:, 3])


def linear_ls_triangulation(us, ps):
    """triangulation.py:34-97 generalised from 2 to V views.

    us: [V, N, 2] pixel coordinates, ps: [V, 3, 4].  Returns X [N, 3], status ones [N] (bool).
    """
    us = np.asarray(us, np.float64)
    ps = np.asarray(ps, np.float64)
    n_view, n_pt = us.shape[0], us.shape[1]
    out = np.zeros((n_pt, 3))
    for i in range(n_pt):
        rows = [_ls_rows(us[v, i], ps[v]) for v in range(n_view)]
        a = np.concatenate([r[0] for r in rows], axis=0)
        b = np.concatenate([r[1] for r in rows], axis=0)
        out[i] = lstsq_svd(a, b)
    return out, np.ones(n_pt, dtype=bool)


def iterative_ls_triangulation(us, ps, tolerance=3.0e-5, max_iter=10):
    """triangulation.py:104-181 (Hartley & Sturm iterative re-weighting), V views (reference: V=2).

    Exactly as the reference: depths start at 1 (:151); up to 10 solves (:153); stop when EVERY
    view's depth moved by <= tolerance (absolute, :161-163); otherwise rows of A and b are scaled by
    1/d_new CUMULATIVELY (:166-169).  Status (:176-179): 1 inlier / 0 not converged, minus 1 if
    d_1<=0, minus 2 if d_2<=0 (the ``i < 10`` test at :176 is always true, so "converged" only
    means "in front of all cameras").  For V>2 status = 1 if all depths > 0 else minus the bit mask
    sum(2**v for d_v <= 0), which reduces to the reference's table at V=2.
    """
    us = np.asarray(us, np.float64)
    ps = np.asarray(ps, np.float64)
    n_view, n_pt = us.shape[0], us.shape[1]
    out = np.zeros((n_pt, 3))
    status = np.zeros(n_pt, dtype=np.int64)
    for i in range(n_pt):
        rows = [_ls_rows(us[v, i], ps[v]) for v in range(n_view)]
        a = np.concatenate([r[0] for r in rows], axis=0)
        b = np.concatenate([r[1] for r in rows], axis=0)
        d = np.ones(n_view)
        x = np.zeros(3)
        d_new = d
        for _ in range(max_iter):
            x = lstsq_svd(a, b)
            d_new = ps[:, 2, :3] @ x + ps[:, 2, 3]          # :158-159
            if np.all(np.abs(d_new - d) <= tolerance):       # :161-163
                break
            w = np.repeat(1.0 / d_new, 2)                    # :166-169
            a = a * w[:, None]
            b = b * w
            d = d_new
        out[i] = x
        behind = d_new <= 0
        status[i] = 1 if np.all(d_new > 0) else -int(sum(2 ** v for v in range(n_view) if behind[v]))
    return out, status


def dlt_triangulation(us, ps, max_coordinate_value=1.0e16):
    """triangulation.py:8-27 (linear-eigen == cv2.triangulatePoints), generalised to V views.

    Homogeneous 2Vx4 system, smallest right-singular vector, dehomogenise (:24); status is the
    finiteness test of :25.
    """
    us = np.asarray(us, np.float64)
    ps = np.asarray(ps, np.float64)
    n_view, n_pt = us.shape[0], us.shape[1]
    out = np.zeros((n_pt, 3))
    for i in range(n_pt):
        m = np.empty((2 * n_view, 4))
        for v in range(n_view):
            m[2 * v] = us[v, i, 0] * ps[v, 2] - ps[v, 0]
            m[2 * v + 1] = us[v, i, 1] * ps[v, 2] - ps[v, 1]
        vt = np.linalg.svd(m)[2]
        h = vt[-1]
        out[i] = h[:3] / h[3]
    with np.errstate(invalid="ignore"):
        ok = np.max(np.abs(out), axis=1) <= max_coordinate_value
    return out, ok


def triangulate_pairs(kps, projection_matrices, n_view=2, method="iterative"):
    """img_utils.py:193-209 ``triangulate``: sample i of view v sits at batch index v*G + i.

    kps: [B, J, >=2], projection_matrices: [B, 3, 4].  Returns [B, J, 3] -- the group's 3-D pose is
    repeated for each of its views (np.vstack at :208).
    """
    kps = np.asarray(kps, np.float64)
    pm = np.asarray(projection_matrices, np.float64)
    n_group = kps.shape[0] // n_view
    fn = {"iterative": iterative_ls_triangulation, "ls": linear_ls_triangulation, "dlt": dlt_triangulation,
          "poly": polynomial_triangulation}[method]
    pts = []
    for g in range(n_group):
        idx = [v * n_group + g for v in range(n_view)]
        x, _ = fn(kps[idx][:, :, 0:2], pm[idx])
        pts.append(x)
    pts = np.asarray(pts)
    return np.concatenate([pts] * n_view, axis=0)


# --------------------------------------------------------------------------------------------------------------------
# Polynomial ("optimal", Hartley & Sturm) two-view triangulation -- triangulation.py:184-220.
# The reference delegates the correction of the matches to OpenCV (cv2.correctMatches, calib3d triangulate.cpp, which
# implements Hartley & Zisserman Algorithm 12.1).  Restated here from the published algorithm; validated by its defining
# properties (corrected pairs satisfy the epipolar constraint exactly and minimise the summed squared displacement).
# --------------------------------------------------------------------------------------------------------------------
def fundamental_from_projections(p1, p2):
    """triangulation.py:196-204: P_canon = P2_full inv(P1_full);  F = [t]_x R  (np.cross(t, R, axisb=0).T)."""
    p1f, p2f = np.eye(4), np.eye(4)
    p1f[0:3, :] = np.asarray(p1, np.float64)[0:3, :]
    p2f[0:3, :] = np.asarray(p2, np.float64)[0:3, :]
    pc = p2f @ np.linalg.inv(p1f)
    return np.cross(pc[0:3, 3], pc[0:3, 0:3], axisb=0).T


def correct_matches(f, u1, u2):
    """cv2.correctMatches(F, points1, points2): move each pair (x1, x2) by the smallest amount (sum of squared distances)
    onto a pair that satisfies x2^T F x1 = 0 exactly.  HZ Algorithm 12.1, steps (i)-(x).  u1, u2: [N, 2] -> ([N,2], [N,2])."""
    f = np.asarray(f, np.float64)
    u1, u2 = np.asarray(u1, np.float64), np.asarray(u2, np.float64)
    o1, o2 = np.empty_like(u1), np.empty_like(u2)
    for i in range(u1.shape[0]):
        t1 = np.array([[1, 0, u1[i, 0]], [0, 1, u1[i, 1]], [0, 0, 1.0]])       # T^-1: takes the origin to the point
        t2 = np.array([[1, 0, u2[i, 0]], [0, 1, u2[i, 1]], [0, 0, 1.0]])
        ft = t2.T @ f @ t1                                                      # (ii) F <- T2^-T F T1^-1
        uu, _, vt = np.linalg.svd(ft)
        e1, e2 = vt[2], uu[:, 2]                                                # (iii) F e1 = 0, e2^T F = 0
        e1 = e1 / np.hypot(e1[0], e1[1])
        e2 = e2 / np.hypot(e2[0], e2[1])
        r1 = np.array([[e1[0], e1[1], 0], [-e1[1], e1[0], 0], [0, 0, 1.0]])     # (iv)
        r2 = np.array([[e2[0], e2[1], 0], [-e2[1], e2[0], 0], [0, 0, 1.0]])
        fr = r2 @ ft @ r1.T                                                     # (v)
        a, b, c, d, f1, f2 = fr[1, 1], fr[1, 2], fr[2, 1], fr[2, 2], e1[2], e2[2]   # (vi)
        coeffs = polynomial_g(a, b, c, d, f1, f2)                               # (vii) g(t), degree 6, highest power first
        roots = np.roots(coeffs)
        cand = list(np.real(roots))                                             # (viii) cost at the real part of every root ...

        def cost(t):
            return t * t / (1 + f1 * f1 * t * t) + (c * t + d) ** 2 / ((a * t + b) ** 2 + f2 * f2 * (c * t + d) ** 2)
        costs = [cost(t) for t in cand]
        c_inf = 1.0 / (f1 * f1) + c * c / (a * a + f2 * f2 * c * c) if f1 != 0 else np.inf   # ... and at t = infinity
        k = int(np.argmin(costs))
        if c_inf < costs[k]:
            l1 = np.array([f1, 0.0, -1.0])                                      # lambda(t)/t and lambda'(t)/t as t -> infinity
            l2 = np.array([-f2 * c, a, c])
        else:
            t = cand[k]
            l1 = np.array([t * f1, 1.0, -t])                                    # (ix)
            l2 = np.array([-f2 * (c * t + d), a * t + b, c * t + d])
        x1 = np.array([-l1[0] * l1[2], -l1[1] * l1[2], l1[0] ** 2 + l1[1] ** 2])    # closest point of a line to the origin
        x2 = np.array([-l2[0] * l2[2], -l2[1] * l2[2], l2[0] ** 2 + l2[1] ** 2])
        x1 = t1 @ r1.T @ x1                                                     # (x) back to the original frames
        x2 = t2 @ r2.T @ x2
        o1[i], o2[i] = x1[:2] / x1[2], x2[:2] / x2[2]
    return o1, o2


def polynomial_g(a, b, c, d, f1, f2):
    """g(t) = t((at+b)^2 + f2^2 (ct+d)^2)^2 - (ad-bc)(1+f1^2 t^2)^2 (at+b)(ct+d)   (HZ eq. 12.7), coefficients t^6 ... t^0."""
    p = np.polynomial.polynomial
    atb, ctd = np.array([b, a]), np.array([d, c])                               # ascending powers
    q = p.polyadd(p.polymul(atb, atb), f2 * f2 * p.polymul(ctd, ctd))
    g = p.polysub(p.polymul([0.0, 1.0], p.polymul(q, q)),
                  (a * d - b * c) * p.polymul(p.polymul([1.0, 0.0, f1 * f1], [1.0, 0.0, f1 * f1]), p.polymul(atb, ctd)))
    g = np.concatenate([g, np.zeros(7 - len(g))])
    return g[::-1]


def polynomial_triangulation(us, ps):
    """triangulation.py:184-220 (two views): F from the projection matrices, optimal correction of the matches, then the
    linear-eigen (DLT) triangulation of the corrected points."""
    us = np.asarray(us, np.float64)
    ps = np.asarray(ps, np.float64)
    assert us.shape[0] == 2, "polynomial triangulation is a two-view method"
    f = fundamental_from_projections(ps[0], ps[1])
    c1, c2 = correct_matches(f, us[0], us[1])
    return dlt_triangulation(np.stack([c1, c2]), ps)


# --------------------------------------------------------------------------------------------------------------------
# Fundamental matrix from 2-D matches: the reference calls cv2.findFundamentalMat(u1, u2, cv2.FM_8POINT)
# (triangulation.py:216, the fall-back of polynomial_triangulation) and cv2.FM_LMEDS (cameras.py:136-143).  OpenCV is not
# vendored; this restates the published normalised 8-point algorithm as OpenCV's run8Point implements it (calib3d
# fundam.cpp): isotropic normalisation (centroid, mean distance sqrt 2), the eigenvector of A^T A with the smallest
# eigenvalue, rank-2 enforcement by SVD, de-normalisation, scaling to F[2,2] = 1.  Convention: x2^T F x1 = 0.
# LMedS (randomised sampling with OpenCV's internal RNG) is not reproducible and is not restated; on outlier-free matches
# it converges to the same F.  Parity unpinned at the OpenCV layer (see oracle/__init__.py).
# --------------------------------------------------------------------------------------------------------------------
def fundamental_8point(u1, u2):
    """u1, u2 [N, 2] (N >= 8) -> (F [3, 3], ok).  ok False: degenerate input (all points coincide)."""
    u1, u2 = np.asarray(u1, np.float64), np.asarray(u2, np.float64)
    n = u1.shape[0]
    if n < 8:
        return np.zeros((3, 3)), False
    c1, c2 = u1.mean(0), u2.mean(0)
    d1 = np.hypot(*(u1 - c1).T).sum() / n
    d2 = np.hypot(*(u2 - c2).T).sum() / n
    if d1 < np.finfo(np.float32).eps or d2 < np.finfo(np.float32).eps:          # FLT_EPSILON on the mean distance
        return np.zeros((3, 3)), False
    s1, s2 = np.sqrt(2.0) / d1, np.sqrt(2.0) / d2
    a, b = (u1 - c1) * s1, (u2 - c2) * s2
    rows = np.stack([b[:, 0] * a[:, 0], b[:, 0] * a[:, 1], b[:, 0], b[:, 1] * a[:, 0], b[:, 1] * a[:, 1], b[:, 1], a[:, 0], a[:, 1],
                     np.ones(n)], axis=1)
    w, v = np.linalg.eigh(rows.T @ rows)
    f0 = v[:, 0].reshape(3, 3)                                  # smallest eigenvalue
    uu, sv, vt = np.linalg.svd(f0)
    sv[2] = 0.0
    f0 = uu @ np.diag(sv) @ vt
    t1 = np.array([[s1, 0, -s1 * c1[0]], [0, s1, -s1 * c1[1]], [0, 0, 1.0]])
    t2 = np.array([[s2, 0, -s2 * c2[0]], [0, s2, -s2 * c2[1]], [0, 0, 1.0]])
    f = t2.T @ f0 @ t1
    if abs(f[2, 2]) > np.finfo(np.float32).eps:
        f = f / f[2, 2]
    return f, True


def essential_from_fundamental(f, k1, k2=None):
    """cameras.py:133-134: E = K^T F K (the reference uses one camera matrix for both views; k2 generalises it: E = K2^T F K1)."""
    k1 = np.asarray(k1, np.float64)
    k2 = k1 if k2 is None else np.asarray(k2, np.float64)
    return k2.T @ np.asarray(f, np.float64) @ k1


# ---- cv2.findFundamentalMat(u1, u2, cv2.FM_LMEDS) (cameras.py:136-143) ---------------------------------------------------------------
# The algorithm lives in OpenCV (conda pin 4.1.0; absent here): modules/calib3d/src/fundam.cpp (findFundamentalMat, run7Point,
# FMEstimatorCallback), modules/calib3d/src/ptsetreg.cpp (LMeDSPointSetRegistrator::run, getSubset, RANSACUpdateNumIters),
# modules/core (cv::RNG, solveCubic).  Restated from its published source; parity with OpenCV itself is UNPINNED (no cv2 in the image).
# What makes it deterministic is OpenCV's own design: the sampler is a fixed-seed generator, `RNG rng((uint64)-1)`.
CV_RNG_COEFF = 4164903690


class CvRNG:
    """cv::RNG: multiply-with-carry, state 64 bit; next() returns the low 32 bits of the new state."""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state if state else 0xFFFFFFFF

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * CV_RNG_COEFF + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform_int(self, a, b):
        return a if a == b else int(self.next() % (b - a) + a)


def ransac_update_num_iters(p, ep, model_points, max_iters):
    """ptsetreg.cpp RANSACUpdateNumIters."""
    p, ep = min(max(p, 0.0), 1.0), min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, np.finfo(np.float64).tiny)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < np.finfo(np.float64).tiny:
        return 0
    num, denom = np.log(num), np.log(denom)
    return max_iters if denom >= 0 or -num >= max_iters * (-denom) else int(np.rint(num / denom))


def _have_collinear_points(pts, count):
    """fundam.cpp haveCollinearPoints: is the LAST selected point on a line through two earlier ones (or on top of one)?"""
    i = count - 1
    eps = float(np.finfo(np.float32).eps)
    for j in range(i):
        dx1, dy1 = float(pts[j][0]) - float(pts[i][0]), float(pts[j][1]) - float(pts[i][1])
        for k in range(j):
            dx2, dy2 = float(pts[k][0]) - float(pts[i][0]), float(pts[k][1]) - float(pts[i][1])
            if abs(dx2 * dy1 - dy2 * dx1) <= eps * (abs(dx1) + abs(dy1) + abs(dx2) + abs(dy2)):
                return True
    return False


def fm_get_subset(rng, m1, m2, model_points=7, max_attempts=1000):
    """ptsetreg.cpp getSubset: model_points DISTINCT indices (a duplicate is redrawn), accepted when neither sample is degenerate."""
    count = len(m1)
    for _ in range(max_attempts):
        idx = []
        for _i in range(model_points):
            v = rng.uniform_int(0, count)
            while v in idx:
                v = rng.uniform_int(0, count)
            idx.append(v)
        if not _have_collinear_points(m1[idx], model_points) and not _have_collinear_points(m2[idx], model_points):
            return idx
    return None


def fm_run_7point(s1, s2):
    """fundam.cpp run7Point: the matrices of the two-dimensional null space of the 7 epipolar equations with det = 0, F[2][2] = 1.
    (Any basis of the null space gives the same set of matrices; OpenCV takes the last two right-singular vectors.)"""
    x0, y0, x1, y1 = s1[:, 0].astype(np.float64), s1[:, 1].astype(np.float64), s2[:, 0].astype(np.float64), s2[:, 1].astype(np.float64)
    a = np.stack([x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, np.ones(7)], axis=1)
    vt = np.linalg.svd(a, full_matrices=True)[2]
    f1, f2 = vt[7].copy(), vt[8].copy()
    f1 -= f2                                          # f = lambda * f1 + f2 after this line, lambda = the original mixing weight
    det = lambda m: float(np.linalg.det(m.reshape(3, 3)))
    # the cubic det(lambda * f1 + f2) through four samples of lambda (exact for a cubic)
    lam = np.array([-1.0, 0.0, 1.0, 2.0])
    vals = np.array([det(l * f1 + f2) for l in lam])
    coeffs = np.linalg.solve(np.vander(lam, 4), vals)       # c3 l^3 + c2 l^2 + c1 l + c0
    roots = np.roots(coeffs) if abs(coeffs[0]) > 0 else np.roots(coeffs[1:])
    out = []
    for r in roots:
        if abs(r.imag) > 1e-9 * max(1.0, abs(r.real)):
            continue
        lam_k, mu = float(r.real), 1.0
        s = f1[8] * lam_k + f2[8]
        f = np.empty(9)
        if abs(s) > np.finfo(np.float64).eps:
            mu = 1.0 / s
            lam_k *= mu
            f[8] = 1.0
        else:
            f[8] = 0.0
        f[:8] = f1[:8] * lam_k + f2[:8] * mu
        out.append(f.reshape(3, 3))
    return out


def fm_compute_error(f, m1, m2):
    """fundam.cpp FMEstimatorCallback::computeError: max of the two squared point-to-epipolar-line distances, stored as float32."""
    f = np.asarray(f, np.float64)
    x1, y1, x2, y2 = (m1[:, 0].astype(np.float64), m1[:, 1].astype(np.float64), m2[:, 0].astype(np.float64), m2[:, 1].astype(np.float64))
    a = f[0, 0] * x1 + f[0, 1] * y1 + f[0, 2]
    b = f[1, 0] * x1 + f[1, 1] * y1 + f[1, 2]
    c = f[2, 0] * x1 + f[2, 1] * y1 + f[2, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        s2 = 1.0 / (a * a + b * b)
        d2 = x2 * a + y2 * b + c
        a = f[0, 0] * x2 + f[1, 0] * y2 + f[2, 0]
        b = f[0, 1] * x2 + f[1, 1] * y2 + f[2, 1]
        c = f[0, 2] * x2 + f[1, 2] * y2 + f[2, 2]
        s1 = 1.0 / (a * a + b * b)
        d1 = x1 * a + y1 * b + c
        e1, e2 = d1 * d1 * s1, d2 * d2 * s2
    return np.where(e1 < e2, e2, e1).astype(np.float32)       # std::max(e1, e2)


def fm_median(err):
    """LMeDSPointSetRegistrator: the sorted errors' middle element, or the mean of the two middle ones (float sum, then * 0.5)."""
    e = np.sort(np.asarray(err, np.float32))
    n = len(e)
    return float(e[n // 2]) if n % 2 else float(np.float32(e[n // 2 - 1] + e[n // 2])) * 0.5


def fundamental_lmeds(u1, u2, confidence=0.99, max_iters=1000):
    """cv2.findFundamentalMat(u1, u2, cv2.FM_LMEDS) -> (F [3, 3] float64 or None, mask uint8 [N]).  u1, u2 [N, 2]; OpenCV converts the
    points to float32 first (the reference hands it int32 pixel coordinates, cameras.py:137-138)."""
    m1, m2 = np.asarray(u1, np.float32).reshape(-1, 2), np.asarray(u2, np.float32).reshape(-1, 2)
    count, model_points = len(m1), 7
    if count < model_points:
        return None, np.zeros(count, np.uint8)
    niters = max(ransac_update_num_iters(confidence, 0.45, model_points, max_iters), 3)
    rng = CvRNG(0xFFFFFFFFFFFFFFFF)
    best, min_median = None, np.inf
    for it in range(niters if count > model_points else 1):
        if count > model_points:
            idx = fm_get_subset(rng, m1, m2, model_points)
            if idx is None:
                if it == 0:
                    return None, np.zeros(count, np.uint8)
                break
            s1, s2 = m1[idx], m2[idx]
        else:
            s1, s2 = m1, m2
        for f in fm_run_7point(s1, s2):
            median = fm_median(fm_compute_error(f, m1, m2))
            if median < min_median:
                min_median, best = median, f
    if best is None:
        return None, np.zeros(count, np.uint8)
    with np.errstate(divide="ignore", invalid="ignore"):     # count == 7: 5 / 0 = inf as in C++ (every point an inlier; inf * 0 = NaN -> 0.001)
        sigma = 2.5 * 1.4826 * (1.0 + np.float64(5.0) / np.float64(count - model_points)) * np.sqrt(min_median)
    sigma = sigma if sigma > 0.001 else 0.001                # MAX(sigma, 0.001)
    thresh = np.float32(sigma * sigma)
    mask = (fm_compute_error(best, m1, m2) <= thresh).astype(np.uint8)
    if int(mask.sum()) < model_points:
        return None, np.zeros(count, np.uint8)
    return best, mask
