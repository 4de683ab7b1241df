"""Two-view (and V-view) triangulation -- mirror of the reference's ``lib/utils/triangulation.py``.

Same call signatures (``f(u1, P1, u2, P2) -> (X [N,3] float64, status [N])``), arithmetic on the GPU
(``epi_triangulate_*``; float64, one thread per point) instead of per-point ``cv2.solve`` calls.
"""
import numpy as np
import torch

from .. import hip

output_dtype = float


def set_triangl_output_dtype(output_dtype_):
    """triangulation.py:226-231."""
    global output_dtype
    output_dtype = output_dtype_


def triangulate_views(us, ps, method="iterative", tolerance=3.e-5, max_iter=10, device=None):
    """us [V,N,2], ps [V,3,4] (array-likes) -> (X [N,3] float64 ndarray, status [N] int32 ndarray)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    us = torch.as_tensor(np.asarray(us, dtype=np.float64), device=device)
    ps = torch.as_tensor(np.asarray(ps, dtype=np.float64), device=device)
    n_view, n_pt = us.shape[0], us.shape[1]
    # one "group" whose J axis carries the N points: batch index = view
    x, st = hip.triangulate(us.reshape(n_view, n_pt, 2), ps.reshape(n_view, 3, 4), n_view, method, tolerance, max_iter)
    return x.reshape(n_pt, 3).cpu().numpy().astype(output_dtype), st.reshape(n_pt).cpu().numpy()


def iterative_LS_triangulation(u1, P1, u2, P2, tolerance=3.e-5):
    """triangulation.py:104-181.  status: 1 / 0 / -1 / -2 / -3 as the reference."""
    x, st = triangulate_views(np.stack([u1, u2]), np.stack([P1[0:3, 0:4], P2[0:3, 0:4]]), "iterative", tolerance)
    return x, st.astype(int)


def linear_LS_triangulation(u1, P1, u2, P2):
    """triangulation.py:34-97."""
    x, _ = triangulate_views(np.stack([u1, u2]), np.stack([P1[0:3, 0:4], P2[0:3, 0:4]]), "ls")
    return x, np.ones(len(u1), dtype=bool)


def linear_eigen_triangulation(u1, P1, u2, P2, max_coordinate_value=1.e16):
    """triangulation.py:8-27 (cv2.triangulatePoints)."""
    x, _ = triangulate_views(np.stack([u1, u2]), np.stack([P1[0:3, 0:4], P2[0:3, 0:4]]), "dlt")
    with np.errstate(invalid="ignore"):
        status = np.max(np.abs(x), axis=1) <= max_coordinate_value
    return x, status


def polynomial_triangulation(u1, P1, u2, P2):
    """triangulation.py:184-220: optimal (Hartley & Sturm) correction of the matches, then the linear-eigen solve.  When the
    correction with the projection-derived F yields NaN for every point (:213), F is re-estimated from the matches themselves
    (8-point, :216) as the reference does."""
    x, st = triangulate_views(np.stack([u1, u2]), np.stack([P1[0:3, 0:4], P2[0:3, 0:4]]), "poly")
    if np.isnan(x).all():
        f, ok = find_fundamental_mat_8point(u1, u2)
        if ok:
            c1, c2 = correct_matches(f, np.asarray(u1, np.float64)[None], np.asarray(u2, np.float64)[None])
            return linear_eigen_triangulation(c1[0], P1, c2[0], P2)
    return x, st.astype(bool)


def find_fundamental_mat_8point(u1, u2, device=None):
    """cv2.findFundamentalMat(u1, u2, cv2.FM_8POINT)[0] (triangulation.py:216): u1, u2 [N, 2] -> (F [3, 3] float64, ok)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    a = torch.as_tensor(np.asarray(u1, np.float64).reshape(1, -1, 2), device=device)
    b = torch.as_tensor(np.asarray(u2, np.float64).reshape(1, -1, 2), device=device)
    f, st = hip.fundamental_8point(a, b)
    return f[0].cpu().numpy(), bool(st[0].item())


# ---- cv2.findFundamentalMat(u1, u2, cv2.FM_LMEDS) (cameras.py:136-143) ---------------------------------------------------------------
# OpenCV 4.1.0's least-median-of-squares estimator (calib3d: fundam.cpp, ptsetreg.cpp; core: cv::RNG) restated: 300 seven-point samples drawn
# by ITS fixed-seed generator (`RNG rng((uint64)-1)`: the estimate is a deterministic function of the input, in OpenCV and here), up to three
# matrices per sample from the cubic det(lambda F1 + (1 - lambda) F2) = 0, the candidate with the least MEDIAN symmetric epipolar error wins,
# inliers within 2.5 * 1.4826 * (1 + 5 / (N - 7)) * sqrt(median).  Sampling and the 7-point solves are host work on 7 points each; the part that
# scales with N -- every candidate's error over every pair and its median -- runs on the GPU (csrc/fundamental.hip).  Parity with OpenCV itself is
# unpinned (no cv2 in the image): pinned by the oracle's independent restatement and by properties (tests/test_hip_selfsup.py).
_CV_RNG_COEFF = 4164903690


def _cv_rng_stream(state=0xFFFFFFFFFFFFFFFF):
    """cv::RNG::next() as a generator of 32-bit values (multiply-with-carry)."""
    state = state or 0xFFFFFFFF
    while True:
        state = ((state & 0xFFFFFFFF) * _CV_RNG_COEFF + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        yield state & 0xFFFFFFFF


def _last_point_collinear(pts):
    """fundam.cpp haveCollinearPoints on a [7, 2] float32 sample: the last point against every pair of earlier ones."""
    d = pts[:-1].astype(np.float64) - pts[-1].astype(np.float64)
    cross = np.abs(d[:, None, 0] * d[None, :, 1] - d[:, None, 1] * d[None, :, 0])          # |dx2 dy1 - dy2 dx1| for every (k, j)
    bound = np.finfo(np.float32).eps * (np.abs(d).sum(1)[:, None] + np.abs(d).sum(1)[None, :])
    k, j = np.tril_indices(len(d), -1)                                                     # pairs with k < j (the diagonal is a point with itself)
    return bool((cross[j, k] <= bound[j, k]).any())


def _lmeds_samples(m1, m2, niters, model_points=7, max_attempts=1000):
    """ptsetreg.cpp getSubset, `niters` times on one generator: [S, 7] index rows (S < niters when sampling fails, as the reference breaks)."""
    count, rng, rows = len(m1), _cv_rng_stream(), []
    for _ in range(niters):
        for _attempt in range(max_attempts):
            idx = []
            while len(idx) < model_points:
                v = int(next(rng) % count)
                if v not in idx:
                    idx.append(v)
            if not _last_point_collinear(m1[idx]) and not _last_point_collinear(m2[idx]):
                rows.append(idx)
                break
        else:
            break
    return np.asarray(rows, np.int64).reshape(-1, model_points)


def _seven_point_candidates(s1, s2):
    """fundam.cpp run7Point for a batch of samples s1, s2 [S, 7, 2] -> candidates [H, 3, 3] float64 in sample order (1 .. 3 per sample)."""
    x0, y0, x1, y1 = (s1[..., 0].astype(np.float64), s1[..., 1].astype(np.float64), s2[..., 0].astype(np.float64), s2[..., 1].astype(np.float64))
    a = np.stack([x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, np.ones_like(x0)], axis=-1)        # [S, 7, 9]
    vt = np.linalg.svd(a, full_matrices=True)[2]
    f2 = vt[:, 8]
    f1 = vt[:, 7] - f2
    lam = np.array([-1.0, 0.0, 1.0, 2.0])
    vals = np.linalg.det((lam[None, :, None] * f1[:, None, :] + f2[:, None, :]).reshape(len(a), 4, 3, 3))       # the cubic through four samples
    coeffs = np.linalg.solve(np.vander(lam, 4), vals.T).T                                                      # [S, 4]: c3 .. c0
    eps = np.finfo(np.float64).eps
    out = []
    for c, g1, g2 in zip(coeffs, f1, f2):
        if not np.all(np.isfinite(c)):
            continue
        for r in (np.roots(c) if abs(c[0]) > 0 else np.roots(c[1:])):
            if abs(r.imag) > 1e-9 * max(1.0, abs(r.real)):
                continue
            lam_k, mu = float(r.real), 1.0
            s = g1[8] * lam_k + g2[8]
            f = np.empty(9)
            if abs(s) > eps:
                mu = 1.0 / s
                lam_k *= mu
                f[8] = 1.0
            else:
                f[8] = 0.0
            f[:8] = g1[:8] * lam_k + g2[:8] * mu
            out.append(f.reshape(3, 3))
    return np.asarray(out, np.float64).reshape(-1, 3, 3)


def find_fundamental_mat_lmeds(u1, u2, confidence=0.99, device=None):
    """cv2.findFundamentalMat(u1, u2, cv2.FM_LMEDS) -> (F [3, 3] float64 or None, mask uint8 [N, 1]).  u1, u2 [N, 2] (converted to float32 as
    OpenCV does; the reference passes int32 pixel coordinates)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    m1, m2 = np.asarray(u1, np.float32).reshape(-1, 2), np.asarray(u2, np.float32).reshape(-1, 2)
    count, model_points = len(m1), 7
    none = (None, np.zeros((count, 1), np.uint8))
    if count < model_points or len(m2) != count:
        return none
    if count > model_points:
        num, denom = np.log(max(1.0 - confidence, np.finfo(np.float64).tiny)), np.log(1.0 - (1.0 - 0.45) ** model_points)
        niters = int(np.rint(num / denom)) if -num < 1000 * (-denom) else 1000           # RANSACUpdateNumIters(confidence, 0.45, 7, 1000) = 300
        rows = _lmeds_samples(m1, m2, niters)
        if len(rows) == 0:
            return none
        cand = _seven_point_candidates(m1[rows], m2[rows])
    else:
        cand = _seven_point_candidates(m1[None], m2[None])
    if len(cand) == 0:
        return none
    d1, d2 = torch.as_tensor(m1.astype(np.float64), device=device), torch.as_tensor(m2.astype(np.float64), device=device)
    med = hip.fundamental_lmeds_medians(torch.as_tensor(cand, device=device), d1, d2).cpu().numpy()
    med = np.where(np.isnan(med), np.inf, med)
    best = int(np.argmin(med))                           # the FIRST least median: the reference updates on `median < minMedian`
    if not np.isfinite(med[best]):
        return none
    f = cand[best]
    with np.errstate(divide="ignore", invalid="ignore"):     # (N == 7: 5 / 0 = inf as in C++ -- every point an inlier)
        sigma = 2.5 * 1.4826 * (1.0 + np.float64(5.0) / np.float64(count - model_points)) * np.sqrt(med[best])
    sigma = sigma if sigma > 0.001 else 0.001
    err = hip.fundamental_errors(torch.as_tensor(f, device=device), d1, d2).cpu().numpy()
    mask = (err <= np.float32(sigma * sigma)).astype(np.uint8)
    if int(mask.sum()) < model_points:
        return none
    return f, mask.reshape(-1, 1)


def essential_matrix(F, K1, K2=None):
    """cameras.py:133-134 (Camera.get_essential_matrix): E = K^T F K; K2 generalises to two different cameras (E = K2^T F K1).
    Dataset-time 3x3 host arithmetic."""
    k1 = np.asarray(K1, np.float64)
    k2 = k1 if K2 is None else np.asarray(K2, np.float64)
    return k2.T.dot(np.asarray(F, np.float64)).dot(k1)


def correct_matches(F, points1, points2, device=None):
    """cv2.correctMatches(F, points1, points2) with points [1,N,2] (or [N,2]) -> arrays of the same shape."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    p1, p2 = np.asarray(points1, np.float64), np.asarray(points2, np.float64)
    f = torch.as_tensor(np.asarray(F, np.float64).reshape(1, 3, 3), device=device)
    o1, o2 = hip.correct_matches(f, torch.as_tensor(p1.reshape(1, -1, 2), device=device),
                                 torch.as_tensor(p2.reshape(1, -1, 2), device=device))
    return o1.cpu().numpy().reshape(p1.shape), o2.cpu().numpy().reshape(p2.shape)
