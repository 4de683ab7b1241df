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
