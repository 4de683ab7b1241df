"""Import the REAL reference (``/root/reference``) in this container despite its missing dependencies.

Used ONLY by ``tests/golden/make_golden.py`` (golden-vector generation) and by CPU tests that are
skipped when ``/root/reference`` is absent (it does not exist on the GPU box).

The reference's own Python executes unmodified.  What is substituted is third-party code that is not
installed here and cannot be (no network):

* ``cv2``      -> a stub with float64 NumPy equivalents of the primitives the hot path calls:
  ``solve(A, b, dst, DECOMP_SVD)``, ``getAffineTransform``, ``triangulatePoints``, ``invert`` and
  ``correctMatches`` (the last by an independent algorithm, see ``_cv2_correct_matches``)
  (OpenCV 4.1.0 documented algorithms; see oracle/triangulation.py, oracle/geometry.py headers).
* ``easydict`` -> a 10-line attribute dict.        * ``h5py`` -> empty stub (never called here).
* ``np.int`` / ``np.float`` -> the builtins they aliased before NumPy 1.24 (prep_h36m.py:179-180,202).
* ``torch.cuda.comm.broadcast`` -> identity on CPU (integral_loss.py:61-63 uses a CUDA-only API to
  copy an ``arange``; BASELINE.md section 4 prescribes this shim).
"""
import importlib
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lib"))


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def _cv2_solve(a, b, dst=None, flags=0):
    x = np.linalg.lstsq(np.asarray(a, np.float64), np.asarray(b, np.float64), rcond=None)[0]
    if dst is not None:
        dst[...] = x          # OpenCV writes into a passed (row-strided) destination in place
        return True, dst
    return True, x


def _cv2_get_affine_transform(src, dst):
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    a = np.concatenate([src, np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, dst).T


def _cv2_triangulate_points(p1, p2, pts1, pts2):
    pts1 = np.asarray(pts1, np.float64)
    pts2 = np.asarray(pts2, np.float64)
    n = pts1.shape[1]
    out = np.zeros((4, n))
    for i in range(n):
        m = np.stack([pts1[0, i] * p1[2] - p1[0], pts1[1, i] * p1[2] - p1[1],
                      pts2[0, i] * p2[2] - p2[0], pts2[1, i] * p2[2] - p2[1]])
        out[:, i] = np.linalg.svd(m)[2][-1]
    return out


def _cv2_invert(a, flags=0):
    return 1.0, np.linalg.inv(np.asarray(a, np.float64))


def _cv2_correct_matches(f, points1, points2):
    """Stand-in for cv2.correctMatches.  DELIBERATELY a different algorithm from the oracle's (and OpenCV's)
    Hartley-Sturm polynomial: Kanatani/Sugaya/Niitsuma's iterated first-order optimal correction, which converges to the
    same minimiser of |x1-x1'|^2 + |x2-x2'|^2 subject to x2'^T F x1' = 0.  The golden vectors therefore cross-check the
    oracle's polynomial solution against an independent solver, through the reference's own glue code."""
    f = np.asarray(f, np.float64)
    p1 = np.asarray(points1, np.float64).reshape(-1, 2)
    p2 = np.asarray(points2, np.float64).reshape(-1, 2)
    o1, o2 = np.empty_like(p1), np.empty_like(p2)
    for i in range(len(p1)):
        x, xp = np.array([p1[i, 0], p1[i, 1], 1.0]), np.array([p2[i, 0], p2[i, 1], 1.0])
        xh, xph, xt, xpt = x.copy(), xp.copy(), np.zeros(3), np.zeros(3)
        for _ in range(100):
            a, b = f.T @ xph, f @ xh                      # gradients of the constraint w.r.t. x and x'
            a[2] = b[2] = 0.0
            lam = (xph @ f @ xh + xph @ f @ xt + xpt @ f @ xh) / (a @ a + b @ b)
            xt_new, xpt_new = lam * a, lam * b
            done = max(np.abs(xt_new - xt).max(), np.abs(xpt_new - xpt).max()) < 1e-14 * (1 + np.abs(x).max())
            xt, xpt = xt_new, xpt_new
            xh, xph = x - xt, xp - xpt
            if done:
                break
        o1[i], o2[i] = xh[:2], xph[:2]
    shape = np.asarray(points1).shape
    return o1.reshape(shape), o2.reshape(shape)


def _install_stubs():
    if "cv2" not in sys.modules:
        cv2 = types.ModuleType("cv2")
        cv2.DECOMP_SVD = 1
        cv2.solve = _cv2_solve
        cv2.getAffineTransform = _cv2_get_affine_transform
        cv2.triangulatePoints = _cv2_triangulate_points
        cv2.invert = _cv2_invert
        cv2.correctMatches = _cv2_correct_matches
        cv2.__stub__ = True
        sys.modules["cv2"] = cv2
    if "easydict" not in sys.modules:
        ed = types.ModuleType("easydict")
        ed.EasyDict = _EasyDict
        sys.modules["easydict"] = ed
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    import torch
    import torch.cuda.comm
    torch.cuda.comm.broadcast = lambda t, devices=None, out=None: [t]


def load_reference():
    """Returns a namespace of live reference modules (``lib.*``)."""
    if not reference_available():
        raise RuntimeError("/root/reference is not available")
    sys.dont_write_bytecode = True
    _install_stubs()
    import matplotlib
    matplotlib.use("Agg")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    ns = types.SimpleNamespace()
    ns.integral_loss = importlib.import_module("lib.core.integral_loss")
    ns.pose3d_resnet = importlib.import_module("lib.models.pose3d_resnet")
    ns.triangulation = importlib.import_module("lib.utils.triangulation")
    ns.prep_h36m = importlib.import_module("lib.utils.prep_h36m")
    ns.cameras = importlib.import_module("lib.utils.cameras")
    ns.img_utils = importlib.import_module("lib.utils.img_utils")
    ns.inference = importlib.import_module("lib.core.inference")
    ns.config = importlib.import_module("lib.core.config")
    ns.EasyDict = _EasyDict
    return ns
