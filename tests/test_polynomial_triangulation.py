"""CPU: polynomial (Hartley & Sturm) triangulation -- the oracle's defining properties, the reference's own glue around
cv2.correctMatches run live at golden-generation time, and the DEVICE math compiled for the host (tests/hostcheck) against
the oracle.  The GPU run of the same kernels is tests/test_hip_selfsup.py."""
import ctypes
import os
import sys

import numpy as np
import pytest

from oracle import triangulation as o_tri

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostcheck"))


def _dp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def hostlib():
    import build as hostcheck_build
    return ctypes.CDLL(hostcheck_build.build())


def _scene(seed, n=64, noise=2.0, pair=(0, 1)):
    from epipolarpose_amd.synthetic import make_cameras, project
    cams = make_cameras(4)
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 300, size=(n, 3)) + [0, 0, 900]
    a, b = pair
    p1 = np.ascontiguousarray(cams[a]["projection_matrix"], dtype=np.float64)
    p2 = np.ascontiguousarray(cams[b]["projection_matrix"], dtype=np.float64)
    u1 = np.ascontiguousarray(project(x, cams[a])[0] + rng.normal(0, noise, (n, 2)))
    u2 = np.ascontiguousarray(project(x, cams[b])[0] + rng.normal(0, noise, (n, 2)))
    return x, u1, u2, p1, p2


def _h(u):
    return np.concatenate([u, np.ones((len(u), 1))], axis=1)


def _epi_residual(f, u1, u2):
    """distance-like residual: |x2^T F x1| / |F|"""
    return np.abs(np.einsum("ni,ij,nj->n", _h(u2), f, _h(u1))) / np.abs(f).max()


@pytest.mark.parametrize("pair", [(0, 1), (0, 3), (1, 2)])
def test_oracle_correction_is_epipolar_and_minimal(pair):
    _, u1, u2, p1, p2 = _scene(3, n=24, noise=3.0, pair=pair)
    f = o_tri.fundamental_from_projections(p1, p2)
    c1, c2 = o_tri.correct_matches(f, u1, u2)
    assert _epi_residual(f, c1, c2).max() < 1e-9 * (1 + np.abs(u1).max()) ** 2
    moved = ((c1 - u1) ** 2).sum(1) + ((c2 - u2) ** 2).sum(1)
    # minimality (HZ 12.5): no exactly-epipolar pair is closer.  Candidates: for y1 near c1, the closest point of the
    # epipolar line F y1 to u2.
    rng = np.random.default_rng(0)
    for _ in range(200):
        y1 = c1 + rng.normal(0, 0.5, c1.shape)
        line = _h(y1) @ f.T                                   # l2 = F y1
        nrm = np.hypot(line[:, 0], line[:, 1])
        dist = (line * _h(u2)).sum(1) / nrm
        y2 = u2 - (dist / nrm)[:, None] * line[:, :2]
        cand = ((y1 - u1) ** 2).sum(1) + ((y2 - u2) ** 2).sum(1)
        assert (cand >= moved - 1e-9).all()
    # the Sampson correction is the first-order approximation of the same quantity
    e = np.einsum("ni,ij,nj->n", _h(u2), f, _h(u1))
    fx1, ftx2 = _h(u1) @ f.T, _h(u2) @ f
    sampson = e ** 2 / (fx1[:, 0] ** 2 + fx1[:, 1] ** 2 + ftx2[:, 0] ** 2 + ftx2[:, 1] ** 2)
    np.testing.assert_allclose(moved, sampson, rtol=5e-2)


def test_oracle_noise_free_matches_stay_put_and_scale_invariance():
    x, u1, u2, p1, p2 = _scene(5, n=16, noise=0.0)
    f = o_tri.fundamental_from_projections(p1, p2)
    assert _epi_residual(f, u1, u2).max() < 1e-9 * (1 + np.abs(u1).max()) ** 2      # F really is the pair's fundamental matrix
    c1, c2 = o_tri.correct_matches(f, u1, u2)
    np.testing.assert_allclose(c1, u1, atol=1e-5)
    np.testing.assert_allclose(c2, u2, atol=1e-5)
    xt, st = o_tri.polynomial_triangulation(np.stack([u1, u2]), np.stack([p1, p2]))
    np.testing.assert_allclose(xt, x, atol=1e-3)
    assert st.all()
    _, n1, n2, _, _ = _scene(6, n=16, noise=2.0)
    a1, a2 = o_tri.correct_matches(f, n1, n2)
    b1, b2 = o_tri.correct_matches(-37.5 * f, n1, n2)
    np.testing.assert_allclose(a1, b1, atol=1e-8)
    np.testing.assert_allclose(a2, b2, atol=1e-8)


def test_reference_polynomial_triangulation_golden(golden):
    """The reference's polynomial_triangulation run live (tests/golden/make_golden.py) -- its own F construction and
    hand-over to linear_eigen_triangulation -- with cv2.invert / cv2.correctMatches / cv2.triangulatePoints supplied by
    the shims (OpenCV itself is not in this image: "parity unpinned" at the OpenCV layer, see oracle/__init__.py)."""
    g = golden("triangulation")
    x, st = o_tri.polynomial_triangulation(np.stack([g["poly/u1"], g["poly/u2"]]), np.stack([g["poly/P1"], g["poly/P2"]]))
    np.testing.assert_allclose(x, g["poly/X"], rtol=1e-9, atol=1e-7)
    np.testing.assert_array_equal(st, g["poly/status"].astype(bool))
    np.testing.assert_allclose(o_tri.fundamental_from_projections(g["poly/P1"], g["poly/P2"]), g["poly/F"], rtol=1e-12, atol=1e-9)


# ------------------------------------------------------------------ device math compiled for the host
def test_hostcheck_real_root_isolation(hostlib):
    rng = np.random.default_rng(1)
    for trial in range(200):
        kind = trial % 4
        if kind == 0:
            r = rng.uniform(-3, 3, 6)
        elif kind == 1:                                         # two complex pairs
            r = np.concatenate([rng.uniform(-2, 2, 2), [0.3 + 0.4j, 0.3 - 0.4j, -1.5 + 0.1j, -1.5 - 0.1j]])
        elif kind == 2:                                         # widely spread magnitudes
            r = rng.uniform(-1, 1, 6) * 10.0 ** rng.uniform(-4, 4, 6)
        else:                                                   # degree drops (leading coefficients vanish)
            r = rng.uniform(-0.9, 0.9, 4)
        c = np.real(np.poly(r))[::-1].copy()
        c = np.concatenate([c, np.zeros(7 - len(c))]) * 10.0 ** rng.uniform(-3, 3)
        out = np.zeros(6)
        n = hostlib.hostcheck_real_roots6(_dp(c), _dp(out))
        want = np.sort(np.real(r[(np.abs(np.imag(r)) < 1e-12) & (np.abs(np.real(r)) <= 1)]))
        assert n == len(want), (trial, out[:n], want)
        np.testing.assert_allclose(np.sort(out[:n]), want, rtol=1e-9, atol=1e-13)


def test_hostcheck_svd3(hostlib):
    rng = np.random.default_rng(2)
    for trial in range(50):
        a = rng.normal(size=(3, 3))
        if trial % 3 == 0:
            a[2] = 0.3 * a[0] - 2.0 * a[1]                      # rank 2, as a fundamental matrix
        a = np.ascontiguousarray(a)
        u, s, v = np.zeros(9), np.zeros(3), np.zeros(9)
        hostlib.hostcheck_svd3(_dp(a), _dp(u), _dp(s), _dp(v))
        u, v = u.reshape(3, 3), v.reshape(3, 3)
        np.testing.assert_allclose(s, np.linalg.svd(a, compute_uv=False), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(v.T @ v, np.eye(3), atol=1e-13)
        np.testing.assert_allclose(a @ v, u * s, atol=1e-13)


@pytest.mark.parametrize("pair", [(0, 1), (0, 3), (1, 2), (2, 3)])
@pytest.mark.parametrize("noise", [0.0, 0.5, 5.0, 50.0])
def test_hostcheck_polynomial_triangulation_matches_oracle(hostlib, pair, noise):
    x, u1, u2, p1, p2 = _scene(7, n=128, noise=noise, pair=pair)
    f = np.ascontiguousarray(o_tri.fundamental_from_projections(p1, p2))
    fd = np.zeros(9)
    hostlib.hostcheck_fundamental(_dp(p1), _dp(p2), _dp(fd))
    np.testing.assert_allclose(fd.reshape(3, 3), f, rtol=0, atol=1e-13 * np.abs(f).max())
    c1, c2 = o_tri.correct_matches(f, u1, u2)
    d1, d2 = np.zeros_like(u1), np.zeros_like(u2)
    hostlib.hostcheck_correct_matches(_dp(f), _dp(u1), _dp(u2), len(u1), _dp(d1), _dp(d2))
    tol = 1e-5 if noise == 0.0 else 1e-9                        # noise-free: t = 0 is a multiple root, ill-conditioned for both
    np.testing.assert_allclose(d1, c1, atol=tol)
    np.testing.assert_allclose(d2, c2, atol=tol)
    assert _epi_residual(f, d1, d2).max() < 1e-9 * (1 + np.abs(u1).max()) ** 2
    xo, so = o_tri.polynomial_triangulation(np.stack([u1, u2]), np.stack([p1, p2]))
    xd, sd = np.zeros_like(xo), np.zeros(len(u1), np.int32)
    hostlib.hostcheck_poly_triangulate(_dp(u1), _dp(u2), _dp(p1), _dp(p2), len(u1), _dp(xd), _dp(sd))
    np.testing.assert_allclose(xd, xo, atol=1e-4 if noise == 0.0 else 1e-6)
    np.testing.assert_array_equal(sd.astype(bool), so)


# ------------------------------------------------------------------ fundamental matrix from matches (8-point)
@pytest.mark.parametrize("pair", [(0, 1), (0, 3), (1, 2)])
def test_oracle_8point_properties(pair):
    _, u1, u2, p1, p2 = _scene(11, n=17, noise=0.0, pair=pair)
    f, ok = o_tri.fundamental_8point(u1, u2)
    ft = o_tri.fundamental_from_projections(p1, p2)
    assert ok and abs(f[2, 2] - 1.0) < 1e-12
    np.testing.assert_allclose(f, ft / ft[2, 2], rtol=0, atol=1e-9 * np.abs(ft / ft[2, 2]).max())     # exact on noise-free matches
    assert np.linalg.svd(f, compute_uv=False)[2] < 1e-12 * np.linalg.svd(f, compute_uv=False)[0]       # rank 2
    _, n1, n2, _, _ = _scene(12, n=17, noise=1.0, pair=pair)
    fa, _ = o_tri.fundamental_8point(n1, n2)
    fb, _ = o_tri.fundamental_8point(n2, n1)                                                              # swapped views: F^T
    np.testing.assert_allclose(fb / np.linalg.norm(fb), fa.T / np.linalg.norm(fa), atol=1e-9)
    fs, _ = o_tri.fundamental_8point(n1 + [100.0, -50.0], n2)                                             # translation of view 1
    t = np.array([[1, 0, -100.0], [0, 1, 50.0], [0, 0, 1.0]])
    np.testing.assert_allclose(fs / np.linalg.norm(fs), (fa @ t) / np.linalg.norm(fa @ t), atol=1e-8)
    assert not o_tri.fundamental_8point(n1[:7], n2[:7])[1]
    assert not o_tri.fundamental_8point(np.zeros((9, 2)), n2[:9])[1]
    k = np.array([[1100.0, 0, 500.0], [0, 1100.0, 510.0], [0, 0, 1.0]])
    e = o_tri.essential_from_fundamental(ft, k)
    sv = np.linalg.svd(e, compute_uv=False)
    assert sv[2] < 1e-9 * sv[0]                                                                           # an essential matrix is rank 2


@pytest.mark.parametrize("n,noise", [(8, 0.0), (17, 0.0), (17, 1.0), (32, 3.0), (9, 0.5)])
def test_hostcheck_8point_matches_oracle(hostlib, n, noise):
    _, u1, u2, _, _ = _scene(13, n=n, noise=noise, pair=(1, 3))
    f, ok = o_tri.fundamental_8point(u1, u2)
    fd = np.zeros(9)
    okd = hostlib.hostcheck_fundamental_8point(_dp(u1), _dp(u2), n, _dp(fd))
    assert ok and okd == 1
    np.testing.assert_allclose(fd.reshape(3, 3), f, rtol=0, atol=1e-11 * np.abs(f).max())
    assert hostlib.hostcheck_fundamental_8point(_dp(u1), _dp(u2), 7, _dp(fd)) == 0 and not fd.any()


# ------------------------------------------------------------------ fundamental matrix from matches (least median of squares)
def _sym_epi_dist(f, u1, u2):
    """the larger of the two point-to-epipolar-line distances, in pixels"""
    l2, l1 = _h(u1) @ f.T, _h(u2) @ f
    d2 = np.abs((l2 * _h(u2)).sum(1)) / np.hypot(l2[:, 0], l2[:, 1])
    d1 = np.abs((l1 * _h(u1)).sum(1)) / np.hypot(l1[:, 0], l1[:, 1])
    return np.maximum(d1, d2)


def test_oracle_cv_rng_and_iteration_count():
    """cv::RNG is a multiply-with-carry generator (x = x_lo * 4164903690 + x_hi) and OpenCV seeds the LMedS sampler with (uint64)-1;
    the iteration count for confidence 0.99, 45 % outliers, 7 points is round(log(0.01) / log(1 - 0.55^7)) = 300."""
    r = o_tri.CvRNG()
    s = 0xFFFFFFFFFFFFFFFF
    for _ in range(5):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert r.next() == s & 0xFFFFFFFF
    assert o_tri.CvRNG(0).state == 0xFFFFFFFF                            # a zero seed is replaced
    assert o_tri.ransac_update_num_iters(0.99, 0.45, 7, 1000) == 300
    assert o_tri.ransac_update_num_iters(0.99, 0.45, 7, 100) == 100      # capped
    assert o_tri.ransac_update_num_iters(0.99, 0.0, 7, 1000) == 0        # no outliers: 1 - 1^7 = 0, "no iterations needed"
    assert o_tri.ransac_update_num_iters(0.99, 1.0, 7, 1000) == 1000     # only outliers: log(1) = 0 -> the cap
    draws = [o_tri.CvRNG().uniform_int(0, 60) for _ in range(2)]
    assert draws[0] == draws[1] and 0 <= draws[0] < 60                   # fixed seed: the same stream every time


@pytest.mark.parametrize("pair,outliers", [((0, 1), 0), ((0, 2), 20), ((1, 3), 26)])
def test_oracle_lmeds_properties(pair, outliers):
    """findFundamentalMat(FM_LMEDS) restated: up to 43 % gross outliers are rejected, the estimate explains the inliers to the noise level,
    has rank 2 and F[2][2] = 1, is a deterministic function of its input, and its 7-point candidates interpolate their samples."""
    n = 60
    _, u1, u2, p1, p2 = _scene(21 + outliers, n=n, noise=0.5, pair=pair)
    rng = np.random.default_rng(3)
    bad = rng.choice(n, outliers, replace=False)
    u2 = u2.copy()
    u2[bad] += rng.uniform(40, 120, (outliers, 2)) * rng.choice([-1, 1], (outliers, 2))
    i1, i2 = np.int32(u1), np.int32(u2)                    # the reference casts to integer pixels (cameras.py:137-138)
    f, mask = o_tri.fundamental_lmeds(i1, i2)
    assert f is not None and mask.shape == (n,) and abs(f[2, 2] - 1.0) < 1e-12
    sv = np.linalg.svd(f, compute_uv=False)
    assert sv[2] < 1e-9 * sv[0]                            # a member of the pencil with det = 0
    good = np.setdiff1d(np.arange(n), bad)
    assert mask[good].mean() >= 0.8                        # (the 2.5-sigma rule on a median-based sigma trims the tail of the genuine pairs too: 52 of 60 kept)
    assert mask[bad].sum() <= max(1, outliers // 8)        # a gross outlier survives only by landing near its epipolar line
    assert np.median(_sym_epi_dist(f, i1[good].astype(float), i2[good].astype(float))) < 1.5
    ft = o_tri.fundamental_from_projections(p1, p2)
    assert np.median(_sym_epi_dist(ft, i1[good].astype(float), i2[good].astype(float))) < 1.5     # (the yardstick: the true matrix on the same integer pixels)
    f2, mask2 = o_tri.fundamental_lmeds(i1, i2)
    assert np.array_equal(f, f2) and np.array_equal(mask, mask2)
    # the 7-point candidates of a sample pass through its seven pairs
    idx = o_tri.fm_get_subset(o_tri.CvRNG(), i1.astype(np.float32), i2.astype(np.float32))
    assert len(set(idx)) == 7
    cands = o_tri.fm_run_7point(i1[idx].astype(np.float32), i2[idx].astype(np.float32))
    assert 1 <= len(cands) <= 3
    for c in cands:
        assert _epi_residual(c, i1[idx].astype(float), i2[idx].astype(float)).max() < 1e-6
        assert abs(np.linalg.det(c)) < 1e-9 * np.abs(c).max() ** 3 + 1e-18
    # too few points / the minimal case
    assert o_tri.fundamental_lmeds(i1[:6], i2[:6])[0] is None
    f7, m7 = o_tri.fundamental_lmeds(i1[good[:7]], i2[good[:7]])
    assert f7 is not None and m7.sum() == 7


def test_oracle_lmeds_median_and_error():
    """the error is the larger squared epipolar distance as float32; the median of an even count is the mean of the two middle elements"""
    assert o_tri.fm_median(np.float32([4, 1, 3])) == 3.0 and o_tri.fm_median(np.float32([4, 1, 3, 2])) == 2.5
    _, u1, u2, p1, p2 = _scene(5, n=9, noise=0.0)
    f = o_tri.fundamental_from_projections(p1, p2)
    e = o_tri.fm_compute_error(f / f[2, 2], u1.astype(np.float32), u2.astype(np.float32))
    assert e.dtype == np.float32 and e.max() < 1e-3
    off = u2 + [0.0, 7.0]
    e2 = o_tri.fm_compute_error(f / f[2, 2], u1.astype(np.float32), off.astype(np.float32))
    np.testing.assert_allclose(np.sqrt(e2), _sym_epi_dist(f, u1.astype(np.float32).astype(float), off.astype(np.float32).astype(float)), rtol=1e-5)


def test_hostcheck_lmeds_error_matches_oracle(hostlib):
    """csrc/fundamental.hip fm_error compiled for the host: the float32 symmetric epipolar error the scoring kernels compute per pair == the oracle's
    (the same float64 arithmetic, rounded to float32 once), and the median taken from it picks the oracle's candidate."""
    _, u1, u2, p1, p2 = _scene(31, n=41, noise=1.0)
    i1, i2 = np.float32(np.int32(u1)), np.float32(np.int32(u2))
    a, b = np.ascontiguousarray(i1, np.float64), np.ascontiguousarray(i2, np.float64)
    rng = np.random.default_rng(32)
    ft = o_tri.fundamental_from_projections(p1, p2)
    cands = [ft / ft[2, 2], ft / ft[2, 2] + 1e-7 * rng.normal(size=(3, 3)), rng.normal(size=(3, 3))]
    meds = []
    for f in cands:
        f = np.ascontiguousarray(f, np.float64)
        err = np.zeros(len(a), np.float32)
        hostlib.hostcheck_fm_errors(_dp(f), _dp(a), _dp(b), len(a), err.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
        ref = o_tri.fm_compute_error(f, i1, i2)
        np.testing.assert_allclose(err, ref, rtol=2e-6, atol=1e-12)
        meds.append(o_tri.fm_median(err))
    assert int(np.argmin(meds)) == int(np.argmin([o_tri.fm_median(o_tri.fm_compute_error(f, i1, i2)) for f in cands]))
