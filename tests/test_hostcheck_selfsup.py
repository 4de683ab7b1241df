"""CPU: the float64 per-thread DEVICE math of csrc/selfsup.hip (triangulators, patch affines, re-projection), compiled for
the host by tests/hostcheck, against the golden vectors of the live reference -- the same numbers the GPU parity tests
(tests/test_hip_selfsup.py) check, available without a GPU."""
import ctypes
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostcheck"))


def _dp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def hostlib():
    import build as hostcheck_build
    lib = ctypes.CDLL(hostcheck_build.build())
    d = ctypes.c_double
    lib.hostcheck_triangulate2.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int, d, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.hostcheck_patch_affines.argtypes = [d] * 8 + [ctypes.c_void_p] * 2
    lib.hostcheck_reproject.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4 + [d] * 9 + [ctypes.c_void_p]
    return lib


def _tri(lib, method, u1, u2, p1, p2, tol=3.0e-5, max_iter=10):
    u1, u2 = np.ascontiguousarray(u1, np.float64), np.ascontiguousarray(u2, np.float64)
    p1, p2 = np.ascontiguousarray(p1, np.float64), np.ascontiguousarray(p2, np.float64)
    x, st = np.zeros((len(u1), 3)), np.zeros(len(u1), np.int32)
    lib.hostcheck_triangulate2(method, _dp(u1), _dp(u2), _dp(p1), _dp(p2), len(u1), tol, max_iter, _dp(x), _dp(st))
    return x, st


@pytest.mark.parametrize("noise", [0, 2])
def test_device_triangulators_match_reference_golden(golden, hostlib, noise):
    g = golden("triangulation")
    u, ps = g["u/noise%d" % noise], g["P"]
    for va, vb in ((0, 1), (0, 3), (1, 2)):
        for grp in range(3):
            tag = "noise%d/v%d%d/g%d" % (noise, va, vb, grp)
            x, st = _tri(hostlib, 0, u[va, grp], u[vb, grp], ps[va], ps[vb])
            np.testing.assert_allclose(x, g[tag + "/iter_x"], atol=1e-6)
            np.testing.assert_array_equal(st, g[tag + "/iter_status"])
            x, _ = _tri(hostlib, 1, u[va, grp], u[vb, grp], ps[va], ps[vb])
            np.testing.assert_allclose(x, g[tag + "/ls_x"], atol=1e-6)
            x, st = _tri(hostlib, 2, u[va, grp], u[vb, grp], ps[va], ps[vb])
            np.testing.assert_allclose(x, g[tag + "/eigen_x"], atol=1e-6)
            np.testing.assert_array_equal(st.astype(bool), g[tag + "/eigen_status"])


@pytest.mark.parametrize("noise", [0, 2])
def test_bulk_variants_match_reference_golden(golden, hostlib, noise):
    """The fp32-storage (bulk) variants of csrc/selfsup.hip compiled for the host and fed the golden float64 inputs: the mixed-precision
    iterative solver (float64 first round, float32 corrections with the stopping rule on the depth difference) must land within the fp32
    path's 1e-2 mm of the live reference AND reproduce its status codes; the normal-equation and Gram variants are float64 throughout."""
    g = golden("triangulation")
    u, ps = g["u/noise%d" % noise], g["P"]
    worst, worst32 = 0.0, 0.0
    for va, vb in ((0, 1), (0, 3), (1, 2)):
        for grp in range(3):
            tag = "noise%d/v%d%d/g%d" % (noise, va, vb, grp)
            x, st = _tri(hostlib, 3, u[va, grp], u[vb, grp], ps[va], ps[vb])
            worst = max(worst, float(np.abs(x - g[tag + "/iter_x"]).max()))
            np.testing.assert_allclose(x, g[tag + "/iter_x"], atol=5e-3)
            np.testing.assert_array_equal(st, g[tag + "/iter_status"])
            x, st = _tri(hostlib, 5, u[va, grp], u[vb, grp], ps[va], ps[vb])
            np.testing.assert_allclose(x, g[tag + "/iter_x"], atol=1e-5)
            np.testing.assert_array_equal(st, g[tag + "/iter_status"])
            x, st = _tri(hostlib, 4, u[va, grp], u[vb, grp], ps[va], ps[vb])
            np.testing.assert_allclose(x, g[tag + "/eigen_x"], atol=1e-4)
            x, st = _tri(hostlib, 6, u[va, grp], u[vb, grp], ps[va], ps[vb])        # round 5: LS through float64 normal equations (the fp32-storage path)
            np.testing.assert_allclose(x, g[tag + "/ls_x"], atol=1e-4)
            assert (st == 1).all()
            # round 6: the float32-storage instantiations (inputs rounded to float32, rows built in float32 with one rounding, sums of products in
            # float64): inside the fp32 path's 1e-2 mm of the live reference, and within 1e-3 mm of the all-float64 arithmetic on the SAME rounded inputs
            u32 = lambda a: np.asarray(a, np.float32).astype(np.float64)        # noqa: E731
            for m_new, m_wide, key in ((7, 6, "/ls_x"), (8, 4, "/eigen_x")):
                xn, stn = _tri(hostlib, m_new, u[va, grp], u[vb, grp], ps[va], ps[vb])
                xw, _ = _tri(hostlib, m_wide, u32(u[va, grp]), u32(u[vb, grp]), u32(ps[va]), u32(ps[vb]))
                worst32 = max(worst32, float(np.abs(xn - xw).max()))
                np.testing.assert_allclose(xn, g[tag + key], atol=1e-2)
                np.testing.assert_allclose(xn, xw, atol=1e-3)
                assert (stn == 1).all()
    print("mixed-precision iterative solver: worst |x - reference| = %.2e mm (noise %d px); float32-row LS / DLT against float64 rows on the same "
          "inputs: %.2e mm" % (worst, noise, worst32))


def test_device_status_codes_behind_cameras(golden, hostlib):
    g = golden("triangulation")
    x, st = _tri(hostlib, 0, g["behind/u0"], g["behind/u1"], g["P"][0], g["P"][1])
    np.testing.assert_array_equal(st, g["behind/status"])
    np.testing.assert_allclose(x, g["behind/x"], rtol=1e-7, atol=1e-5)


def test_device_patch_affines_match_reference_golden(golden, hostlib):
    g = golden("geometry")
    for p, fwd, inv in zip(g["affine/params"], g["affine/fwd"], g["affine/inv"]):
        a, b = np.zeros(6), np.zeros(6)
        hostlib.hostcheck_patch_affines(*[float(v) for v in p[:6]], 256.0, 256.0, _dp(a), _dp(b))
        np.testing.assert_allclose(a.reshape(2, 3), inv, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(b.reshape(2, 3), fwd, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tag,n_group,j,seed", [("h36m", 3, 17, 48), ("mpii", 2, 16, 47)])
def test_device_reprojection_matches_reference_golden(golden, hostlib, tag, n_group, j, seed):
    from epipolarpose_amd.synthetic import SyntheticScenes
    g = golden("geometry")
    sc = SyntheticScenes(n_group=n_group, n_view=2, num_joints=j, seed=seed)
    want = g[tag + "/labels_from_world/label"]
    m = sc.meta
    for n in range(2 * n_group):
        x = np.ascontiguousarray(sc.world[n % n_group], np.float64)
        lab = np.zeros(3 * j, np.float32)
        r, t = np.ascontiguousarray(m["R"][n], np.float64), np.ascontiguousarray(m["T"][n], np.float64).reshape(3)
        f, c = np.ascontiguousarray(m["f"][n], np.float64).reshape(2), np.ascontiguousarray(m["c"][n], np.float64).reshape(2)
        hostlib.hostcheck_reproject(_dp(x), j, 0, _dp(r), _dp(t), _dp(f), _dp(c), float(m["center_x"][n]), float(m["center_y"][n]),
                                    float(m["width"][n]), float(m["height"][n]), float(m["scale"][n]), float(m["rot"][n]), 256.0, 256.0,
                                    2000.0, _dp(lab))
        np.testing.assert_allclose(lab, want[n], atol=1e-7)
