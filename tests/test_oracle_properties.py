"""CPU property tests (hypothesis) of the oracle -- the size-independent invariants the GPU suite re-checks at full size."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from epipolarpose_amd.synthetic import make_cameras, patch_affine, project
from oracle import geometry, integral, triangulation

CAMS = make_cameras(4)
PS = np.stack([c["projection_matrix"] for c in CAMS])


def observe(seed, n_pt=5, noise=1.0):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 300, size=(n_pt, 3)) + [0, 0, 900]
    u = np.stack([project(x, c)[0] for c in CAMS]) + rng.normal(0, noise, size=(4, n_pt, 2))
    return x, u


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10_000), st.permutations([0, 1, 2, 3]))
def test_view_permutation_invariance(seed, perm):
    _, u = observe(seed)
    perm = list(perm)
    for fn in (triangulation.dlt_triangulation, triangulation.linear_ls_triangulation):
        a, _ = fn(u, PS)
        b, _ = fn(u[perm], PS[perm])
        np.testing.assert_allclose(a, b, atol=1e-6)


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10_000), st.floats(0.1, 50.0))
def test_dlt_projective_scale_invariance(seed, k):
    _, u = observe(seed)
    a, _ = triangulation.dlt_triangulation(u, PS)
    b, _ = triangulation.dlt_triangulation(u, PS * k)
    np.testing.assert_allclose(a, b, atol=1e-6)


@settings(max_examples=10, deadline=None)
@given(st.integers(0, 10_000))
def test_noise_free_recovery_all_solvers(seed):
    x, u = observe(seed, noise=0.0)
    for fn in (triangulation.dlt_triangulation, triangulation.linear_ls_triangulation, triangulation.iterative_ls_triangulation):
        for views in ([0, 1], [0, 2, 3], [0, 1, 2, 3]):
            got, status = fn(u[views], PS[views])
            np.testing.assert_allclose(got, x, atol=1e-6)
            assert np.all(np.asarray(status) == 1)


@settings(max_examples=15, deadline=None)
@given(st.integers(0, 10_000), st.floats(-30.0, 30.0))
def test_softargmax_shift_invariance_and_range(seed, shift):
    rng = np.random.default_rng(seed)
    j, d, h, w = 2, 4, 6, 8
    logits = rng.normal(0, 3, size=(2, j * d, h, w))
    a = integral.softmax_integral(logits, j, w, h, d)
    b = integral.softmax_integral(logits + shift, j, w, h, d)
    np.testing.assert_allclose(a, b, atol=1e-12)
    xyz = a.reshape(2, j, 3)
    assert np.all(xyz >= -0.5) and np.all(xyz[:, :, 0] <= 0.5 - 1.0 / w) and np.all(xyz[:, :, 2] <= 0.5 - 1.0 / d)
    # gradient rows sum to zero (softmax Jacobian annihilates constants)
    g = integral.softmax_integral_backward(logits, j, w, h, d, rng.normal(size=(2, 3 * j)))
    np.testing.assert_allclose(g.reshape(2 * j, -1).sum(axis=1), 0, atol=1e-12)


@settings(max_examples=25, deadline=None)
@given(st.floats(200, 800), st.floats(200, 800), st.floats(150, 600), st.floats(150, 600), st.floats(0.75, 1.25), st.floats(-60, 60))
def test_crop_affine_round_trip_and_closed_form(cx, cy, bw, bh, scale, rot):
    fwd = geometry.gen_trans_from_patch(cx, cy, bw, bh, 256, 256, scale, rot, inv=False)
    inv = geometry.gen_trans_from_patch(cx, cy, bw, bh, 256, 256, scale, rot, inv=True)
    pts = np.array([[0.0, 0.0], [255.0, 13.0], [128.0, 128.0], [31.5, 200.25]])
    back = geometry.trans_points2d(geometry.trans_points2d(pts, inv), fwd)
    np.testing.assert_allclose(back, pts, atol=1e-7)
    # the product's closed-form affine (synthetic.patch_affine, mirrored by the HIP kernel) equals the 3-point solve
    np.testing.assert_allclose(patch_affine(cx, cy, bw, bh, 256, 256, scale, rot), fwd, rtol=1e-9, atol=1e-7)
    np.testing.assert_allclose(patch_affine(cx, cy, bw, bh, 256, 256, scale, rot, inverse=True), inv, rtol=1e-9, atol=1e-7)


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10_000))
def test_label_codec_round_trip(seed):
    joints = np.random.default_rng(seed).uniform(-200, 400, size=(17, 3))
    lab, _ = integral.generate_joint_location_label(256.0, 256.0, joints, np.ones((17, 3)))
    np.testing.assert_allclose(integral.reverse_joint_location_label(256.0, 256.0, lab), joints, atol=1e-10)
