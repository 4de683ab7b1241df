"""CPU: the oracle (float64 restatement) against golden vectors produced by the reference itself."""
import zlib

import numpy as np
import pytest

from det_weights import seeded_array
from make_golden_cases import INTEGRAL_CASES, dlogits_stride
from oracle import geometry, inference, integral, triangulation


def case_logits(g, name, b, j, d, h, w, sc):
    if name + "/logits" in g:
        return g[name + "/logits"]
    logits = seeded_array("logits/" + name, (b, j * d, h, w), scale=sc)
    assert np.uint32(zlib.crc32(logits.tobytes())) == g[name + "/logits_crc"], "seeded input drifted"
    return logits


@pytest.mark.parametrize("case", INTEGRAL_CASES, ids=[c[0] for c in INTEGRAL_CASES])
def test_softmax_integral_and_losses(golden, case):
    g = golden("integral")
    name, b, j, d, h, w, sc = case
    logits = case_logits(g, *case)
    xyz = integral.softmax_integral(logits, j, w, h, d)
    # The pin proper: the reference's code run on float64 tensors (make_golden.py) -- the oracle restates that arithmetic to rounding.  The reference's
    # float32 run is itself some way from it (2e-6 for the small cases, 1.8e-5 in the coordinates and 2.9e-4 of the largest gradient element at the
    # configuration's shape "cfg": fp32 sums over 262 144 voxels): the oracle is held to the float32 run by that distance plus a margin.
    np.testing.assert_allclose(xyz, g[name + "/xyz64"], atol=1e-12)
    np.testing.assert_allclose(xyz, g[name + "/xyz"], atol=1.5 * np.abs(g[name + "/xyz"] - g[name + "/xyz64"]).max() + 2e-6)       # reference is fp32
    for kind in integral.LOSS_KINDS:
        for norm in (False, True):
            key = "%s/%s/norm%d" % (name, kind, int(norm))
            loss, _ = integral.joint_location_loss(logits, g[name + "/gt"], g[name + "/wt"], j, kind, norm)
            dl = integral.joint_location_loss_backward(logits, g[name + "/gt"], g[name + "/wt"], j, kind, norm)
            ref = g[key + "/dlogits"]
            if ref.ndim == 1:
                dl = dl.reshape(-1)[::dlogits_stride(dl.size)]
            scale = np.abs(ref).max()
            own = 0.0                                       # the float32 reference's distance from its float64 run
            if key + "/loss64" in g:
                np.testing.assert_allclose(loss, g[key + "/loss64"], rtol=1e-11, atol=1e-14)
                np.testing.assert_allclose(dl, g[key + "/dlogits64"], atol=1e-11 * scale + 1e-18)
                own = np.abs(ref - g[key + "/dlogits64"]).max()
            np.testing.assert_allclose(loss, g[key + "/loss"], rtol=2e-5, atol=1e-7)
            np.testing.assert_allclose(dl, ref, atol=1.5 * own + 2e-5 * scale + 1e-8)   # (fp32 autograd noise floor)
    if name + "/decode256" in g:
        dec = integral.get_joint_location_result(256, 256, logits)
        own_xyz = float(np.abs(g[name + "/xyz"] - g[name + "/xyz64"]).max())
        np.testing.assert_allclose(dec, g[name + "/decode256"], atol=256 * (1.5 * own_xyz + 4e-6))   # fp32 coords * 256


def test_label_codec(golden):
    g = golden("integral")
    lab, _ = integral.generate_joint_location_label(256.0, 256.0, g["label/joints"], np.ones((17, 3)))
    np.testing.assert_allclose(lab, g["label/label"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(integral.reverse_joint_location_label(256.0, 256.0, lab), g["label/reverse"], atol=1e-12)


def test_projection_matrix(golden):
    g = golden("triangulation")
    np.testing.assert_allclose(g["P_ours"], g["P"], rtol=1e-14, atol=1e-9)


@pytest.mark.parametrize("noise", [0, 2])
def test_two_view_triangulators(golden, noise):
    g = golden("triangulation")
    u, ps = g["u/noise%d" % noise], g["P"]
    for va, vb in ((0, 1), (0, 3), (1, 2)):
        for grp in range(3):
            tag = "noise%d/v%d%d/g%d" % (noise, va, vb, grp)
            us = np.stack([u[va, grp], u[vb, grp]])
            pp = np.stack([ps[va], ps[vb]])
            x, st = triangulation.iterative_ls_triangulation(us, pp)
            np.testing.assert_allclose(x, g[tag + "/iter_x"], atol=1e-7)
            np.testing.assert_array_equal(st, g[tag + "/iter_status"])
            x, _ = triangulation.linear_ls_triangulation(us, pp)
            np.testing.assert_allclose(x, g[tag + "/ls_x"], atol=1e-7)
            x, ok = triangulation.dlt_triangulation(us, pp)
            np.testing.assert_allclose(x, g[tag + "/eigen_x"], atol=1e-5)
            np.testing.assert_array_equal(ok, g[tag + "/eigen_status"])
            if noise == 0:      # analytic known answer: noise-free projections recover the point
                np.testing.assert_allclose(x, g["world"][grp], atol=1e-5)


def test_triangulation_status_codes(golden):
    g = golden("triangulation")
    x, st = triangulation.iterative_ls_triangulation(np.stack([g["behind/u0"], g["behind/u1"]]), g["P"][:2])
    np.testing.assert_array_equal(st, g["behind/status"])
    assert set(g["behind/status"].tolist()) >= {1, -1, -2}
    np.testing.assert_allclose(x, g["behind/x"], rtol=1e-7, atol=1e-5)


def test_multiview_dlt_known_answer(golden):
    g = golden("triangulation")
    u, ps = g["u/noise0"], g["P"]
    for grp in range(3):
        for fn in (triangulation.dlt_triangulation, triangulation.linear_ls_triangulation,
                   triangulation.iterative_ls_triangulation):
            x, _ = fn(u[:, grp], ps)
            np.testing.assert_allclose(x, g["world"][grp], atol=1e-5)


def test_crop_affine(golden):
    g = golden("geometry")
    for i, p in enumerate(g["affine/params"]):
        fwd = geometry.gen_trans_from_patch(p[0], p[1], p[2], p[3], 256, 256, p[4], p[5], inv=False)
        inv = geometry.gen_trans_from_patch(p[0], p[1], p[2], p[3], 256, 256, p[4], p[5], inv=True)
        np.testing.assert_allclose(fwd, g["affine/fwd"][i], rtol=1e-12, atol=1e-10)
        np.testing.assert_allclose(inv, g["affine/inv"][i], rtol=1e-12, atol=1e-10)
        out = geometry.trans_coords_from_patch_to_org_3d(g["decode/coords_patch"][i], p[0], p[1], p[2], p[3],
                                                         256, 256, 2000, 2000, scale=p[4], rot=p[5])
        np.testing.assert_allclose(out, g["decode/coords_img"][i], rtol=1e-12, atol=1e-9)


def scene_meta(n_group, j, seed, n_view=2):
    from epipolarpose_amd.synthetic import SyntheticScenes
    return SyntheticScenes(n_group=n_group, n_view=n_view, num_joints=j, seed=seed, noise_px=0.0)


@pytest.mark.parametrize("tag,n_group,j", [("h36m", 3, 17), ("mpii", 2, 16)])
def test_reprojection_and_ss_geometry(golden, tag, n_group, j):
    g = golden("geometry")
    sc = scene_meta(n_group, j, 31 + j)
    pt2d, pt3d = geometry.world_to_image_joints(sc.world[0], sc.meta["R"][0], sc.meta["T"][0], sc.meta["f"][0],
                                                sc.meta["c"][0])
    np.testing.assert_allclose(pt2d, g[tag + "/w2i/pt2d"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(pt3d, g[tag + "/w2i/pt3d"], rtol=1e-12, atol=1e-9)
    lab, wt = geometry.labels_from_global_coords(np.concatenate([sc.world] * 2), sc.meta)
    np.testing.assert_allclose(lab, g[tag + "/labels_from_world/label"], atol=1e-7)
    np.testing.assert_array_equal(wt, g[tag + "/labels_from_world/weight"])
    # the synthetic generator's labels are the reference's labels for exact 3-D
    np.testing.assert_allclose(sc.label, g[tag + "/labels_from_world/label"], atol=2e-6)
    np.testing.assert_array_equal(sc.label, g[tag + "/scene_label"])
    # decode -> triangulate -> re-project
    cp = g[tag + "/ss/coords_patch"]
    lab, wt, xw, kps = geometry.self_supervision(None, sc.meta, n_view=2, coords_patch=cp)
    np.testing.assert_allclose(kps, g[tag + "/ss/kps_img"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(xw, g[tag + "/ss/x_world"], atol=1e-6)
    np.testing.assert_allclose(lab, g[tag + "/ss/label"], atol=1e-7)


def test_full_self_supervision_from_logits(golden):
    g = golden("geometry")
    sc = scene_meta(2, 5, 77)
    lab, wt, _, _ = geometry.self_supervision(g["ss_full/logits"], sc.meta, n_view=2)
    np.testing.assert_allclose(lab, g["ss_full/label"], atol=5e-6)    # fp32 soft-argmax stage in the reference
    np.testing.assert_array_equal(wt, g["ss_full/weight"])


def test_get_max_preds(golden):
    g = golden("maxpreds")
    preds, maxvals = inference.get_max_preds(g["heatmaps"])
    np.testing.assert_array_equal(preds, g["preds"])
    np.testing.assert_array_equal(maxvals, g["maxvals"])


@pytest.mark.parametrize("tag,mpii", [("h36m", False), ("mpii", True)])
def test_evaluation_metrics(golden, tag, mpii):
    from oracle import evaluation
    g = golden("evaluation")
    gt = g[tag + "/gt_joints"]
    if mpii:        # the reference permutes its 17-joint records into MPII order (h36m.py:219-220)
        from epipolarpose_amd.dataset.h36m_eval import H36M_TO_MPII_PERM
        gt = gt[:, H36M_TO_MPII_PERM, :]
    metrics, per_sample, per_joint = evaluation.evaluate(g[tag + "/preds"], gt, g[tag + "/pelvis"], g[tag + "/fl"], g[tag + "/c_p"],
                                                         mpii_order=mpii)
    np.testing.assert_allclose(metrics, g[tag + "/metrics"], rtol=1e-10, atol=1e-9)
    assert list(evaluation.METRIC_NAMES) == g[tag + "/names"].tolist()
    np.testing.assert_allclose(metrics[0], g[tag + "/perf"], rtol=1e-12)


def test_procrustes(golden):
    from oracle import evaluation
    g = golden("evaluation")
    for i in range(6):
        t, b, c = evaluation.similarity_transform(g["procrustes/x"][i], g["procrustes/y"][i])
        np.testing.assert_allclose(t, g["procrustes/T"][i], atol=1e-12)
        np.testing.assert_allclose(b, g["procrustes/b"][i], rtol=1e-12)
        np.testing.assert_allclose(c, g["procrustes/c"][i], atol=1e-9)
