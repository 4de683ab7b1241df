"""CPU: the input pipeline's host side against golden vectors of the LIVE reference (tests/golden/make_golden.py
gen_pipeline: do_augmentation, paste_over, occlude_with_objects, get_single_patch_sample executed with seeded generators; the OpenCV
primitives -- imread, warpAffine, resize -- are stand-ins, see ref_shims.py / oracle/imgproc.py: that layer stays unpinned).
First the oracle restatement is pinned, then the product's host helpers (epipolarpose_amd/utils/img_utils.py, utils/augmentation.py)."""
import random

import numpy as np

from oracle import imgproc as o_img


def test_oracle_augmentation_draws_match_reference(golden):
    g = golden("pipeline")
    seed = int(g["aug/seed"])
    np_rng, py_rng = np.random.RandomState(seed), random.Random(seed)
    got = np.array([[d[0], d[1], float(d[2])] + list(d[3]) for d in (o_img.do_augmentation(np_rng, py_rng) for _ in range(64))])
    np.testing.assert_array_equal(got, g["aug/draws"])
    assert (g["aug/draws"][:, 1] == 0).mean() > 0.2 and (g["aug/draws"][:, 1] != 0).mean() > 0.4          # both rotation branches taken


def test_product_do_augmentation_matches_reference(golden):
    """the reference-signature function (module-level generator states) and the explicit-generator form"""
    from epipolarpose_amd.utils import img_utils as iu
    g = golden("pipeline")
    seed = int(g["aug/seed"])
    random.seed(seed)
    np.random.seed(seed)
    got = np.array([[d[0], d[1], float(d[2])] + list(d[3]) for d in (iu.do_augmentation() for _ in range(64))])
    np.testing.assert_array_equal(got, g["aug/draws"])
    np_rng, py_rng = np.random.RandomState(seed), random.Random(seed)
    got = np.array([[d[0], d[1], float(d[2])] + list(d[3]) for d in (iu.do_augmentation(np_rng, py_rng) for _ in range(64))])
    np.testing.assert_array_equal(got, g["aug/draws"])
    cfg = iu.get_default_augment_config()
    assert (cfg.scale_factor, cfg.rot_factor, cfg.color_factor, cfg.do_flip_aug, cfg.rot_aug_rate, cfg.flip_aug_rate) == (0.25, 30, 0.2, False, 0.6, 0.5)


def test_oracle_paste_over_and_occlusion_match_reference(golden):
    from epipolarpose_amd.utils.augmentation import load_occluders
    g = golden("pipeline")
    for t in range(8):
        res = g["paste/%d/dst" % t].copy()
        o_img.paste_over(g["paste/%d/src" % t], res, g["paste/%d/center" % t])
        np.testing.assert_array_equal(res, g["paste/%d/out" % t])
    occluders = load_occluders(seed=5, count=6)
    seed = int(g["occlude/seed"])
    got = o_img.occlude_with_objects(g["occlude/im"], occluders, np.random.RandomState(seed), random.Random(seed))
    np.testing.assert_array_equal(got, g["occlude/out"])
    assert (got != g["occlude/im"]).any(axis=2).mean() > 0.02
    # 384 px (configs[4]): the reference's occlude_with_objects with draws that GROW the occluder -- resize_by_factor's INTER_LINEAR branch (:122)
    seed = int(g["occlude384/seed"])
    draws = o_img.draw_occlusion(g["occlude384/im"].shape, len(occluders), np.random.RandomState(seed), random.Random(seed))
    assert any(f > 1.0 for _, f, _ in draws) and any(f <= 1.0 for _, f, _ in draws)
    got = o_img.occlude_with_objects(g["occlude384/im"], occluders, np.random.RandomState(seed), random.Random(seed))
    np.testing.assert_array_equal(got, g["occlude384/out"])


def test_resize_area_properties():
    rng = np.random.default_rng(1)
    im = rng.integers(0, 256, (12, 18, 4)).astype(np.uint8)
    np.testing.assert_array_equal(o_img.resize_area(im, (18, 12)), im)                                   # identity
    half = o_img.resize_area(im, (9, 6))                                                                   # integer factor: plain block mean, rounded half up
    want = (im.reshape(6, 2, 9, 2, 4).astype(np.int64).sum(axis=(1, 3)) * 2 + 4) // 8
    np.testing.assert_array_equal(half, want.astype(np.uint8))
    flat = np.full((7, 11, 3), 93, np.uint8)
    assert (o_img.resize_area(flat, (5, 3)) == 93).all()                                                  # weights sum to the area
    assert o_img.resize_by_factor(im, 0.5).shape == (6, 9, 4) and o_img.resize_by_factor(im, 1.7).shape == (20, 31, 4)      # round(12 * 1.7), round(18 * 1.7): grown with INTER_LINEAR


def _scene():
    from epipolarpose_amd.synthetic import SyntheticScenes, project
    sc = SyntheticScenes(n_group=2, n_view=2, num_joints=17, seed=31, augment=False)
    joints = []
    for i in range(sc.batch_size):
        v, g = divmod(i, 2)
        uv, xc = project(sc.world[g], sc.cams[v])
        joints.append(np.concatenate([uv, xc[:, 2:3] - xc[0, 2]], axis=1))
    return sc, np.stack(joints)


def test_oracle_single_patch_sample_matches_reference(golden):
    from epipolarpose_amd.dataset.synthetic_frames import render_frame
    from epipolarpose_amd.utils.augmentation import load_occluders
    g = golden("pipeline")
    sc, joints = _scene()
    occluders = load_occluders(seed=5, count=6)
    mean, std = np.array([123.675, 116.280, 103.530]), np.array([58.395, 57.120, 57.375])
    base = int(g["sample/seed_base"])
    for i in range(sc.batch_size):
        frame = render_frame(joints[i][:, :2], 1000, seed=500 + i)
        for occ in (0, 1):
            tag = "sample/%d/occ%d" % (i, occ)
            img, label, weight, scale, rot = o_img.single_patch_sample(
                frame, sc.meta["center_x"][i], sc.meta["center_y"][i], sc.meta["width"][i], sc.meta["height"][i], joints[i], np.ones((17, 3)),
                256, 256, 2000., mean, std, True, np.random.RandomState(base + i), random.Random(base + i), occluders=occluders if occ else None)
            # one float32 ulp of slack: the golden run evaluated (x - mean) / std under NumPy 2 (float64 scalar -> float64 arithmetic,
            # rounded once), the reference's pinned NumPy 1.16 -- which the oracle and the kernel follow -- evaluates it in float32
            np.testing.assert_allclose(img[:, ::8, ::8], g[tag + "/img_sub"], rtol=0, atol=5e-7, err_msg=tag)
            np.testing.assert_array_equal(label, g[tag + "/label"])
            np.testing.assert_array_equal(weight, g[tag + "/weight"])
            np.testing.assert_array_equal(np.array([scale, rot], np.float64), g[tag + "/scale_rot"])


def test_product_host_side_of_the_pipeline_matches_reference(golden):
    """do_augmentation -> draw_occlusion -> patch_affines_batch -> patch_labels_batch per sample, generators seeded as the golden run"""
    from epipolarpose_amd.utils import augmentation as aug
    from epipolarpose_amd.utils import img_utils as iu
    g = golden("pipeline")
    sc, joints = _scene()
    occluders = aug.load_occluders(seed=5, count=6)
    hw = np.array([[o.shape[0], o.shape[1]] for o in occluders])
    base = int(g["sample/seed_base"])
    for i in range(sc.batch_size):
        for occ in (0, 1):
            np_rng, py_rng = np.random.RandomState(base + i), random.Random(base + i)
            scale, rot, flip, color = iu.do_augmentation(np_rng, py_rng)
            if occ:
                place = aug.draw_occlusion((256, 256), hw, np_rng, py_rng)
                draws = o_img.draw_occlusion((256, 256, 3), len(occluders), np.random.RandomState(base + i), random.Random(base + i))
                assert (place[:, 0] >= 0).sum() >= 1
            tag = "sample/%d/occ%d" % (i, occ)
            np.testing.assert_array_equal(np.array([scale, rot], np.float64), g[tag + "/scale_rot"])
            trans = iu.patch_affines_batch([sc.meta["center_x"][i]], [sc.meta["center_y"][i]], [sc.meta["width"][i]], [sc.meta["height"][i]], 256, 256,
                                           [scale], [rot])
            label, weight = iu.patch_labels_batch(joints[i][None], np.ones((1, 17, 3)), trans, [sc.meta["width"][i]], [scale], 256, 256, 2000.)
            np.testing.assert_allclose(label[0], g[tag + "/label"], atol=2e-6)
            np.testing.assert_array_equal(weight[0], g[tag + "/weight"])
    # the placements the kernel receives reproduce the oracle's paste rectangles
    np_rng, py_rng = np.random.RandomState(77), random.Random(77)
    place = aug.draw_occlusion((256, 256), hw, np_rng, py_rng)
    draws = o_img.draw_occlusion((256, 256, 3), len(occluders), np.random.RandomState(77), random.Random(77))
    for row, (idx, factor, center) in zip(place, draws):
        small = o_img.resize_by_factor(occluders[idx], factor)
        c = np.round(center).astype(np.int32)
        assert tuple(row) == (idx, small.shape[1], small.shape[0], c[0] - small.shape[1] // 2, c[1] - small.shape[0] // 2)
