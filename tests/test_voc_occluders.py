"""The reference's occluder bank from its on-disk format (lib/utils/augmentation.py:9-58): a Pascal-VOC tree -> RGBA cut-outs.

``tests/golden/voc_fixture/`` is a small tree in that format and ``tests/golden/voc_occluders.npz`` what the reference's OWN ``load_occluders``
returned for it (tests/golden/make_voc_fixture.py: the reference's code running; its three OpenCV calls served by the oracle's restatements -- that layer
stays unpinned, as everywhere OpenCV is involved).  Here: the oracle's restatement and the product's reader must reproduce those arrays bit for bit, the
bank built from them must feed the batched crop kernel, and the kernel's output must equal the oracle's ``occlude_with_objects`` on the same draws."""
import os
import random

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "voc_fixture")


@pytest.fixture(scope="module")
def golden():
    z = np.load(os.path.join(HERE, "golden", "voc_occluders.npz"))
    return [z["occluder%d" % i] for i in range(int(z["count"]))]


def test_fixture_exercises_every_filter_of_the_reference(golden):
    # 9 objects in 4 annotations: an unsegmented image, two persons, a truncated and a difficult object, one below 500 mask pixels -> 3 survive
    assert len(golden) == 3
    assert [g.shape for g in golden] == [(40, 46, 4), (26, 56, 4), (38, 52, 4)]          # halves of 80 x 91, 51 x 111 and the odd 75 x 104 ... boxes
    for g in golden:
        assert g.dtype == np.uint8
        alpha = g[..., 3]
        assert alpha.max() == 255 and (alpha == 0).any()
        assert ((alpha > 0) & (alpha < 255)).any()                                        # the softened ring, blended by the halving


def test_oracle_restatement_matches_the_reference_run(golden):
    from oracle import imgproc as o_img
    got = o_img.load_occluders(ROOT)
    assert len(got) == len(golden)
    for a, b in zip(got, golden):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)


def test_product_reader_matches_the_reference_run(golden):
    from epipolarpose_amd.utils import augmentation as aug
    got = aug.load_occluders(ROOT)
    assert len(got) == len(golden)
    for a, b in zip(got, golden):
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    # no tree -> the procedural bank, as before (the reference's default path is its author's disk)
    assert len(aug.load_occluders("/nonexistent/VOCdevkit/VOC2012", count=5)) == 5
    assert len(aug.load_occluders(None, count=4)) == 4


def test_erosion_and_area_resize_restatements_agree_between_product_and_oracle():
    from epipolarpose_amd.utils import augmentation as aug
    from oracle import imgproc as o_img
    rng = np.random.default_rng(3)
    element = o_img.structuring_ellipse((8, 8))
    assert element.shape == (8, 8) and int(element.sum()) == 53 and element[0].tolist() == [0, 0, 0, 0, 1, 0, 0, 0]
    taps = aug._ellipse_element(8, 8)
    assert len(taps) == 53
    for shape in ((31, 17), (8, 8), (3, 50), (64, 64)):
        mask = np.where(rng.random(shape) > 0.15, 255, 0).astype(np.uint8)
        assert np.array_equal(aug._erode(mask, taps), o_img.erode(mask, element))
    solid = np.full((20, 20), 255, np.uint8)
    assert np.array_equal(aug._erode(solid, taps), solid)                                # the border does not erode (outside pixels do not take part)
    for (h, w), new in (((57, 43), (22, 28)), ((57, 43), (21, 29)), ((10, 10), (5, 5)), ((9, 7), (1, 1))):
        im = rng.integers(0, 256, (h, w, 4)).astype(np.uint8)
        assert np.array_equal(aug.resize_area(im, new), o_img.resize_area(im, new))
    im = rng.integers(0, 256, (12, 8, 4)).astype(np.uint8)                               # exact halving = the 2 x 2 mean rounded half up
    want = (im.reshape(6, 2, 4, 2, 4).astype(np.int32).sum(axis=(1, 3)) + 2) // 4
    assert np.array_equal(aug.resize_area(im, (4, 6)), want.astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("patch", [256, 384])
def test_crop_kernel_pastes_the_voc_bank_like_the_oracle(golden, patch):
    """epi_crop_patches_occluded with the bank read from the VOC tree against the oracle's occlude_with_objects on the same draws (uint8 stage: bytes exact).
    256 px: every occluder is shrunk (INTER_AREA); 384 px (configs[4]): im_scale_factor 1.5, so four draws in ten GROW it (INTER_LINEAR, augmentation.py:122)."""
    import torch
    from epipolarpose_amd import hip
    from epipolarpose_amd.utils import augmentation as aug
    from oracle import imgproc as o_img
    dev = torch.device("cuda:0")
    occluders = aug.load_occluders(ROOT)
    bank = aug.OccluderBank(occluders, dev)
    assert bank.count == 3 and bank.hw_host.tolist() == [[40, 46], [26, 56], [38, 52]]
    rng = np.random.default_rng(11)
    n, size = 8, patch + 44
    frames = rng.integers(0, 256, (n, size, size, 3)).astype(np.uint8)
    buf = torch.from_numpy(frames.reshape(-1)).to(dev)
    offs = torch.arange(n, dtype=torch.int64, device=dev) * (size * size * 3)
    hw = torch.tensor([[size, size]] * n, dtype=torch.int32, device=dev)
    place = np.zeros((n, aug.MAX_OCCLUDERS, 5), np.int32)
    trans, patches = [], []
    for i in range(n):
        place[i] = aug.draw_occlusion((patch, patch), bank.hw_host, np.random.RandomState(700 + i), random.Random(700 + i))
        p, t = o_img.generate_patch_image(frames[i], size / 2.0, size / 2.0, float(patch), float(patch), patch, patch, False, 1.0, 0.0)
        patches.append(p)
        trans.append(t)
    assert ((place[:, :, 0] >= 0).sum(axis=1) >= 1).all()
    assert set(place[:, :, 0][place[:, :, 0] >= 0].tolist()) == {0, 1, 2}                                          # every occluder of the tree is used
    used = place[place[:, :, 0] >= 0]
    grown = (used[:, 1] > bank.hw_host[used[:, 0], 1]) | (used[:, 2] > bank.hw_host[used[:, 0], 0])
    assert grown.any() == (patch > 256) and (~grown).any()
    raw = hip.crop_patches(buf, offs, hw, torch.from_numpy(np.stack(trans)).to(dev), patch, patch, occluders=bank.tensors(),
                           placements=torch.from_numpy(place).to(dev)).cpu().numpy()
    for i in range(n):
        image = o_img.occlude_with_objects(patches[i][:, :, ::-1], occluders, np.random.RandomState(700 + i), random.Random(700 + i))
        np.testing.assert_array_equal(raw[i], np.transpose(image, (2, 0, 1)).astype(np.float32), err_msg="sample %d" % i)
        assert (image != patches[i][:, :, ::-1]).any(axis=2).mean() > 0.002
