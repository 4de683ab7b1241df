"""The reference's on-disk data format (lib/dataset/h36m.py:91-165: ``annot/*.pkl`` with pickled ``Camera`` objects, JPEG frames) through
``epipolarpose_amd.dataset.h36m`` -- against what the LIVE reference's ``H36M_Integral`` made of the same files (tests/golden/make_h36m_fixture.py ->
tests/golden/h36m_files.npz; the files themselves are committed under tests/golden/h36m_fixture/).  CPU: annotation forms, ``db`` order under the same
seeds, camera / crop meta, augmentation draws, labels and weights, JPEG decode checksums.  GPU: the image patches of ``__getitem__`` and of the batched
``FramePatchLoader`` over the HBM frame store."""
import copy
import os
import random
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "h36m_fixture")
CASES = (("train_tri", True, "train", True, 1.0), ("train_flat", False, "train", True, 0.5), ("valid", False, "valid", False, 1.0),
         ("valid_percam", False, "train", False, 1.0))
META_KEYS = ("center_x", "center_y", "width", "height", "scale", "rot", "R", "T", "f", "c", "projection_matrix")


def _cfg(tri, z_weight):
    from epipolarpose_amd.core.config import default_config
    cfg = default_config()
    cfg.MODEL.IMAGE_SIZE = [256, 256]
    cfg.DATASET.TRI, cfg.DATASET.NUM_CAMS, cfg.DATASET.OCCLUSION, cfg.DATASET.Z_WEIGHT = tri, 4, False, z_weight
    return cfg


def _dataset(name, tri, image_set, is_train, z_weight):
    from epipolarpose_amd.dataset import h36m
    random.seed(101)
    np.random.seed(101)
    return h36m(_cfg(tri, z_weight), ROOT, image_set, is_train)


def test_annotation_forms_and_camera_pickles():
    from epipolarpose_amd.dataset.h36m import H36M_Integral, read_annotations
    from epipolarpose_amd.utils.cameras import Camera
    per_cam, flat = read_annotations(ROOT, "train"), read_annotations(ROOT, "valid")
    assert isinstance(per_cam, dict) and sorted(per_cam) == [1, 2, 3, 4] and isinstance(flat, list) and len(flat) == 12
    rec = per_cam[2][1]
    assert {"image", "cam", "joints_3d", "joints_3d_vis", "center_x", "center_y", "width", "height", "flip_pairs", "parent_ids"} <= set(rec)
    cam = rec["cam"]
    assert isinstance(cam, Camera)                       # the reference's class path lib.utils.cameras.Camera, resolved by install_as_lib()
    # the pickled attributes are the reference's; our Camera recomputes the same projection matrix from (R, T, f, c) (cameras.py:120-131,149-150)
    again = Camera((cam.R, cam.T, cam.f, cam.c, cam.k, cam.p, cam.name))
    np.testing.assert_allclose(again.projection_matrix, cam.projection_matrix, rtol=0, atol=1e-9)
    assert isinstance(_dataset(*CASES[0]), H36M_Integral)
    from epipolarpose_amd.dataset import SyntheticH36M, h36m
    assert isinstance(h36m(_cfg(False, 1.0), "", "train", True, n_group=2), SyntheticH36M)       # no annotation file: the synthetic stand-in


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_db_order_labels_and_meta_match_the_live_reference(golden, case):
    name, tri, image_set, is_train, z_weight = case
    g = golden("h36m_files")
    ds = _dataset(*case)
    assert len(ds) == int(g[name + "/db_length"])
    if tri:
        order = np.array([[r["image"] for r in ds.db[c]] for c in range(4)])
    else:
        order = np.array([r["image"] for r in ds.db])
    np.testing.assert_array_equal(order, g[name + "/db_order"])
    for idx in range(min(len(ds), 3)):
        random.seed(500 + idx)
        np.random.seed(500 + idx)
        if tri:                                                                  # h36m.py:33-47, the same draws in the same order
            cam_1 = np.random.randint(ds.num_cams)
            cam_2 = ds.cam_config[cam_1][0] if random.random() <= 0.5 else ds.cam_config[cam_1][1]
            samples = [("%s/item%d/cam_1" % (name, idx), ds.host_sample(copy.deepcopy(ds.db[cam_1][idx]))),
                       ("%s/item%d/cam_2" % (name, idx), ds.host_sample(copy.deepcopy(ds.db[cam_2][idx])))]
        else:
            samples = [("%s/item%d" % (name, idx), ds.host_sample(copy.deepcopy(ds.db[idx])))]
        for tag, hs in samples:
            assert os.path.relpath(hs["meta"]["image"], ROOT) == str(g[tag + "/image"])
            np.testing.assert_allclose(hs["label"], g[tag + "/label"], rtol=0, atol=2e-6, err_msg=tag)
            np.testing.assert_array_equal(hs["weight"], g[tag + "/weight"])
            for k in META_KEYS:
                np.testing.assert_allclose(np.asarray(hs["meta"][k], np.float64).reshape(g[tag + "/meta/" + k].shape), g[tag + "/meta/" + k], rtol=0, atol=1e-9,
                                           err_msg=tag + " " + k)
    if name == "train_flat":
        assert any((g["train_flat/item%d/weight" % i] == 0.5).any() for i in range(3))          # DATASET.Z_WEIGHT reached the weights


def test_jpeg_decode_checksums(golden):
    """PIL decode -> BGR of the committed JPEG files = what the fixture generator's decoder saw (the same library here and on the GPU box)."""
    from epipolarpose_amd.dataset.h36m import decode_bgr, read_annotations
    g = golden("h36m_files")
    for rec in read_annotations(ROOT, "valid"):
        bgr = decode_bgr(os.path.join(ROOT, rec["image"]))
        assert bgr.dtype == np.uint8 and bgr.shape == (int(g["frame"]), int(g["frame"]), 3)
        assert np.uint32(zlib.crc32(bgr.tobytes())) == g["crc/" + rec["image"]], rec["image"]
    with pytest.raises(IOError):
        decode_bgr(os.path.join(ROOT, "images", "missing.jpg"))


def _close_patches(ours, ref_sub, tag):
    """our device warp against the reference's per-sample pipeline (cv2.warpAffine restated): equal up to the +-1 grey levels a 1e-12 difference in the
    affine turns into at a handful of pixels (tests/test_hip_pipeline.py)."""
    d = np.abs(ours[:, ::8, ::8] - ref_sub)
    assert d.max() <= 0.05 and (d > 1e-5).mean() <= 5e-3, (tag, float(d.max()), float((d > 1e-5).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES[:3], ids=[c[0] for c in CASES[:3]])
def test_items_match_the_live_reference_on_the_gpu(golden, case):
    name, tri = case[0], case[1]
    g = golden("h36m_files")
    ds = _dataset(*case)
    for idx in range(min(len(ds), 3)):
        random.seed(500 + idx)
        np.random.seed(500 + idx)
        item = ds[idx]
        bundles = [("%s/item%d/cam_1" % (name, idx), item["cam_1"]), ("%s/item%d/cam_2" % (name, idx), item["cam_2"])] if tri else [("%s/item%d" % (name, idx), item)]
        for tag, (img, label, weight, meta) in bundles:
            assert img.dtype == np.float32 and img.shape == (3, 256, 256) and label.dtype == np.float32 and weight.dtype == np.float32
            _close_patches(img, g[tag + "/img_sub"], tag)
            np.testing.assert_allclose(label, g[tag + "/label"], rtol=0, atol=2e-6)
            np.testing.assert_allclose(meta["projection_matrix"], g[tag + "/meta/projection_matrix"], rtol=0, atol=1e-9)


@pytest.mark.gpu
def test_frame_store_feeds_the_batched_loader(golden):
    """The files decoded once into HBM, a whole multi-view batch per launch (FramePatchLoader): without augmentation the patches, labels and meta of a
    batch equal the per-item path's, sample by sample, in view-major order."""
    import torch
    from epipolarpose_amd.dataset.synthetic_frames import FramePatchLoader
    ds = _dataset(*CASES[0])
    store = ds.frame_store()
    assert (store.n_view, store.n_group) == (4, 3) and store.frames.dtype == torch.uint8 and store.frames.is_cuda
    loader = FramePatchLoader(store, groups_per_batch=3, augment=False, shuffle=False, dtype=torch.float32, channels_last=False)
    data, label, weight, meta = next(iter(loader))
    assert tuple(data.shape) == (12, 3, 256, 256)
    ds.is_train = False                                    # the per-item path without the augmentation draw
    for v in range(4):
        for grp in range(3):
            img, lab, wt, m = ds.get_data(copy.deepcopy(ds.db[v][grp]))
            i = v * 3 + grp
            np.testing.assert_array_equal(data[i].cpu().numpy(), img)
            np.testing.assert_allclose(label[i].cpu().numpy(), lab, rtol=0, atol=1e-7)
            np.testing.assert_array_equal(weight[i].cpu().numpy(), wt)
            np.testing.assert_allclose(meta["projection_matrix"][i].numpy(), m["projection_matrix"], rtol=0, atol=0)
