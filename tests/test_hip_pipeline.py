"""GPU: the input pipeline on the device -- ``epi_crop_patches_occluded`` against the oracle (bytes exact on the uint8
stage, float32 exact after colour scaling / normalisation) and against the live reference's get_single_patch_sample golden; the
frame-resident loader feeding ``train_integral``."""
import random

import numpy as np
import pytest
import torch

from oracle import imgproc as o_img

pytestmark = pytest.mark.gpu
MEAN, STD = np.array([123.675, 116.280, 103.530]), np.array([58.395, 57.120, 57.375])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _scene():
    from epipolarpose_amd.synthetic import SyntheticScenes, project
    sc = SyntheticScenes(n_group=2, n_view=2, num_joints=17, seed=31, augment=False)
    joints = []
    for i in range(sc.batch_size):
        v, g = divmod(i, 2)
        uv, xc = project(sc.world[g], sc.cams[v])
        joints.append(np.concatenate([uv, xc[:, 2:3] - xc[0, 2]], axis=1))
    return sc, np.stack(joints)


def test_occluded_crop_kernel_vs_oracle_and_reference_golden(golden, dev):
    from epipolarpose_amd import hip
    from epipolarpose_amd.dataset.synthetic_frames import render_frame
    from epipolarpose_amd.utils import augmentation as aug
    from epipolarpose_amd.utils import img_utils as iu
    g = golden("pipeline")
    sc, joints = _scene()
    occluders = aug.load_occluders(seed=5, count=6)
    bank = aug.OccluderBank(occluders, dev)
    frames = [render_frame(joints[i][:, :2], 1000, seed=500 + i) for i in range(sc.batch_size)]
    buf = torch.from_numpy(np.stack(frames).reshape(-1)).to(dev)
    offs = torch.arange(sc.batch_size, dtype=torch.int64, device=dev) * 3000000
    hw = torch.tensor([[1000, 1000]] * sc.batch_size, dtype=torch.int32, device=dev)
    base = int(g["sample/seed_base"])
    b = sc.batch_size
    scale, rot, color, place = np.zeros(b), np.zeros(b), np.zeros((b, 3), np.float32), np.zeros((b, aug.MAX_OCCLUDERS, 5), np.int32)
    for i in range(b):
        np_rng, py_rng = np.random.RandomState(base + i), random.Random(base + i)
        scale[i], rot[i], _, cs = iu.do_augmentation(np_rng, py_rng)
        color[i] = cs
        place[i] = aug.draw_occlusion((256, 256), bank.hw_host, np_rng, py_rng)
    # the kernel is fed the oracle's own affines (gen_trans_from_patch_cv: the 3-point solve); the vectorised closed form the loader uses
    # agrees to 1e-12, which the warp's fixed-point rounding turns into a handful of +-1 pixels per patch (checked in the loader test)
    trans = np.stack([o_img.generate_patch_image(frames[i][:2, :2], sc.meta["center_x"][i], sc.meta["center_y"][i], sc.meta["width"][i], sc.meta["height"][i],
                                                 256, 256, False, scale[i], rot[i])[1] for i in range(b)])
    assert np.abs(trans - iu.patch_affines_batch(sc.meta["center_x"], sc.meta["center_y"], sc.meta["width"], sc.meta["height"], 256, 256, scale, rot)).max() <= 1e-11
    args = (buf, offs, hw, torch.from_numpy(trans).to(dev), 256, 256)
    for occ in (0, 1):
        kw = dict(occluders=bank.tensors(), placements=torch.from_numpy(place).to(dev)) if occ else {}
        # uint8 stage (no colour scaling, no normalisation): bytes exact against the oracle
        raw = hip.crop_patches(*args, **kw).cpu().numpy()
        out = hip.crop_patches(*args, color_scale=torch.from_numpy(color).to(dev), mean=MEAN, std=STD, **kw).cpu().numpy()
        for i in range(b):
            patch, _ = o_img.generate_patch_image(frames[i], sc.meta["center_x"][i], sc.meta["center_y"][i], sc.meta["width"][i], sc.meta["height"][i],
                                                  256, 256, False, scale[i], rot[i])
            image = patch[:, :, ::-1]
            if occ:
                image = o_img.occlude_with_objects(image, occluders, *_after_aug(base + i))
            np.testing.assert_array_equal(raw[i], np.transpose(image, (2, 0, 1)).astype(np.float32), err_msg="sample %d occ %d" % (i, occ))
            if occ:
                assert (image != patch[:, :, ::-1]).any(axis=2).mean() > 0.005
            tag = "sample/%d/occ%d" % (i, occ)
            np.testing.assert_allclose(out[i][:, ::8, ::8], g[tag + "/img_sub"], rtol=0, atol=5e-7, err_msg=tag)     # the live reference (1 ulp: NumPy 2 vs 1.16)
    # bf16 NHWC output = the rounded float32 result
    o16 = hip.crop_patches(*args, color_scale=torch.from_numpy(color).to(dev), mean=MEAN, std=STD, dtype=torch.bfloat16, channels_last=True,
                           occluders=bank.tensors(), placements=torch.from_numpy(place).to(dev))
    assert o16.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(o16.float().cpu(), torch.from_numpy(out).to(torch.bfloat16).float())


def _after_aug(seed):
    """both generator streams of a sample after do_augmentation consumed its share (the occlusion draws continue the same streams)"""
    py = random.Random(seed)
    nr = np.random.RandomState(seed)
    o_img.do_augmentation(nr, py)
    return nr, py


def test_frame_loader_batches_and_training(dev):
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.function import train_integral
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.dataset.synthetic_frames import FramePatchLoader, SyntheticFrames
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd.optim import FusedAdam
    j, d, image = 17, 16, 64
    frames = SyntheticFrames(n_group=4, n_view=2, num_joints=j, seed=2, device=dev)
    loader = FramePatchLoader(frames, groups_per_batch=2, patch=image, augment=True, occlusion=True, seed=5)
    assert len(loader) == 2
    batches = list(loader)
    assert len(batches) == 2
    data, label, weight, meta = batches[0]
    assert data.shape == (4, 3, image, image) and data.dtype == torch.bfloat16 and data.is_contiguous(memory_format=torch.channels_last)
    assert label.shape == (4, 3 * j) and label.dtype == torch.float32 and weight.shape == label.shape
    assert set(meta) >= {"center_x", "center_y", "width", "height", "scale", "rot", "R", "T", "f", "c", "projection_matrix"}
    assert meta["projection_matrix"].shape == (4, 3, 4) and float(meta["scale"].min()) >= 0.75 and float(meta["scale"].max()) <= 1.25
    # the labels are what the oracle's get_single_patch_sample computes for the same draws
    loader2 = FramePatchLoader(frames, groups_per_batch=2, patch=image, augment=True, occlusion=True, seed=5, dtype=torch.float32, channels_last=False,
                               shuffle=False)
    np_rng, py_rng = np.random.RandomState(5), random.Random(5)
    d2, l2, w2, m2 = loader2.batch(np.array([0, 1]))
    occluders = loader2.bank
    from epipolarpose_amd.utils.augmentation import load_occluders
    occ_imgs = load_occluders(seed=5)
    for k, idx in enumerate([frames.index(v, g) for v in range(2) for g in (0, 1)]):
        sc = frames.scenes
        img, lab, wt, scale, rot = o_img.single_patch_sample(frames.frames_host[idx], sc.meta["center_x"][idx], sc.meta["center_y"][idx], sc.meta["width"][idx],
                                                             sc.meta["height"][idx], frames.joints[idx], np.ones((j, 3)), image, image, 2000., MEAN, STD,
                                                             True, np_rng, py_rng, occluders=occ_imgs)
        np.testing.assert_allclose(l2[k].cpu().numpy(), lab, atol=2e-6)
        assert float(m2["scale"][k]) == scale and float(m2["rot"][k]) == rot
        diff = d2[k].cpu().numpy() != img
        assert diff.mean() <= 2e-4, diff.mean()            # (see the kernel test: 1e-12 in the affine flips a few fixed-point roundings)
    # and the loop trains on it
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, d, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = 18
    cfg.PRINT_FREQ = 1
    torch.manual_seed(0)
    model = get_pose_net(cfg, is_train=True).to(dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)
    opt = FusedAdam(model, lr=1e-3)
    fixed = FramePatchLoader(frames, groups_per_batch=4, patch=image, augment=False, occlusion=False, seed=1, shuffle=False)
    first = train_integral(cfg, fixed, model, crit, opt, 0)
    for epoch in range(1, 12):
        last = train_integral(cfg, fixed, model, crit, opt, epoch)
    assert np.isfinite(last) and last < 0.8 * first, (first, last)
    aug_loss = train_integral(cfg, loader, model, crit, opt, 12)
    assert np.isfinite(aug_loss)
