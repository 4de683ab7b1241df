"""Self-supervision step -- mirror of the reference's ``lib/utils/img_utils.py:141-243``.

``self_supervision(preds, meta)`` keeps the reference's contract (returns float32 ndarrays [B,3J]); the
device-resident variant ``self_supervision_device`` returns CUDA tensors so the training loop never leaves
the GPU (the reference does a blocking D2H copy and four nested Python loops, img_utils.py:169-209).
"""
import torch

from .. import hip
from ..core.integral_loss import joint_location_result_device

PATCH = 256.0          # img_utils.py:169,178: hard-coded patch size
RECT_3D = 2000.0       # img_utils.py:179: hard-coded depth box in mm


def self_supervision_device(preds, meta, n_view=2, method="iterative", num_joints=None, root_joint=0,
                            want_world=False):
    """preds [B,J*D,H,W] CUDA logits; meta: dict of per-sample tensors/arrays or a ``hip.DeviceMeta``.

    Batch is view-major (sample (v, g) at v*G+g; img_utils.py:194-199).  -> (label, weight[, X_world]) CUDA.
    """
    dmeta = meta if isinstance(meta, hip.DeviceMeta) else hip.DeviceMeta(meta, preds.device)
    xyz = joint_location_result_device(PATCH, PATCH, preds, num_joints)
    return hip.self_supervision(xyz, dmeta, n_view, method, PATCH, PATCH, RECT_3D, root_joint, want_world=want_world)


def self_supervision(preds, meta):
    """img_utils.py:166-190: -> (batch_label f32 ndarray [B,3J], batch_label_weight f32 ndarray [B,3J])."""
    label, weight = self_supervision_device(preds, meta, n_view=2, method="iterative")
    return label.cpu().numpy(), weight.cpu().numpy()


def trans_coords_from_patch_to_org_3d_batch(xyz, meta, patch_width=PATCH, patch_height=PATCH, rect_3d_width=RECT_3D):
    """Batched img_utils.py:150-155 on the device: xyz [B,3J] normalised soft-argmax output -> [B,J,3] f64."""
    dmeta = meta if isinstance(meta, hip.DeviceMeta) else hip.DeviceMeta(meta, xyz.device)
    return hip.decode_to_image(xyz, dmeta, patch_width, patch_height, rect_3d_width)


def triangulate(kps, meta, n_view=2, method="iterative"):
    """img_utils.py:193-209 on the device: kps [B,J,>=2] f64 CUDA -> [B,J,3] (group pose repeated per view)."""
    dmeta = meta if isinstance(meta, hip.DeviceMeta) else hip.DeviceMeta(meta, kps.device)
    x, _ = hip.triangulate(kps, dmeta.tensors["projection_matrix"], n_view, method)
    return torch.cat([x] * n_view, dim=0)


def get_batch_labels_from_global_coords(coords_3d_in_global_frame, meta, n_view=2):
    """img_utils.py:212-243 on the device: [B,J,3] (or [G,J,3]) world joints -> (label, weight) CUDA f32."""
    dmeta = meta if isinstance(meta, hip.DeviceMeta) else hip.DeviceMeta(meta, coords_3d_in_global_frame.device)
    g = dmeta.batch // n_view
    return hip.reproject_labels(coords_3d_in_global_frame[:g].contiguous(), dmeta, n_view, PATCH, PATCH, RECT_3D)


# ------------------------------------------------------------------------------------------------------------------
# Per-sample helpers under their reference names (img_utils.py:63-111,141-155).  Dataset-time host code (NumPy): the
# batched device versions above are what the training step uses; these keep the reference's call signatures for code
# written against ``lib.utils.img_utils`` (validation scripts, data preparation).
# ------------------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402


def rotate_2d(pt_2d, rot_rad):
    """img_utils.py:63-69 (result rounded to float32 as in the reference)."""
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    return np.array([pt_2d[0] * cs - pt_2d[1] * sn, pt_2d[0] * sn + pt_2d[1] * cs], dtype=np.float32)


def _affine_through_3_points(src, dst):
    """What ``cv2.getAffineTransform`` computes (img_utils.py:101,103): the 2x3 map through three point pairs, solved in
    float64 from the float32 points."""
    a = np.concatenate([np.asarray(src, np.float32).astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, np.asarray(dst, np.float32).astype(np.float64)).T


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):
    """img_utils.py:72-105: crop affine (centre, centre + down, centre + right; scale and rotation augmentation)."""
    center = np.array([c_x, c_y], dtype=np.float64)
    rot_rad = np.pi * rot / 180
    down = rotate_2d(np.array([0, src_height * scale * 0.5], dtype=np.float32), rot_rad)
    right = rotate_2d(np.array([src_width * scale * 0.5, 0], dtype=np.float32), rot_rad)
    src = np.stack([center, center + down, center + right]).astype(np.float32)
    dcen = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst = np.stack([dcen, dcen + np.array([0, dst_height * 0.5], dtype=np.float32),
                    dcen + np.array([dst_width * 0.5, 0], dtype=np.float32)]).astype(np.float32)
    return _affine_through_3_points(dst, src) if inv else _affine_through_3_points(src, dst)


def trans_point2d(pt_2d, trans):
    """img_utils.py:108-111."""
    return np.dot(trans, np.array([pt_2d[0], pt_2d[1], 1.]))[0:2]


def trans_coords_from_patch_to_org(coords_in_patch, c_x, c_y, bb_width, bb_height, patch_width, patch_height, scale=1.0, rot=0):
    """img_utils.py:141-147."""
    out = coords_in_patch.copy()
    trans = gen_trans_from_patch_cv(c_x, c_y, bb_width, bb_height, patch_width, patch_height, scale, rot, inv=True)
    out[:, 0:2] = np.asarray(coords_in_patch[:, 0:2], np.float64) @ trans[:, :2].T + trans[:, 2]
    return out


def trans_coords_from_patch_to_org_3d(coords_in_patch, c_x, c_y, bb_width, bb_height, patch_width, patch_height, rect_3d_width,
                                      rect_3d_height, scale=1.0, rot=0):
    """img_utils.py:150-155."""
    out = trans_coords_from_patch_to_org(coords_in_patch, c_x, c_y, bb_width, bb_height, patch_width, patch_height, scale, rot)
    out[:, 2] = coords_in_patch[:, 2] / patch_width * rect_3d_width
    return out


# ------------------------------------------------------------------------------------------------------------------
# Input pipeline on the GPU: img_utils.py:114-127 + :265-279, batched.
# ------------------------------------------------------------------------------------------------------------------
IMAGENET_MEAN = (123.675, 116.280, 103.530)        # JointIntegralDataset.py:67
IMAGENET_STD = (58.395, 57.120, 57.375)            # JointIntegralDataset.py:68


def generate_patch_images_device(frames, center_x, center_y, bb_width, bb_height, patch_width, patch_height, do_flip=None, scale=None,
                                 rot=None, color_scale=None, mean=IMAGENET_MEAN, std=IMAGENET_STD, dtype=torch.float32,
                                 channels_last=False, device=None):
    """The reference's per-sample ``generate_patch_image_cv`` + colour / normalisation stage for a whole batch in one launch.

    frames: list of uint8 arrays / tensors [H, W, 3] in BGR order (what ``cv2.imread`` yields), sizes may differ.  The per-sample
    affine (``gen_trans_from_patch_cv``; with ``do_flip`` the centre is mirrored first, :120-122) is set up on the host, the
    bilinear warp, BGR -> RGB, colour scaling, clipping and (x - mean) / std run on the device.
    -> (patches [B, 3, ph, pw] on the device, trans float64 ndarray [B, 2, 3])."""
    b = len(frames)
    device = device or torch.device("cuda", torch.cuda.current_device())
    scale = np.ones(b) if scale is None else np.asarray(scale, np.float64)
    rot = np.zeros(b) if rot is None else np.asarray(rot, np.float64)
    flip = np.zeros(b, np.int32) if do_flip is None else np.asarray(do_flip).astype(np.int32)
    hw = np.zeros((b, 2), np.int32)
    offs = np.zeros(b, np.int64)
    trans = np.zeros((b, 2, 3))
    flat, total = [], 0
    for i, f in enumerate(frames):
        t = f if isinstance(f, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(f))
        assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, "frames must be uint8 [H, W, 3] (BGR)"
        hw[i] = (t.shape[0], t.shape[1])
        offs[i] = total
        total += t.numel()
        flat.append(t.reshape(-1))
        cx = t.shape[1] - float(center_x[i]) - 1 if flip[i] else float(center_x[i])
        trans[i] = gen_trans_from_patch_cv(cx, float(center_y[i]), float(bb_width[i]), float(bb_height[i]), patch_width, patch_height,
                                           float(scale[i]), float(rot[i]), inv=False)
    buf = torch.cat([t.to(device, non_blocking=True) for t in flat])
    cs = None if color_scale is None else torch.as_tensor(np.asarray(color_scale, np.float32).reshape(b, 3), device=device)
    out = hip.crop_patches(buf, torch.from_numpy(offs).to(device), torch.from_numpy(hw).to(device), torch.from_numpy(trans).to(device),
                           int(patch_height), int(patch_width), do_flip=torch.from_numpy(flip).to(device), color_scale=cs, mean=mean, std=std,
                           dtype=dtype, channels_last=channels_last)
    return out, trans


def generate_patch_image_cv(cvimg, c_x, c_y, bb_width, bb_height, patch_width, patch_height, do_flip, scale, rot):
    """img_utils.py:114-127 for one frame: -> (uint8-valued patch [ph, pw, 3] in BGR as the reference returns it, trans)."""
    out, trans = generate_patch_images_device([cvimg], [c_x], [c_y], [bb_width], [bb_height], patch_width, patch_height, [do_flip], [scale],
                                              [rot], mean=None, std=None)
    rgb = out[0].permute(1, 2, 0).round().to(torch.uint8).cpu().numpy()
    return rgb[:, :, ::-1].copy(), trans[0]


def convert_cvimg_to_tensor(cvimg, occlusion_aug=True):
    """img_utils.py:130-138: HWC -> CHW float32 (no colour reordering there)."""
    return np.transpose(np.asarray(cvimg), (2, 0, 1)).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------
# Augmentation parameters and the per-sample pipeline (img_utils.py:17-39, 246-298), batched: host-side draws and label
# arithmetic (a few hundred bytes per sample), image work on the device.
# ------------------------------------------------------------------------------------------------------------------
import random as _random  # noqa: E402


class _AugmentConfig(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def get_default_augment_config():
    """img_utils.py:17-26."""
    config = _AugmentConfig()
    config.scale_factor = 0.25
    config.rot_factor = 30
    config.color_factor = 0.2
    config.do_flip_aug = False
    config.rot_aug_rate = 0.6       # possibility to rot aug
    config.flip_aug_rate = 0.5      # possibility to flip aug
    return config


def do_augmentation(np_rng=np.random, py_rng=_random):
    """img_utils.py:29-39: -> (scale, rot, do_flip, color_scale).  Draws from the ``numpy.random`` / ``random`` module states like the
    reference (pass a ``numpy.random.RandomState`` / ``random.Random`` pair for a private, seeded stream); the calls are made in the
    reference's order: the rotation's ``random()`` gate before its ``randn()``, no flip draw while ``do_flip_aug`` is off."""
    aug_config = get_default_augment_config()
    scale = np.clip(np_rng.randn(), -1.0, 1.0) * aug_config.scale_factor + 1.0
    rot = np.clip(np_rng.randn(), -2.0, 2.0) * aug_config.rot_factor if py_rng.random() <= aug_config.rot_aug_rate else 0
    do_flip = aug_config.do_flip_aug and py_rng.random() <= aug_config.flip_aug_rate
    c_up = 1.0 + aug_config.color_factor
    c_low = 1.0 - aug_config.color_factor
    color_scale = [py_rng.uniform(c_low, c_up), py_rng.uniform(c_low, c_up), py_rng.uniform(c_low, c_up)]
    return scale, rot, do_flip, color_scale


def patch_affines_batch(c_x, c_y, bb_width, bb_height, patch_width, patch_height, scale, rot):
    """``gen_trans_from_patch_cv(..., inv=False)`` (img_utils.py:72-105) for a whole batch: float64 [B, 2, 3].  The three-point affine in
    closed form (the patch-side triple is axis aligned) with the reference's float32 roundings of the points, vectorised."""
    c_x, c_y, bb_width, bb_height, scale, rot = (np.asarray(v, np.float64) for v in (c_x, c_y, bb_width, bb_height, scale, rot))
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    f32 = lambda v: np.asarray(v, np.float64).astype(np.float32).astype(np.float64)
    hh = f32((bb_height * scale * 0.5).astype(np.float32))      # np.array([0, src_height * scale * 0.5], dtype=np.float32)
    hw = f32((bb_width * scale * 0.5).astype(np.float32))
    down = np.stack([f32(0.0 * cs - hh * sn), f32(0.0 * sn + hh * cs)], axis=1)              # rotate_2d, rounded to float32
    right = np.stack([f32(hw * cs - 0.0 * sn), f32(hw * sn + 0.0 * cs)], axis=1)
    center = np.stack([c_x, c_y], axis=1)
    s0, s1, s2 = f32(center), f32(center + down), f32(center + right)
    dcx, dcy, dhw, dhh = f32(patch_width * 0.5), f32(patch_height * 0.5), f32(patch_width * 0.5), f32(patch_height * 0.5)
    ex, ey = (s2 - s0) / dhw, (s1 - s0) / dhh                   # images of the patch axes in the frame
    inv = np.zeros((len(c_x), 2, 3))
    inv[:, :, 0], inv[:, :, 1] = ex, ey
    inv[:, :, 2] = s0 - ex * dcx - ey * dcy
    lin = np.linalg.inv(inv[:, :, :2])
    out = np.zeros_like(inv)
    out[:, :, :2] = lin
    out[:, :, 2] = -np.einsum("bij,bj->bi", lin, inv[:, :, 2])
    return out


def patch_labels_batch(joints, joints_vis, trans, bb_width, scale, patch_width, patch_height, rect_3d_width, depth_in_image=False):
    """Steps 4-5 of get_single_patch_sample (img_utils.py:281-296) + generate_joint_location_label (integral_loss.py:170-177), batched:
    joints [B, J, 3] (u, v in the frame, root-relative depth in mm) -> (label f32 [B, 3J], weight f32 [B, 3J])."""
    joints = np.asarray(joints, np.float64)
    xy = np.einsum("bij,bkj->bki", trans[:, :, :2], joints[:, :, :2]) + trans[:, None, :, 2]
    ref = np.asarray(bb_width if depth_in_image else rect_3d_width, np.float64) * np.asarray(scale, np.float64)
    z = joints[:, :, 2] / np.reshape(ref, (-1, 1)) * patch_width
    label = np.stack([xy[:, :, 0] / patch_width - 0.5, xy[:, :, 1] / patch_height - 0.5, z / patch_width], axis=2)
    b = len(joints)
    return label.reshape(b, -1).astype(np.float32), np.asarray(joints_vis, np.float64).reshape(b, -1).astype(np.float32)
