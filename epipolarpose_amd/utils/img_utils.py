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
