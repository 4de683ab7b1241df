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
