"""Integral-regression criteria on MI355X -- host-side mirror of the reference's ``lib/core/integral_loss.py``.

Same public names, constructor arguments, ``forward(preds, gt_joints, gt_joints_vis)`` contract and error
behaviour as the reference (integral_loss.py:93-160, 187-216); the arithmetic runs in the HIP kernels of
``libepipolar_hip.so`` (one streaming pass over the logits forward, one read + one write backward) instead of
softmax + six reductions + autograd.  No CPU path: tensors must live on the GPU.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import hip


class _SoftArgmaxLoss(torch.autograd.Function):
    """loss = weighted_{l1,l2,smooth_l1}(softmax_integral(preds), gt, vis); d loss / d preds in one kernel."""

    @staticmethod
    def forward(ctx, preds, gt, vis, num_joints, kind, norm, size_average):
        xyz, rmax, rsum = hip.softargmax3d_fwd(preds, num_joints)
        loss, gxyz = hip.joint_loss(xyz, gt, vis, kind, norm, size_average, need_grad=True)
        ctx.save_for_backward(preds, rmax, rsum, xyz, gxyz)
        ctx.num_joints = num_joints
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        preds, rmax, rsum, xyz, gxyz = ctx.saved_tensors
        gscale = grad_out.to(torch.float32).contiguous()
        # channels-last logits: the kernel also delivers the per-channel sums of the gradient it writes -- the bias gradient of the final 1x1 convolution,
        # which that layer's backward then takes instead of re-reading the whole gradient (csrc/torch_glue.cpp take_column_sums; it checks that the
        # tensor it receives is this very memory, unmodified, and computes the sums itself otherwise)
        glue = hip.glue()
        if preds.dim() == 4 and preds.is_contiguous(memory_format=torch.channels_last) and not preds.is_contiguous() and glue.column_sums_wanted():
            sums = torch.zeros(preds.shape[1], dtype=torch.float32, device=preds.device)
            dlogits, delivered = hip.softargmax3d_bwd(preds, ctx.num_joints, rmax, rsum, xyz, gxyz, gscale, col_sums=sums)
            if delivered:
                glue.offer_column_sums(dlogits, sums)
            return dlogits, None, None, None, None, None, None
        dlogits = hip.softargmax3d_bwd(preds, ctx.num_joints, rmax, rsum, xyz, gxyz, gscale)
        return dlogits, None, None, None, None, None, None


class _SoftArgmax(torch.autograd.Function):
    """Differentiable ``softmax_integral_tensor`` (integral_loss.py:71-86)."""

    @staticmethod
    def forward(ctx, preds, num_joints):
        xyz, rmax, rsum = hip.softargmax3d_fwd(preds, num_joints)
        ctx.save_for_backward(preds, rmax, rsum, xyz)
        ctx.num_joints = num_joints
        return xyz

    @staticmethod
    def backward(ctx, grad_xyz):
        preds, rmax, rsum, xyz = ctx.saved_tensors
        return hip.softargmax3d_bwd(preds, ctx.num_joints, rmax, rsum, xyz, grad_xyz.to(torch.float32).contiguous()), None


def softmax_integral_tensor(preds, num_joints, output_3d, hm_width, hm_height, hm_depth):
    """Reference signature (integral_loss.py:71).  ``preds`` is [B, J*D, H, W] (or already [B, J, D*H*W])."""
    assert output_3d, 'Not Implemented!'            # integral_loss.py:79-80
    if preds.dim() != 4:
        preds = preds.reshape(preds.shape[0], num_joints * hm_depth, hm_height, hm_width)
    assert preds.shape[1] == num_joints * hm_depth and preds.shape[2] == hm_height and preds.shape[3] == hm_width
    return _SoftArgmax.apply(preds, num_joints)


def _assert_no_grad(tensor):
    assert not tensor.requires_grad, \
        "nn criterions don't compute the gradient w.r.t. targets - please mark these tensors as not requiring gradients"


class _JointLocationLoss(nn.Module):
    kind = None

    def __init__(self, num_joints, size_average=True, reduce=True, norm=False):
        super().__init__()
        self.size_average = size_average
        self.reduce = reduce
        self.num_joints = num_joints
        self.norm = norm

    def forward(self, preds, *args):
        gt_joints, gt_joints_vis = args[0], args[1]
        _assert_no_grad(gt_joints)
        _assert_no_grad(gt_joints_vis)
        if preds.shape[-3] % self.num_joints:
            raise ValueError("channel count %d is not a multiple of num_joints %d" % (preds.shape[-3], self.num_joints))
        return _SoftArgmaxLoss.apply(preds, gt_joints, gt_joints_vis, self.num_joints, self.kind, self.norm,
                                     self.size_average)


class L1JointLocationLoss(_JointLocationLoss):
    """integral_loss.py:118-138."""
    kind = "l1"


class SmoothL1JointLocationLoss(_JointLocationLoss):
    """integral_loss.py:140-160."""
    kind = "smoothl1"


class L2JointLocationLoss(_JointLocationLoss):
    """integral_loss.py:93-116 as intended (the reference's forward references an undefined attribute and
    cannot run; the working pieces -- softmax_integral_tensor + weighted_mse_loss -- are composed here)."""
    kind = "l2"


def get_loss_func(config):
    """integral_loss.py:162-168 (the reference passes ``output_3d`` where ``num_joints`` is expected; here the
    config must carry ``num_joints``)."""
    if config.loss_type == 'L1':
        return L1JointLocationLoss(config.num_joints)
    elif config.loss_type == 'L2':
        return L2JointLocationLoss(config.num_joints)
    else:
        assert 0, 'Error. Unknown heatmap type {}'.format(config.heatmap_type)


def generate_joint_location_label(patch_width, patch_height, joints, joints_vis):
    """integral_loss.py:170-177 -- host-side label codec, mutates ``joints`` in place like the reference."""
    joints[:, 0] = joints[:, 0] / patch_width - 0.5
    joints[:, 1] = joints[:, 1] / patch_height - 0.5
    joints[:, 2] = joints[:, 2] / patch_width
    return joints.reshape((-1)), joints_vis.reshape((-1))


def reverse_joint_location_label(patch_width, patch_height, joints):
    """integral_loss.py:179-185."""
    joints = joints.reshape((joints.shape[0] // 3, 3))
    joints[:, 0] = (joints[:, 0] + 0.5) * patch_width
    joints[:, 1] = (joints[:, 1] + 0.5) * patch_height
    joints[:, 2] = joints[:, 2] * patch_width
    return joints


def joint_location_result_device(patch_width, patch_height, preds, num_joints=None):
    """Device-resident decode: -> xyz [B,3J] f32 (normalised) without leaving the GPU (used by the SS step)."""
    if num_joints is None:
        num_joints = preds.shape[1] // preds.shape[-1]      # the reference assumes D == W (integral_loss.py:191-192)
    return hip.softargmax3d_fwd(preds.detach(), num_joints)[0]


def get_joint_location_result(patch_width, patch_height, preds, num_joints=None):
    """integral_loss.py:187-207 -> float64 ndarray [B, J, 4] (x, y, z in patch pixels, score 1).

    ``num_joints`` (extension) overrides the reference's D == W_heatmap assumption."""
    xyz = joint_location_result_device(patch_width, patch_height, preds, num_joints)
    coords = xyz.cpu().numpy().astype(float)
    coords = coords.reshape((coords.shape[0], coords.shape[1] // 3, 3))
    coords[:, :, 0] = (coords[:, :, 0] + 0.5) * patch_width
    coords[:, :, 1] = (coords[:, :, 1] + 0.5) * patch_height
    coords[:, :, 2] = coords[:, :, 2] * patch_width
    scores = np.ones((coords.shape[0], coords.shape[1], 1), dtype=float)
    return np.concatenate((coords, scores), axis=2)


def get_label_func():
    return generate_joint_location_label


def get_result_func():
    return get_joint_location_result


def merge_flip_func(a, b, flip_pair):
    return a


def get_merge_func(loss_config):
    return merge_flip_func
