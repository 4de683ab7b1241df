"""Oracle (test infrastructure): integral regression -- soft-argmax, joint losses, label codec.

float64 NumPy restatement of ``/root/reference/lib/core/integral_loss.py``.
Every function names the reference lines it follows.  Not imported by the product.
"""
import numpy as np

LOSS_KINDS = ("l1", "l2", "smoothl1")


def softmax_rows(logits, num_joints):
    """Global softmax over each joint's (D*H*W) voxels. integral_loss.py:71-74."""
    b = logits.shape[0]
    rows = np.asarray(logits, dtype=np.float64).reshape(b, num_joints, -1)
    rows = rows - rows.max(axis=2, keepdims=True)
    e = np.exp(rows)
    return e / e.sum(axis=2, keepdims=True)


def integral_expectation(prob, num_joints, x_dim, y_dim, z_dim):
    """Marginals + dot with arange. integral_loss.py:49-69 (volume is [B,J,z,y,x], :52).

    Returns raw expectations (E[x], E[y], E[z]) each [B, J] in voxel units.
    """
    vol = prob.reshape(prob.shape[0], num_joints, z_dim, y_dim, x_dim)
    px = vol.sum(axis=(2, 3))          # :54-55  accu_x over z then y
    py = vol.sum(axis=(2, 4))          # :56-57
    pz = vol.sum(axis=(3, 4))          # :58-59
    ex = (px * np.arange(x_dim, dtype=np.float64)).sum(axis=2)   # :61,65
    ey = (py * np.arange(y_dim, dtype=np.float64)).sum(axis=2)   # :62,66
    ez = (pz * np.arange(z_dim, dtype=np.float64)).sum(axis=2)   # :63,67
    return ex, ey, ez


def softmax_integral(logits, num_joints, hm_width, hm_height, hm_depth):
    """integral_loss.py:71-86 -> [B, 3J], (x,y,z) interleaved per joint, each in [-0.5, 0.5)."""
    prob = softmax_rows(logits, num_joints)
    ex, ey, ez = integral_expectation(prob, num_joints, hm_width, hm_height, hm_depth)
    xyz = np.stack([ex / hm_width - 0.5, ey / hm_height - 0.5, ez / hm_depth - 0.5], axis=2)  # :81-84
    return xyz.reshape(xyz.shape[0], num_joints * 3)                                          # :85


def _normalise(pred, target, norm):
    if not norm:
        return pred, target, None, None
    n_p = np.abs(pred).sum()      # torch.norm(x, 1) of the WHOLE tensor, integral_loss.py:9-11,22-24,35-37
    n_t = np.abs(target).sum()
    return pred / n_p, target / n_t, n_p, n_t


def joint_loss(pred, target, weights, kind, norm=False, size_average=True):
    """weighted_{mse,l1,smooth_l1}_loss, integral_loss.py:7-47.  Reduction is sum/len(input)=sum/B."""
    pred = np.asarray(pred, np.float64)
    target = np.asarray(target, np.float64)
    weights = np.asarray(weights, np.float64)
    p, t, _, _ = _normalise(pred, target, norm)
    diff = p - t
    if kind == "l1":
        out = np.abs(diff)                                  # :26
    elif kind == "l2":
        out = diff ** 2                                     # :13
    elif kind == "smoothl1":
        a = np.abs(diff)
        out = np.where(a < 1.0, 0.5 * diff ** 2, a - 0.5)   # :39-41
    else:
        raise ValueError(kind)
    total = (out * weights).sum()
    return total / pred.shape[0] if size_average else total


def joint_loss_grad(pred, target, weights, kind, norm=False, size_average=True):
    """Analytic d(loss)/d(pred) of :func:`joint_loss` (what autograd yields in the reference)."""
    pred = np.asarray(pred, np.float64)
    target = np.asarray(target, np.float64)
    weights = np.asarray(weights, np.float64)
    p, t, n_p, _ = _normalise(pred, target, norm)
    diff = p - t
    if kind == "l1":
        g = np.sign(diff)
    elif kind == "l2":
        g = 2.0 * diff
    else:
        g = np.where(np.abs(diff) < 1.0, diff, np.sign(diff))
    g = g * weights
    if size_average:
        g = g / pred.shape[0]
    if norm:
        # p = pred / ||pred||_1  ->  dp_i/dpred_j = delta_ij/n - pred_i*sign(pred_j)/n^2
        g = g / n_p - np.sign(pred) * (g * pred).sum() / (n_p * n_p)
    return g


def softmax_integral_backward(logits, num_joints, hm_width, hm_height, hm_depth, grad_xyz):
    """d/dlogits of :func:`softmax_integral` contracted with ``grad_xyz`` [B,3J].

    dlogit_i = p_i * ( gx*(x_i - E[x])/W + gy*(y_i - E[y])/H + gz*(z_i - E[z])/D )
    (the closed form of autograd through integral_loss.py:71-86).
    """
    b = logits.shape[0]
    prob = softmax_rows(logits, num_joints)                                # [B,J,N]
    ex, ey, ez = integral_expectation(prob, num_joints, hm_width, hm_height, hm_depth)
    g = np.asarray(grad_xyz, np.float64).reshape(b, num_joints, 3)
    zz, yy, xx = np.meshgrid(np.arange(hm_depth), np.arange(hm_height), np.arange(hm_width), indexing="ij")
    xx = xx.reshape(-1).astype(np.float64)
    yy = yy.reshape(-1).astype(np.float64)
    zz = zz.reshape(-1).astype(np.float64)
    t = (g[:, :, 0:1] * (xx[None, None] - ex[:, :, None]) / hm_width
         + g[:, :, 1:2] * (yy[None, None] - ey[:, :, None]) / hm_height
         + g[:, :, 2:3] * (zz[None, None] - ez[:, :, None]) / hm_depth)
    return (prob * t).reshape(np.shape(logits))


def joint_location_loss(logits, gt_joints, gt_joints_vis, num_joints, kind, norm=False):
    """{L1,SmoothL1,L2}JointLocationLoss.forward, integral_loss.py:93-160 (D = C // num_joints, :132,154)."""
    hm_width, hm_height = logits.shape[-1], logits.shape[-2]
    hm_depth = logits.shape[-3] // num_joints
    pred = softmax_integral(logits, num_joints, hm_width, hm_height, hm_depth)
    return joint_loss(pred, gt_joints, gt_joints_vis, kind, norm), pred


def joint_location_loss_backward(logits, gt_joints, gt_joints_vis, num_joints, kind, norm=False):
    hm_width, hm_height = logits.shape[-1], logits.shape[-2]
    hm_depth = logits.shape[-3] // num_joints
    pred = softmax_integral(logits, num_joints, hm_width, hm_height, hm_depth)
    g = joint_loss_grad(pred, gt_joints, gt_joints_vis, kind, norm)
    return softmax_integral_backward(logits, num_joints, hm_width, hm_height, hm_depth, g)


def generate_joint_location_label(patch_width, patch_height, joints, joints_vis):
    """integral_loss.py:170-177 (x/pw-.5, y/ph-.5, z/pw -- no offset on z).  Pure (no in-place write)."""
    j = np.array(joints, dtype=np.float64, copy=True)
    j[:, 0] = j[:, 0] / patch_width - 0.5
    j[:, 1] = j[:, 1] / patch_height - 0.5
    j[:, 2] = j[:, 2] / patch_width
    return j.reshape(-1), np.asarray(joints_vis).reshape(-1)


def reverse_joint_location_label(patch_width, patch_height, joints):
    """integral_loss.py:179-185."""
    j = np.array(joints, dtype=np.float64, copy=True).reshape(-1, 3)
    j[:, 0] = (j[:, 0] + 0.5) * patch_width
    j[:, 1] = (j[:, 1] + 0.5) * patch_height
    j[:, 2] = j[:, 2] * patch_width
    return j


def get_joint_location_result(patch_width, patch_height, logits, num_joints=None, fp32_stage=True):
    """integral_loss.py:187-207 -> float64 [B,J,4] (x,y,z in patch pixels, score 1).

    The reference infers D = W_heatmap (:191) and J = C // D (:192); ``num_joints`` overrides that
    (SURVEY section 7 hazard).  ``fp32_stage`` reproduces the float32 tensor -> float64 cast at :195-196.
    """
    hm_width, hm_height = logits.shape[-1], logits.shape[-2]
    if num_joints is None:
        hm_depth = hm_width
        num_joints = logits.shape[1] // hm_depth
    else:
        hm_depth = logits.shape[1] // num_joints
    pred = softmax_integral(logits, num_joints, hm_width, hm_height, hm_depth)
    if fp32_stage:
        pred = pred.astype(np.float32).astype(np.float64)
    c = pred.reshape(pred.shape[0], num_joints, 3).copy()
    c[:, :, 0] = (c[:, :, 0] + 0.5) * patch_width
    c[:, :, 1] = (c[:, :, 1] + 0.5) * patch_height
    c[:, :, 2] = c[:, :, 2] * patch_width
    return np.concatenate([c, np.ones((c.shape[0], num_joints, 1))], axis=2)
