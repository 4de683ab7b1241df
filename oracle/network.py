"""Oracle (test infrastructure): plain fp32 torch-CPU restatement of the pose network.

Functional re-expression of ``/root/reference/lib/models/pose3d_resnet.py`` driven directly by a
reference-format ``state_dict`` (key names of pose3d_resnet.py:93-126): stem (:186-189), residual
stages (BasicBlock :31-47 / Bottleneck :68-88, downsample :130-136), deconv head (:158-183) and the
final conv (:116-122).  Also the fp32 torch restatement of the criterion used as the CPU baseline.
Not imported by the product.
"""
import torch
import torch.nn.functional as F

STAGE_BLOCKS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottle", [3, 4, 6, 3]),
                101: ("bottle", [3, 4, 23, 3]), 152: ("bottle", [3, 8, 36, 3])}   # pose3d_resnet.py:288-292


def _bn(x, sd, prefix, training, stats):
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training and stats is not None:
        rm, rv = rm.clone(), rv.clone()
        stats[prefix] = (rm, rv)
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], training, 0.1, 1e-5)


def forward(sd, x, num_layers, num_deconv=3, training=True, new_stats=None):
    """Network forward.  sd: reference-format state dict of fp32 CPU tensors (may require grad)."""
    kind, blocks = STAGE_BLOCKS[num_layers]
    x = F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(x, sd, "bn1", training, new_stats))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, nblk in enumerate(blocks, start=1):
        for bi in range(nblk):
            p = "layer%d.%d" % (li, bi)
            stride = 2 if (li > 1 and bi == 0) else 1
            res = x
            if kind == "basic":
                o = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
                o = F.relu(_bn(o, sd, p + ".bn1", training, new_stats))
                o = F.conv2d(o, sd[p + ".conv2.weight"], None, padding=1)
                o = _bn(o, sd, p + ".bn2", training, new_stats)
            else:
                o = F.conv2d(x, sd[p + ".conv1.weight"])
                o = F.relu(_bn(o, sd, p + ".bn1", training, new_stats))
                o = F.conv2d(o, sd[p + ".conv2.weight"], None, stride=stride, padding=1)
                o = F.relu(_bn(o, sd, p + ".bn2", training, new_stats))
                o = F.conv2d(o, sd[p + ".conv3.weight"])
                o = _bn(o, sd, p + ".bn3", training, new_stats)
            if (p + ".downsample.0.weight") in sd:
                res = F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride)
                res = _bn(res, sd, p + ".downsample.1", training, new_stats)
            x = F.relu(o + res)
    for di in range(num_deconv):
        w = sd["deconv_layers.%d.weight" % (3 * di)]
        b = sd.get("deconv_layers.%d.bias" % (3 * di))
        x = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=0)   # k=4 (:146-148)
        x = F.relu(_bn(x, sd, "deconv_layers.%d" % (3 * di + 1), training, new_stats))
    pad = 1 if sd["final_layer.weight"].shape[-1] == 3 else 0
    return F.conv2d(x, sd["final_layer.weight"], sd["final_layer.bias"], padding=pad)


def softmax_integral(preds, num_joints):
    """fp32 torch restatement of integral_loss.py:71-86 (differentiable)."""
    b, c, h, w = preds.shape
    d = c // num_joints
    p = F.softmax(preds.reshape(b, num_joints, -1), dim=2).reshape(b, num_joints, d, h, w)
    ex = (p.sum(dim=(2, 3)) * torch.arange(w, dtype=p.dtype, device=p.device)).sum(dim=2, keepdim=True)
    ey = (p.sum(dim=(2, 4)) * torch.arange(h, dtype=p.dtype, device=p.device)).sum(dim=2, keepdim=True)
    ez = (p.sum(dim=(3, 4)) * torch.arange(d, dtype=p.dtype, device=p.device)).sum(dim=2, keepdim=True)
    return torch.cat((ex / w - 0.5, ey / h - 0.5, ez / d - 0.5), dim=2).reshape(b, num_joints * 3)


def joint_location_loss(preds, gt, vis, num_joints, kind="smoothl1", norm=False):
    """fp32 torch restatement of integral_loss.py:7-47 + :93-160."""
    pj = softmax_integral(preds, num_joints)
    if norm:
        pj = pj / torch.norm(pj, 1)
        gt = gt / torch.norm(gt, 1)
    diff = pj - gt
    if kind == "l1":
        out = diff.abs()
    elif kind == "l2":
        out = diff ** 2
    else:
        a = diff.abs()
        out = torch.where(a < 1.0, 0.5 * diff ** 2, a - 0.5)
    return (out * vis).sum() / len(pj)
