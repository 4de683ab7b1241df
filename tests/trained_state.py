"""Shared by the GPU parity tests: every parameter gradient of the bf16 PRODUCT path against the fp32-grade mode of the same kernels
(epipolarpose_amd/models/precise.py) at a TRAINED state of the network.

Why a trained state: at the random-weight golden states the early-layer gradients of this network are a tiny difference of large, nearly equal
terms (dz - mean(dz) under training-mode BatchNorm with a diffuse soft-argmax) and NO bf16 run reproduces them (stock kernels under bf16
autocast reach cosines of 0.0 .. 0.9 against the fp32 reference there).  After a few optimisation steps the problem is well conditioned
(tools/probe_conditioning.py: 0.87 .. 0.95 on the early layers, 1.000 on the head from step 5 on, reference network on the CPU).  The fp32-grade
mode itself is pinned to the live reference's golden vectors (logits, loss, gradients, 20-step trajectories: tests/test_hip_precise.py), so it
is the on-device stand-in for the reference at states no fixture can hold (a ResNet-50 state is 137 MB)."""
import numpy as np
import torch

from det_weights import seeded_array


def cosine(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def tensor_class(key):
    """stem / layer1 .. layer4 / head: the classes the bf16 product path is held to the stock-bf16 yardstick on."""
    if key.startswith(("conv1", "bn1")):
        return "stem"
    if key.startswith("layer"):
        return key.split(".")[0]
    return "head"


def assert_not_worse_than_stock(rep, margin=0.02, norm_margin=0.15):
    """Per tensor class, on the class's CONCATENATED gradient (one vector per class: per-tensor cosines of the small BatchNorm tensors are two independent
    bf16 noise draws, the concatenation is not): cosine(ours, fp32-grade) >= cosine(stock bf16, fp32-grade) - margin, and the gradient norm within
    norm_margin of the fp32-grade norm or of the stock-bf16 norm's own deviation.  A dropped term or a wrong scale in one layer class shows up here however
    well-correlated it stays, because the library kernels do not make that mistake."""
    bad = []
    for name, c in sorted(rep["by_class"].items()):
        if c["cos_ours"] < c["cos_stock"] - margin:
            bad.append((name, "cosine", c["cos_ours"], c["cos_stock"]))
        dev_ours, dev_stock = abs(c["norm_ratio_ours"] - 1.0), abs(c["norm_ratio_stock"] - 1.0)
        if dev_ours > norm_margin and dev_ours > dev_stock + 0.5 * norm_margin:
            bad.append((name, "norm", c["norm_ratio_ours"], c["norm_ratio_stock"]))
    assert not bad, bad


def bf16_vs_precise_at_trained_state(layers, image, j, d, b, steps=10, seed=7, tag="trained"):
    """`steps` Adam steps in the fp32-grade mode from a seeded initialisation (torch's default backbone initialisation, the reference's own
    N(0, 0.001) head, pose3d_resnet.py:222-239), then one forward + backward of BOTH paths on the same weights and batch.
    Returns dict(loss_precise, loss_bf16, cos{param: cosine}, min_cos, p05_cos, median_cos, head_min_cos, n_params)."""
    from epipolarpose_amd import hip
    # ordered BatchNorm sums in BOTH paths and deterministic library convolutions (the fp32-grade mode leaves the 7x7 stem to MIOpen): removes most of
    # the run-to-run spread of the trained state (what is left comes from the library's stem kernels)
    was = hip.set_deterministic(True)
    cd, cb = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    try:
        return _compare(layers, image, j, d, b, steps, seed, tag)
    finally:
        hip.set_deterministic(was)
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = cd, cb


def _compare(layers, image, j, d, b, steps, seed, tag):
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models import precise
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    dev = torch.device("cuda:0")
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, d, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    torch.manual_seed(seed)
    model = get_pose_net(cfg, is_train=True).to(dev)
    init = model.state_dict()
    for k, v in init.items():
        if v.dim() == 4 and (k.startswith("deconv_layers") or k.startswith("final_layer")):
            v.normal_(0, 0.001)
    sd = {k: v.detach().clone().float() if v.dtype.is_floating_point else v.detach().clone() for k, v in init.items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    x = torch.from_numpy(seeded_array("img/" + tag, (b, 3, image, image))).to(dev)
    gt = torch.from_numpy(seeded_array("gt/" + tag, (b, 3 * j), scale=0.2)).to(dev)
    wt = torch.ones(b, 3 * j, device=dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-3)
    for _ in range(steps):
        opt.zero_grad()
        crit(precise.forward(sd, x, layers, training=True), gt, wt).backward()
        opt.step()
    state = {k: v.detach().clone() for k, v in sd.items()}
    opt.zero_grad()
    loss32 = crit(precise.forward(sd, x, layers, training=True), gt, wt)
    loss32.backward()
    model.load_state_dict(state)
    model.train()
    model.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = model(x)
    loss16 = crit(logits, gt, wt)
    loss16.backward()
    torch.cuda.synchronize()
    # the STOCK yardstick (round 5): the oracle network -- PyTorch-ROCm library kernels -- under the same bf16 autocast, from the same trained weights on the
    # same batch.  What bf16 activations cost against the fp32-grade gradients is measured, per tensor class, on a path that shares no kernel with ours.
    from oracle import network as o_net
    sd_stock = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.detach().clone()) for k, v in state.items()}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits_stock = o_net.forward(sd_stock, x, layers, training=True, new_stats={})
    loss_stock = crit(logits_stock.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), gt, wt)
    loss_stock.backward()
    torch.cuda.synchronize()
    cos, dims, cos_stock = {}, {}, {}
    classes = {}
    for k, p in model.named_parameters():
        if p.grad is not None:
            ours, ref, stock = p.grad.float().cpu(), sd[k].grad.cpu(), sd_stock[k].grad.float().cpu()
            cos[k] = cosine(ours, ref)
            cos_stock[k] = cosine(stock, ref)
            dims[k] = p.dim()
            c = classes.setdefault(tensor_class(k), {"ours": [], "ref": [], "stock": [], "n": 0})
            c["ours"].append(ours.reshape(-1).double()); c["ref"].append(ref.reshape(-1).double()); c["stock"].append(stock.reshape(-1).double())
            c["n"] += 1
    by_class = {}
    for name, c in classes.items():
        o, r, t = torch.cat(c["ours"]), torch.cat(c["ref"]), torch.cat(c["stock"])
        keys = [k for k in cos if tensor_class(k) == name]
        by_class[name] = {"n": c["n"], "cos_ours": cosine(o, r), "cos_stock": cosine(t, r), "norm_ratio_ours": float(o.norm() / (r.norm() + 1e-300)),
                          "norm_ratio_stock": float(t.norm() / (r.norm() + 1e-300)),
                          "median_cos_ours": float(np.median([cos[k] for k in keys])), "median_cos_stock": float(np.median([cos_stock[k] for k in keys])),
                          "min_cos_ours": float(min(cos[k] for k in keys)), "min_cos_stock": float(min(cos_stock[k] for k in keys))}
    vals = np.sort(np.asarray(list(cos.values())))
    head = [v for k, v in cos.items() if k.startswith(("deconv_layers", "final_layer"))]
    # the weight tensors proper (convolutions, deconvolutions, the final layer) apart from the BatchNorm scales / biases, whose gradients are the
    # small differences of large sums that bf16 activations hurt most (the minima over ALL tensors are always layer-1 / stem BatchNorm biases)
    wvals = np.sort(np.asarray([v for k, v in cos.items() if dims[k] >= 2]))
    return {"loss_precise": float(loss32.item()), "loss_bf16": float(loss16.item()), "loss_stock_bf16": float(loss_stock.item()), "cos": cos, "cos_stock": cos_stock,
            "by_class": by_class, "n_params": len(cos),
            "min_cos": float(vals[0]), "p05_cos": float(vals[len(vals) // 20]), "median_cos": float(np.median(vals)), "head_min_cos": float(min(head)),
            "n_weights": int(len(wvals)), "min_cos_weights": float(wvals[0]), "p05_cos_weights": float(wvals[len(wvals) // 20]),
            "median_cos_weights": float(np.median(wvals)), "worst": sorted(cos.items(), key=lambda kv: kv[1])[:5],
            "worst_weights": sorted(((k, v) for k, v in cos.items() if dims[k] >= 2), key=lambda kv: kv[1])[:3]}
