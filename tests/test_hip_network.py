"""GPU: the whole network on the training path (hand-written implicit-GEMM convolutions, fused BatchNorm / ReLU, MFMA head, all bf16 with
fp32 accumulation) against the golden vectors produced by the reference's PoseResNet (fp32).  The bar here is the bf16 one -- logits within
3 % of max|logit| in inference mode, loss within 0.5 %, gradients against the stock-bf16 yardstick where bf16 can reproduce them at all;
the fp32-grade proof on the same golden vectors is tests/test_hip_precise.py."""
import ast

import numpy as np
import pytest
import torch

from det_weights import fill_state_dict, seeded_array
from make_golden_cases import BIG_HEAD_STD, LOGIT_STRIDE, NETWORK_BIG_CASES, NETWORK_CASES

pytestmark = pytest.mark.gpu


def make_cfg(layers, image, joints, depth):
    from epipolarpose_amd.core.config import default_config
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = joints, depth, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    return cfg


def cosine(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


# Gradients.  At these golden states (random weights, batch 2 .. 4) the early-layer gradients of the network are a small difference of
# large, nearly equal terms (dz - mean(dz) under training-mode BatchNorm with a diffuse soft-argmax): NO bf16 run reproduces them -- stock
# PyTorch-ROCm kernels under bf16 autocast reach cosines of 0.0 .. 0.9 against the fp32 reference there.  So the bf16 path's gradients are
# NOT compared at the random-weight goldens any more (rounds 1-3 did, with a clause that skipped every pair stock bf16 could not reach):
#   * that the gradients are RIGHT at the golden states is proven by the fp32-grade mode of the same kernels against the live-reference
#     golden gradients, 37 .. 42 tensors per configuration (tests/test_hip_precise.py: cosine >= 0.999 at the bench shape, loss to 1e-7);
#   * the bf16 PRODUCT path is held, for EVERY parameter and with no escape clause, to that fp32-grade mode at a trained, well-conditioned
#     state of the same configuration (tests/trained_state.py) -- the check below.
# (min cosine, 5th percentile, median) floors per configuration, set from the values measured on MI355X (gpurun_out/network_trained_state.json)
# Measured on MI355X (tests/trained_state.py runs in deterministic mode: the same figures in every run of one build).  Ranges over the builds of round 4
# (atomics, ordered fp32 sums, ordered float64 sums -- a change in the LAST BITS of the BatchNorm sums moves a trained state of 50 .. 152 bf16 layers this
# far), as min / 5th percentile / median over ALL tensors | min / 5th percentile over the weight tensors proper (convolutions, deconvolutions, final layer):
#   r18            0.956-0.989 / 0.973-0.992 / 0.994-0.999 | 0.979-0.987 / 0.980-0.992      r50 (128 x 128, batch 4)  0.776-0.898 / 0.838-0.914 / 0.935-0.986 | 0.874-0.886 / 0.877-0.889
#   cfg1 (batch 2) 0.962-0.979 / 0.977-0.983 / 0.997-0.999 | 0.980-0.981 / 0.983-0.984      cfg2 (batch 4)            0.783-0.934 / 0.878-0.945 / 0.968-0.989 | 0.910-0.911 / 0.912-0.914
#   cfg5 (batch 8) 0.635-0.781 / 0.727-0.872 / 0.934-0.995 | 0.711-0.803 / 0.734-0.817  -- ResNet-152 at its golden batch of 2 stays ill-conditioned after 10
#   steps (0.07 / 0.32 / 0.71 with the losses equal to 6e-5), so configuration 5 is checked at batch 8.  The bench configuration at batch 32: tests/test_hip_precise.py.
# The minima are stem / layer-1 tensors in every case (first convolution, first BatchNorm biases): 50 .. 150 layers of bf16 activations behind them; the head
# (deconvolutions, final layer) is at >= 0.998 everywhere.  Floors = the lowest value seen minus 0.05: they catch a broken kernel (cosine ~0), not bf16 noise.
TRAINED_FLOORS = {"r18": (0.90, 0.92, 0.97), "r50": (0.70, 0.78, 0.88), "cfg1_r18_128": (0.90, 0.92, 0.97), "cfg2_r50_256": (0.72, 0.82, 0.92),
                  "cfg5_r152_384": (0.40, 0.65, 0.88)}
# (config 5: the MINIMUM over the 476 tensors of a 152-layer bf16 network at batch 8 is its noisiest statistic -- 0.635 and 0.540 on two builds of round 4 that
#  differ in the summation order of ONE deconvolution's backward-data (split K against unsplit), with 5 % quantile 0.727 / 0.711 and median 0.978 / 0.958; the floor
#  on it only catches a tensor that is plainly wrong (cosine ~0, as the fp32-grade checker's planning defect of this round produced), the quantiles carry the statement)
TRAINED_FLOORS_WEIGHTS = {"r18": (0.92, 0.93), "r50": (0.82, 0.82), "cfg1_r18_128": (0.93, 0.93), "cfg2_r50_256": (0.85, 0.86), "cfg5_r152_384": (0.65, 0.68)}
TRAINED_BATCH = {"cfg5_r152_384": 8}
TRAINED_REPORT = {}
LOSS_REL = {}
# loss of the bf16 path against the live reference's fp32 loss, forward in deterministic mode (measured: 4e-4 r18, 6.6e-3 / 8.1e-3 r50 -- a bf16 error of
# THAT random-weight state, identical in every run --, <= 2e-6 at the full configurations, whose N(0, 0.001) head makes the loss insensitive)
LOSS_RTOL = {"r50": 1.5e-2}
LOSS_RTOL_DEFAULT = 2e-3


def _check_network(g, name, layers, image, j, d, b, stride, head_std=None):
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss, softmax_integral_tensor
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from oracle import network as o_net
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    sub = (lambda a: a.reshape(-1)[::stride]) if stride else (lambda a: a)
    model = get_pose_net(make_cfg(layers, image, j, d), is_train=True).to(dev)
    shapes = {k: ast.literal_eval(s) for k, s in zip(g[name + "/keys"].tolist(), g[name + "/shapes"].tolist())}
    assert list(model.state_dict().keys()) == list(shapes.keys())
    model.load_state_dict(fill_state_dict(shapes, seed=1, head_std=head_std))
    x = torch.from_numpy(seeded_array("img/" + name, (b, 3, image, image))).to(dev)
    hm = image // 4
    model.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
    assert tuple(out.shape) == (b, j * d, hm, hm)
    ref = g[name + "/logits_eval"]
    ref_max = float(g[name + "/logits_eval_absmax"]) if stride else np.abs(ref).max()
    assert np.abs(sub(out.float().cpu().numpy()) - ref).max() <= 3e-2 * ref_max
    if stride:
        # decode with the explicit joint count (heat-map width != DEPTH_RES for configs 1 and 5).  With the He-scaled golden
        # weights the logits reach several hundred, so the soft-argmax is practically a hard arg-max and a bf16 rounding that
        # swaps two near-equal top voxels moves a coordinate by whole voxels: the yardstick is again the oracle network under
        # STOCK bf16 autocast -- the fraction of coordinates off by more than 1.5e-2 must not exceed stock's by more than 5 points (eval-mode statistics of the golden
        # weights are random, which makes 152 layers ill-conditioned: stock itself is off on 22 % of the coordinates there).
        xyz = softmax_integral_tensor(out, j, True, hm, hm, d).cpu().numpy()
        sd_e = {k: v.to(dev) for k, v in fill_state_dict(shapes, seed=1, head_std=head_std).items()}
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            stock_eval = o_net.forward(sd_e, x, layers, training=False)
        xyz_stock = softmax_integral_tensor(stock_eval.to(torch.bfloat16), j, True, hm, hm, d).cpu().numpy()
        ref_xyz = g[name + "/xyz_eval"]
        bad_ours, bad_stock = float((np.abs(xyz - ref_xyz) > 1.5e-2).mean()), float((np.abs(xyz_stock - ref_xyz) > 1.5e-2).mean())
        # two bf16 runs are two independent draws of "which coordinates flip": the difference of two such fractions over n coordinates
        # has a standard deviation of sqrt(2 p (1 - p) / n) -- 5.7 points for the 102 coordinates of config 5 at p = 0.21 -- so the
        # slack is three of those (never less than 5 points).  The fp32-grade mode of the same kernels decodes these very inputs to
        # the golden coordinates (tests/test_hip_precise.py), which is the statement about correctness; this one is about bf16.
        slack = max(0.05, 3.0 * float(np.sqrt(2.0 * bad_stock * (1.0 - bad_stock) / xyz.size)))
        assert bad_ours <= bad_stock + slack, (bad_ours, bad_stock, slack)
    model.train()
    from epipolarpose_amd import hip
    hip.set_deterministic(True)          # ordered BatchNorm sums (the reference's CUDNN.DETERMINISTIC): one fixed bf16 result instead of a run-to-run spread
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(x)
    finally:
        hip.set_deterministic(False)
    ref = g[name + "/logits_train"]
    ref_max = float(g[name + "/logits_train_absmax"]) if stride else np.abs(ref).max()
    # Training-mode BatchNorm over a small batch at 2x2 .. 12x12 spatial amplifies bf16 rounding through the whole depth, so
    # the yardstick is the oracle network run through STOCK PyTorch-ROCm kernels under the same bf16 autocast: our error
    # against the fp32 reference must not exceed 1.5x the stock-bf16 error (or 8 % of max|logit|, whichever is larger).
    sd = {k: v.to(dev) for k, v in fill_state_dict(shapes, seed=1, head_std=head_std).items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sd.update(params)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ologits = o_net.forward(sd, x, layers, training=True, new_stats={})
    ours, stock = sub(logits.float().detach().cpu().numpy()), sub(ologits.float().detach().cpu().numpy())
    err_ours, err_stock = np.abs(ours - ref).max(), np.abs(stock - ref).max()
    assert err_ours <= max(8e-2 * ref_max, 1.5 * err_stock), (err_ours, err_stock, ref_max)
    c_ours, c_stock = cosine(torch.from_numpy(ours), torch.from_numpy(ref)), cosine(torch.from_numpy(stock), torch.from_numpy(ref))
    assert c_ours >= min(0.99, c_stock - 0.01), (c_ours, c_stock)
    gt = torch.from_numpy(seeded_array("gt/" + name, (b, 3 * j), scale=0.2)).to(dev)
    loss = SmoothL1JointLocationLoss(num_joints=j)(logits, gt, torch.ones(b, 3 * j, device=dev))
    # bf16 activations.  In the DEFAULT mode two runs of the same code differ: the order of the statistics' atomics can flip a bf16 rounding in the
    # stem, which moves the logits of these random-weight goldens by 1 - 3 % (tools/debug_fwd_bimodal.py) and the loss by 1e-4 .. 5.9e-3 over
    # repeated runs of r50 -- rounds 1-3 held the loss to 1e-2 for that reason.  The forward above ran in deterministic mode (one fixed result,
    # bit-identical reruns: tests/test_hip_deterministic.py).  The fp32-grade mode of the same kernels holds this loss to 3e-7 (tests/test_hip_precise.py).
    LOSS_REL[name] = abs(loss.item() - float(g[name + "/loss"])) / abs(float(g[name + "/loss"]))
    np.testing.assert_allclose(loss.item(), g[name + "/loss"], rtol=LOSS_RTOL.get(name, LOSS_RTOL_DEFAULT))
    loss.backward()
    sd = model.state_dict()
    np.testing.assert_allclose(sd["bn1.running_mean"].cpu().numpy(), g[name + "/bn1.running_mean"], atol=2e-3)
    np.testing.assert_allclose(sd["deconv_layers.7.running_var"].cpu().numpy(), g[name + "/deconv_layers.7.running_var"], rtol=5e-2)
    grads = {k: p.grad for k, p in model.named_parameters()}
    for k, gk in grads.items():
        assert gk is not None and torch.isfinite(gk).all(), k
    # gradients of the bf16 product path: every parameter, against the fp32-grade mode, at a trained state of THIS configuration
    from trained_state import assert_not_worse_than_stock, bf16_vs_precise_at_trained_state
    rep = bf16_vs_precise_at_trained_state(layers, image, j, d, TRAINED_BATCH.get(name, b), tag="trained/" + name)
    TRAINED_REPORT[name] = {k: v for k, v in rep.items() if k not in ("cos", "cos_stock")}
    # (round 5) the teeth: per tensor class against the STOCK bf16 network on the same weights and batch -- cosine not more than 0.02 below stock's, norm
    # within 15 %.  The absolute floors below stay as a backstop only.
    assert_not_worse_than_stock(rep)
    lo, p05, med = TRAINED_FLOORS[name]
    assert rep["n_params"] == len(grads), (rep["n_params"], len(grads))
    assert abs(rep["loss_bf16"] - rep["loss_precise"]) <= 5e-3 * rep["loss_precise"], rep["worst"]
    assert rep["head_min_cos"] >= 0.99, rep["head_min_cos"]
    assert rep["min_cos"] >= lo and rep["p05_cos"] >= p05 and rep["median_cos"] >= med, (rep["min_cos"], rep["p05_cos"], rep["median_cos"], rep["worst"])
    wlo, wp05 = TRAINED_FLOORS_WEIGHTS[name]
    assert rep["min_cos_weights"] >= wlo and rep["p05_cos_weights"] >= wp05, (rep["min_cos_weights"], rep["p05_cos_weights"], rep["worst_weights"])


@pytest.fixture(scope="module", autouse=True)
def trained_report_file():
    yield
    import json
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "network_trained_state.json"), "w") as f:
        json.dump({"trained_state": TRAINED_REPORT, "loss_rel_deterministic": LOSS_REL}, f, indent=1, sort_keys=True)


@pytest.mark.parametrize("case", NETWORK_CASES, ids=[c[0] for c in NETWORK_CASES])
def test_network_vs_reference_golden(golden, case):
    name, layers, image, j, d, b = case
    _check_network(golden("network"), name, layers, image, j, d, b, 0)


@pytest.mark.parametrize("case", NETWORK_BIG_CASES, ids=[c[0] for c in NETWORK_BIG_CASES])
def test_network_full_configs_vs_reference_golden(golden, case):
    """BASELINE.json configs 1 (ResNet-18, 128x128), 2 (ResNet-50, 256x256: the bench shape) and 5 (ResNet-152, 384x384) against
    the reference network executed in fp32 (tests/golden/make_golden.py network_big); sub-sampled logits, full decode + loss."""
    name, layers, image, j, d, b = case
    _check_network(golden("network_big"), name, layers, image, j, d, b, LOGIT_STRIDE, head_std=BIG_HEAD_STD)


def test_training_reduces_loss_and_ss_step_runs():
    from epipolarpose_amd.core.function import train_step
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.hip import DeviceMeta
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd.synthetic import SyntheticScenes
    dev = torch.device("cuda:0")
    j, d, image = 5, 16, 64
    torch.manual_seed(0)
    model = get_pose_net(make_cfg(18, image, j, d), is_train=True).to(dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sc = SyntheticScenes(n_group=2, n_view=2, num_joints=j, seed=4)
    x = torch.from_numpy(sc.images(size=image)).to(dev)
    label, weight = torch.from_numpy(sc.label).to(dev), torch.from_numpy(sc.weight).to(dev)
    losses = [float(train_step(model, crit, opt, x, label, weight)) for _ in range(12)]
    assert losses[-1] < 0.7 * losses[0], losses
    meta = DeviceMeta(sc.meta, dev)
    l_ss = train_step(model, crit, opt, x, None, None, meta=meta, n_view=2)
    assert torch.isfinite(l_ss)


def test_fused_adam_matches_torch_adam():
    """Same arithmetic as torch.optim.Adam: fp32 path to 1e-6; the bf16-training-copy path tracks it to bf16 precision."""
    import copy
    import torch.nn as nn
    from epipolarpose_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    base = nn.Sequential(nn.Conv2d(8, 16, 3, padding=1, bias=False), nn.ReLU(), nn.Conv2d(16, 8, 1, bias=True), nn.Flatten(),
                         nn.Linear(8 * 6 * 5, 7)).to(dev).to(memory_format=torch.channels_last)
    x, y = torch.randn(4, 8, 6, 5, device=dev), torch.randn(4, 7, device=dev)

    def run(opt_factory, steps=4, autocast=False):
        m = copy.deepcopy(base)
        opt = opt_factory(m)
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = m(x)
            ((out.float() - y) ** 2).mean().backward()
            opt.step()
        return [p.detach().float().clone() for p in m.parameters()], opt
    ref, _ = run(lambda m: torch.optim.Adam(m.parameters(), lr=1e-2))
    got, _ = run(lambda m: FusedAdam(m, lr=1e-2, low_precision_convs=False))
    for a, b in zip(got, ref):
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
    got_lp, opt = run(lambda m: FusedAdam(m, lr=1e-2, low_precision_convs=True), autocast=True)
    assert len(opt.training_copies()) == 1                  # only the bias-free conv trains through a bf16 copy
    # Adam moves every weight by ~lr per step whatever the gradient's size, so a bf16-rounded gradient near zero can flip
    # a step's sign: elementwise the two runs may differ by up to 2*steps*lr, but on average they agree closely
    for a, b in zip(got_lp, ref):
        assert (a - b).abs().max().item() <= 2 * 4 * 1e-2 + 1e-3
        assert (a - b).abs().mean().item() <= 1e-2
    for i, lp in opt.training_copies().items():             # copy == bf16(master), same layout
        assert torch.equal(lp.detach().float(), opt._params[i].detach().to(torch.bfloat16).float())
    sd = opt.state_dict()
    opt.load_state_dict(sd)


def test_final_layer_bias_gradient_arrives_with_the_criterions_gradient():
    """Round 4: the criterion's backward kernel also delivers the per-channel sums of the logits' gradient, and the final layer takes them as its
    bias gradient (autograd's grad_output.sum((0, 2, 3)) of pose3d_resnet.py:116-122) instead of re-reading the gradient.  Held against the sums of
    the gradient tensor that actually flowed (captured by a tensor hook), and against the separate column-sum pass with the hand-over switched off;
    a gradient that is NOT the tensor the kernel wrote (scaled by a hook) must make the layer compute the sums itself."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    dev = torch.device("cuda:0")
    j, d, image, b = 4, 16, 64, 4
    torch.manual_seed(11)
    model = get_pose_net(make_cfg(18, image, j, d), is_train=True).to(dev)
    model.train()
    crit = SmoothL1JointLocationLoss(num_joints=j)
    gen = torch.Generator().manual_seed(12)
    x = torch.randn((b, 3, image, image), generator=gen).to(dev)
    gt = ((torch.rand((b, 3 * j), generator=gen) - 0.5) * 0.4).to(dev)
    wt = torch.ones(b, 3 * j, device=dev)
    glue = hip.glue()

    def run(hook=None):
        model.zero_grad(set_to_none=True)
        seen = []
        with torch.autocast("cuda", dtype=torch.bfloat16):
            preds = model(x)
        preds.register_hook(lambda g: seen.append(g.detach().clone()) if hook is None else hook(g, seen))
        crit(preds, gt, wt).backward()
        torch.cuda.synchronize()
        return model.final_layer.bias.grad.detach().double().clone(), seen[0].double()

    mode0 = glue.bias_grad_fuse_mode(1)
    try:
        glue.bias_sums_taken(True)
        got, flowed = run()
        assert glue.bias_sums_taken(True) == 1, "the final layer did not take the sums the criterion's kernel delivered"
        want, mag = flowed.sum(dim=(0, 2, 3)), flowed.abs().sum(dim=(0, 2, 3))
        assert ((got - want).abs() <= 1e-5 * mag + 1e-12).all()
        glue.bias_grad_fuse_mode(0)
        plain, flowed0 = run()
        assert glue.bias_sums_taken(True) == 0
        assert ((plain - flowed0.sum(dim=(0, 2, 3))).abs() <= 1e-5 * flowed0.abs().sum(dim=(0, 2, 3)) + 1e-12).all()
        glue.bias_grad_fuse_mode(1)

        def doubled(g, seen):            # the layer receives ANOTHER tensor than the one the kernel wrote: the offer must not be used
            seen.append((2 * g).detach().clone())
            return 2 * g
        got2, flowed2 = run(doubled)
        assert glue.bias_sums_taken(True) == 0
        assert ((got2 - flowed2.sum(dim=(0, 2, 3))).abs() <= 1e-5 * flowed2.abs().sum(dim=(0, 2, 3)) + 1e-12).all()
    finally:
        glue.bias_grad_fuse_mode(mode0)


def test_whole_network_used_twice_in_one_graph():
    """The model called on TWO inputs before one backward: every layer's weight receives two gradients in one pass, and the autograd engine adds
    the second to the first on the main stream while `.grad` is still undefined.  The second stream / deferred slab sums must not leave the
    first one in flight then (round-3 advisor finding: the head nodes DeconvBnAct / Conv1x1Bias and the stem decided with
    gradient_consumed_after_backward alone).  Reference: the two-stream, deferred configuration against one stream, immediate sums."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    dev = torch.device("cuda:0")
    j, d, image, b = 4, 16, 64, 4
    torch.manual_seed(3)
    model = get_pose_net(make_cfg(18, image, j, d), is_train=True).to(dev)
    model.train()
    crit = SmoothL1JointLocationLoss(num_joints=j)
    gen = torch.Generator().manual_seed(4)
    xa = torch.randn((b, 3, image, image), generator=gen).to(dev)
    xb = torch.randn((b, 3, image, image), generator=gen).to(dev)
    gt = ((torch.rand((b, 3 * j), generator=gen) - 0.5) * 0.4).to(dev)
    wt = torch.ones(b, 3 * j, device=dev)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def grads():
        model.load_state_dict(state)                 # (same BatchNorm buffers for every arm)
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = crit(model(xa), gt, wt) + crit(model(xb), gt, wt)
        loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}

    glue = hip.glue()
    defer0, side0, group0 = glue.defer_wgrad_reduce(False), glue.wgrad_stream_mode(0), glue.wgrad_group_mode(0)
    det0 = hip.set_deterministic(True)      # ordered BatchNorm sums: what is left between the arms is the summation order of the weight gradients
    try:
        ref = grads()
        for defer, side, group in ((True, 1, 2), (True, 1, 0), (False, 1, 2), (True, 0, 2)):
            glue.defer_wgrad_reduce(defer)
            glue.wgrad_stream_mode(side)
            glue.wgrad_group_mode(group)
            got = grads()
            for k in ref:
                assert torch.isfinite(got[k]).all(), (k, defer, side, group)
                # run-to-run noise of a bf16 network with atomically summed BatchNorm statistics is a few % of the norm; a gradient the engine
                # summed while one addend was still being written (or whose slab sum never ran) is O(1) wrong
                assert float((got[k] - ref[k]).norm()) <= 0.05 * float(ref[k].norm()) + 1e-9, (k, defer, side, group)
                assert cosine(got[k], ref[k]) >= 0.998 or float(ref[k].norm()) < 1e-9, (k, defer, side, group)
    finally:
        glue.defer_wgrad_reduce(defer0)
        glue.wgrad_stream_mode(side0)
        glue.wgrad_group_mode(group0)
        hip.set_deterministic(det0)
