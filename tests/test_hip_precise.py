"""GPU: the fp32-grade verification mode (epipolarpose_amd/models/precise.py: fp32 activations through the SAME bf16 MFMA kernels with
split operands) against the live reference's fp32 golden vectors -- logits, loss, running statistics and parameter gradients at
BASELINE.json's configurations 1 and 2 (the bench shape) held to an fp32-grade bar -- and then, as the on-device fp32 yardstick,
against the bf16 training path at a trained (well-conditioned) state.  Every measured deviation is written to
gpurun_out/precise_parity.json."""
import ast
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from det_weights import fill_state_dict, seeded_array
from make_golden_cases import BIG_HEAD_STD, LOGIT_STRIDE, NETWORK_BIG_CASES, TRAJECTORY_CASES, TRAJECTORY_HEAD_STD, grad_stride

pytestmark = pytest.mark.gpu

REPORT = {}


@pytest.fixture(scope="module", autouse=True)
def report_file():
    yield
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "precise_parity.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


@pytest.fixture(autouse=True)
def deterministic_sums():
    """Every test of this file runs with ordered BatchNorm sums (the library's deterministic mode) and deterministic library convolutions (the
    fp32-grade mode leaves the 7x7 stem to MIOpen): the figures asserted here are then the same in every run -- with atomics the fp32-grade
    gradients of ResNet-152 at batch 2 moved by 1e-3 .. 4e-3 in cosine from run to run, across the tolerance."""
    from epipolarpose_amd import hip
    was = hip.set_deterministic(True)
    cd, cb = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    yield
    hip.set_deterministic(was)
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = cd, cb


def cosine(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def rel(a, b):
    """max |a - b| / max |b| in float64"""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def gpu_state(shapes, seed, head_std, dev):
    sd = {k: v.to(dev) for k, v in fill_state_dict(shapes, seed=seed, head_std=head_std).items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    return sd


# ---- operators: the split-operand GEMMs, fp32 BatchNorm and max-pool against float64 on the CPU --------------------------------------
@pytest.mark.parametrize("pieces,tol", [(2, 4e-5), (3, 2e-6)])
def test_precise_convolutions_vs_float64(pieces, tol):
    from epipolarpose_amd.models import precise
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(31 + pieces)
    worst = {}
    # (Cin, Cout, k, stride, H): 1x1, 3x3 stride 1 | 2 (patch-eligible geometry included), the stride-2 projection
    # ... and two 64-channel layers at 4 x 128 x 128 = 65 536 rows: from that row count on a bf16 result with N <= 64 takes the 256 x 64 tile,
    # which has no fp32-result instantiation (rounds 2-3 planned fp32 results for it all the same: wrong layer-1 results from batch 16 on)
    for cin, cout, k, stride, h in ((64, 64, 1, 1, 16), (64, 64, 3, 1, 16), (128, 128, 3, 2, 16), (256, 512, 1, 2, 8), (512, 128, 1, 1, 8), (64, 256, 1, 1, 16),
                                    (64, 64, 1, 1, 128), (256, 64, 1, 1, 128)):
        pad = k // 2
        x = torch.randn((4, cin, h, h), generator=gen)
        w = torch.randn((cout, cin, k, k), generator=gen) * (2.0 / (cin * k * k)) ** 0.5
        xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
        ref = F.conv2d(xd, wd, stride=stride, padding=pad)
        dy = torch.randn(ref.shape, generator=gen)
        ref.backward(dy.double())
        xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = precise.conv2d(xg, wg, stride, pad, pieces)
        y.backward(dy.to(dev))
        for what, got, want in (("y", y, ref), ("dx", xg.grad, xd.grad), ("dw", wg.grad, wd.grad)):
            e = rel(got.detach().cpu(), want.detach())
            worst[what] = max(worst.get(what, 0.0), e)
            assert e <= tol, (cin, cout, k, stride, what, e)
    # transposed convolution and the final 1x1 convolution with bias
    x = torch.randn((2, 128, 8, 8), generator=gen)
    w = torch.randn((128, 64, 4, 4), generator=gen) * 0.05
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv_transpose2d(xd, wd, stride=2, padding=1)
    dy = torch.randn(ref.shape, generator=gen)
    ref.backward(dy.double())
    xg, wg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = precise.deconv4x4s2(xg, wg, pieces)
    y.backward(dy.to(dev))
    for what, got, want in (("deconv y", y, ref), ("deconv dx", xg.grad, xd.grad), ("deconv dw", wg.grad, wd.grad)):
        worst[what] = rel(got.detach().cpu(), want.detach())
        assert worst[what] <= tol, (what, worst[what])
    x = torch.randn((2, 256, 16, 16), generator=gen)
    w = torch.randn((136, 256, 1, 1), generator=gen) * 0.05
    bias = torch.randn(136, generator=gen)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), bias.double().requires_grad_(True)
    ref = F.conv2d(xd, wd, bd)
    dy = torch.randn(ref.shape, generator=gen)
    ref.backward(dy.double())
    xg, wg, bg = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), bias.to(dev).requires_grad_(True)
    y = precise.conv1x1_bias(xg, wg, bg, pieces)
    y.backward(dy.to(dev))
    for what, got, want in (("final y", y, ref), ("final dx", xg.grad, xd.grad), ("final dw", wg.grad, wd.grad), ("final db", bg.grad, bd.grad)):
        worst[what] = rel(got.detach().cpu(), want.detach())
        assert worst[what] <= tol, (what, worst[what])
    REPORT["operators/pieces%d" % pieces] = worst


def test_precise_batchnorm_and_maxpool_vs_float64():
    from epipolarpose_amd.models import precise
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(41)
    worst = {}
    for c, h, relu, has_res in ((64, 16, True, False), (256, 8, True, True), (32, 12, False, False), (1024, 4, True, True)):
        x = torch.randn((4, c, h, h), generator=gen) * 2 + 0.3
        res = torch.randn((4, c, h, h), generator=gen) if has_res else None
        gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.1
        rm, rv = torch.randn(c, generator=gen) * 0.1, torch.rand(c, generator=gen) + 0.5
        xd, gd, bd = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        rd = res.double().requires_grad_(True) if has_res else None
        rmd, rvd = rm.double().clone(), rv.double().clone()
        ref = F.batch_norm(xd, rmd, rvd, gd, bd, True, 0.1, 1e-5)
        if has_res:
            ref = ref + rd
        if relu:
            ref = F.relu(ref)
        dy = torch.randn(ref.shape, generator=gen)
        ref.backward(dy.double())
        sd = {"bn.weight": gamma.to(dev).requires_grad_(True), "bn.bias": beta.to(dev).requires_grad_(True), "bn.running_mean": rm.to(dev),
              "bn.running_var": rv.to(dev), "bn.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
        xg = x.to(dev).requires_grad_(True)
        rg = res.to(dev).requires_grad_(True) if has_res else None
        y = precise.bn_act(xg, sd, "bn", residual=rg, relu=relu, training=True)
        y.backward(dy.to(dev))
        checks = [("y", y, ref), ("dx", xg.grad, xd.grad), ("dgamma", sd["bn.weight"].grad, gd.grad), ("dbeta", sd["bn.bias"].grad, bd.grad),
                  ("running_mean", sd["bn.running_mean"], rmd), ("running_var", sd["bn.running_var"], rvd)]
        if has_res:
            checks.append(("dres", rg.grad, rd.grad))
        for what, got, want in checks:
            e = rel(got.detach().cpu(), want.detach())
            worst[what] = max(worst.get(what, 0.0), e)
            assert e <= 2e-5, (c, h, what, e)
        assert int(sd["bn.num_batches_tracked"].item()) == 1
        # inference mode
        ye = precise.bn_act(x.to(dev), sd, "bn", residual=res.to(dev) if has_res else None, relu=relu, training=False)
        refe = F.batch_norm(x.double(), sd["bn.running_mean"].cpu().double(), sd["bn.running_var"].cpu().double(), gamma.double(), beta.double(), False, 0.1, 1e-5)
        refe = F.relu(refe + (res.double() if has_res else 0)) if relu else refe + (res.double() if has_res else 0)
        assert rel(ye.cpu(), refe) <= 2e-6
    x = torch.randn((3, 64, 18, 18), generator=gen)
    xd = x.double().requires_grad_(True)
    ref = F.max_pool2d(xd, 3, 2, 1)
    dy = torch.randn(ref.shape, generator=gen)
    ref.backward(dy.double())
    xg = x.to(dev).requires_grad_(True)
    y = precise.maxpool3x3s2(xg)
    y.backward(dy.to(dev))
    assert torch.equal(y.detach().cpu().double(), ref.detach())
    assert rel(xg.grad.cpu(), xd.grad) <= 1e-6                 # (an input pixel selected by two windows: one fp32 add, in either order)
    REPORT["operators/batchnorm"] = worst


# ---- the whole network against the live reference's fp32 golden vectors --------------------------------------------------------------
# Bars (fp32 grade), with what was measured on MI355X at three bf16 pieces per operand (gpurun_out/precise_parity.json):
#   logits   |ours - ref| <= 1e-4 * max|ref|   eval 3e-7 / 4e-7, training 8e-7 (config 1) / 2e-5 (bench shape: batch-4 BatchNorm over 50 layers)
#   loss     rel 1e-5  -- BASELINE.md section 5's fp32 tolerance --                       measured 0 (bit-identical) / 1.2e-7
#   running statistics rel 2e-6                                                            measured <= 5e-8
#   every stored parameter gradient: cosine >= 0.999, norm within 0.5 %                  measured 1.0000000 (config 1) / 0.99951 .. 1.0 (bench shape,
#   where the early-layer gradients are so ill-conditioned that bf16 -- ours and stock alike -- reaches 0.06 .. 0.1 on the same vectors)
PRECISE_CASES = [c for c in NETWORK_BIG_CASES if c[0] in ("cfg1_r18_128", "cfg2_r50_256", "cfg5_r152_384")]


@pytest.mark.parametrize("case", PRECISE_CASES, ids=[c[0] for c in PRECISE_CASES])
def test_precise_network_vs_reference_golden(golden, case):
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models import precise
    name, layers, image, j, d, b = case
    g = golden("network_big")
    dev = torch.device("cuda:0")
    shapes = {k: ast.literal_eval(s) for k, s in zip(g[name + "/keys"].tolist(), g[name + "/shapes"].tolist())}
    sub = lambda a: a.reshape(-1)[::LOGIT_STRIDE]
    rep = {}
    x = torch.from_numpy(seeded_array("img/" + name, (b, 3, image, image))).to(dev)
    sd = gpu_state(shapes, 1, BIG_HEAD_STD, dev)
    with torch.no_grad():
        le = precise.forward(sd, x, layers, training=False, pieces=3)
    ref, ref_max = torch.from_numpy(g[name + "/logits_eval"]), float(g[name + "/logits_eval_absmax"])
    got = sub(le.contiguous().cpu())
    rep["logits_eval_err_over_max"] = float((got - ref).abs().max()) / ref_max
    rep["logits_eval_cos"] = cosine(got, ref)
    assert rep["logits_eval_err_over_max"] <= 1e-5 and rep["logits_eval_cos"] >= 1 - 1e-9, rep
    # the decoded joints (soft-argmax over practically one-hot volumes at these golden weights: the bf16 path flips 20-27 % of config 5's
    # coordinates by whole voxels, stock bf16 kernels likewise -- tests/test_hip_network.py; the fp32-grade path must flip none)
    from epipolarpose_amd.core.integral_loss import softmax_integral_tensor
    xyz = softmax_integral_tensor(le, j, True, image // 4, image // 4, d).cpu().numpy()
    rep["xyz_eval_max_err"] = float(np.abs(xyz - g[name + "/xyz_eval"]).max())
    assert rep["xyz_eval_max_err"] <= 1e-3, rep
    logits = precise.forward(sd, x, layers, training=True, pieces=3)         # three bf16 pieces per operand: products to ~2^-24, fp32 itself
    ref, ref_max = torch.from_numpy(g[name + "/logits_train"]), float(g[name + "/logits_train_absmax"])
    got = sub(logits.detach().contiguous().cpu())
    rep["logits_train_err_over_max"] = float((got - ref).abs().max()) / ref_max
    rep["logits_train_cos"] = cosine(got, ref)
    # Yardstick for the deepest network (config 5: batch-2 BatchNorm through 152 layers, where fp32 itself is no longer reproducible to
    # 1e-4 between two implementations): the oracle network on STOCK fp32 kernels (MIOpen / rocBLAS, no autocast) against the same golden
    # vectors.  The fp32-grade path must be as close as twice that, and never worse than 1e-4 where fp32 is.
    from oracle import network as o_net
    sd32 = gpu_state(shapes, 1, BIG_HEAD_STD, dev)
    for k, v in sd32.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    prev = torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    try:
        slogits = o_net.forward(sd32, x, layers, training=True, new_stats={})
    finally:
        torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = prev
    sgot = sub(slogits.detach().float().contiguous().cpu())
    rep["stock_fp32_logits_train_err_over_max"] = float((sgot - ref).abs().max()) / ref_max
    assert rep["logits_train_err_over_max"] <= max(1e-4, 2.0 * rep["stock_fp32_logits_train_err_over_max"]), rep
    assert rep["logits_train_cos"] >= 1 - (1e-8 if rep["logits_train_err_over_max"] <= 1e-4 else 1e-6), rep
    gt = torch.from_numpy(seeded_array("gt/" + name, (b, 3 * j), scale=0.2)).to(dev)
    loss = SmoothL1JointLocationLoss(num_joints=j)(logits, gt, torch.ones(b, 3 * j, device=dev))
    rep["loss"], rep["loss_ref"] = float(loss.item()), float(g[name + "/loss"])
    rep["loss_rel"] = abs(rep["loss"] - rep["loss_ref"]) / abs(rep["loss_ref"])
    sloss = o_net.joint_location_loss(slogits.float(), gt, torch.ones(b, 3 * j, device=dev), j, "smoothl1")
    rep["stock_fp32_loss_rel"] = abs(float(sloss.item()) - rep["loss_ref"]) / abs(rep["loss_ref"])
    assert rep["loss_rel"] <= max(1e-5, 2.0 * rep["stock_fp32_loss_rel"]), rep
    loss.backward()
    sloss.backward()
    rep["bn1.running_mean_rel"] = rel(sd["bn1.running_mean"].cpu(), torch.from_numpy(g[name + "/bn1.running_mean"]))
    rep["deconv7.running_var_rel"] = rel(sd["deconv_layers.7.running_var"].cpu(), torch.from_numpy(g[name + "/deconv_layers.7.running_var"]))
    assert rep["bn1.running_mean_rel"] <= 2e-6 and rep["deconv7.running_var_rel"] <= 2e-6, rep
    for k in sorted(kk[len(name) + 6:] for kk in g if kk.startswith(name + "/grad/")):
        refg = torch.from_numpy(g[name + "/grad/" + k])
        got = sd[k].grad.float().contiguous().cpu().reshape(-1)
        got = got[:: grad_stride(k, got.numel())]
        rep["grad_cos/" + k] = cosine(got, refg)
        rep["grad_norm_ratio/" + k] = float(got.double().norm() / refg.double().norm())
        sg = sd32[k].grad.float().contiguous().cpu().reshape(-1)
        sg = sg[:: grad_stride(k, sg.numel())]
        rep["stock_fp32_grad_cos/" + k] = cosine(sg, refg)
        rep["stock_fp32_grad_norm_ratio/" + k] = float(sg.double().norm() / refg.double().norm())
    REPORT["network/" + name] = rep
    # ResNet-152 at batch 2 (configuration 5): fp32 itself only reaches cosine 0.987 .. 0.994 against the reference there (stock fp32 kernels:
    # the stock_fp32_* entries) and any change of summation order moves a result by ~1e-2 (atomics 0.989, ordered sums 0.982 on the same tensors);
    # the 42 tensors stored since round 4 include the worst-conditioned BatchNorm parameters (norms: bn1.weight 3.7 % here, stock 0.4 %)
    cos_slack, norm_tol = (1.2e-2, 5e-2) if layers >= 152 else (1e-3, 5e-3)
    for k, v in rep.items():
        if k.startswith("grad_cos/"):
            assert v >= min(0.999, rep["stock_fp32_" + k] - cos_slack), (k, v, rep["stock_fp32_" + k])
        if k.startswith("grad_norm_ratio/"):        # (never worse than twice what the stock fp32 kernels reach on the same tensor, plus a small margin)
            assert abs(v - 1.0) <= max(norm_tol, 4.0 * abs(rep["stock_fp32_" + k] - 1.0) + 3e-3), (k, v, rep["stock_fp32_" + k])


# ---- 20 optimisation steps against the live reference's trajectory ------------------------------------------------------------------
@pytest.mark.parametrize("case", TRAJECTORY_CASES, ids=[c[0] for c in TRAJECTORY_CASES])
def test_training_trajectory_vs_reference(golden, case):
    """The reference's model + criterion + torch.optim.Adam stepped 20 times on one fixed batch (tests/golden/make_golden.py
    gen_trajectory) against (a) the fp32-grade mode with torch.optim.Adam and (b) the bf16 TRAINING PATH: train_step with FusedAdam,
    C++ autograd nodes, grouped weight gradients on the second stream -- the product, end to end."""
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.function import train_step
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models import precise
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd.optim import FusedAdam
    name, layers, image, j, d, b, steps, lr = case
    g = golden("trajectory")
    ref = g[name + "/losses"]
    dev = torch.device("cuda:0")
    x = torch.from_numpy(seeded_array("img/traj/" + name, (b, 3, image, image))).to(dev)
    gt = torch.from_numpy(seeded_array("gt/traj/" + name, (b, 3 * j), scale=0.2)).to(dev)
    wt = torch.ones(b, 3 * j, device=dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, d, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    model = get_pose_net(cfg, is_train=True).to(dev)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    # (a) fp32-grade mode
    sd = gpu_state(shapes, 2, TRAJECTORY_HEAD_STD, dev)
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=lr)
    got = []
    for _ in range(steps):
        opt.zero_grad()
        loss = crit(precise.forward(sd, x, layers, training=True), gt, wt)
        loss.backward()
        opt.step()
        got.append(float(loss.item()))
    with torch.no_grad():
        got.append(float(crit(precise.forward(sd, x, layers, training=True), gt, wt).item()))
    got = np.asarray(got)
    rep = {"ref": ref.tolist(), "precise": got.tolist(), "precise_rel": (np.abs(got - ref) / ref).tolist()}
    # (b) the bf16 training path
    model.load_state_dict(fill_state_dict(shapes, seed=2, head_std=TRAJECTORY_HEAD_STD))
    model.train()
    fopt = FusedAdam(model, lr=lr)
    bf = [float(train_step(model, crit, fopt, x, gt, wt)) for _ in range(steps)]
    model.train()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        bf.append(float(crit(model(x), gt, wt).item()))
    bf = np.asarray(bf)
    rep["bf16"] = bf.tolist()
    rep["bf16_rel"] = (np.abs(bf - ref) / ref).tolist()
    REPORT["trajectory/" + name] = rep
    # The first step is one forward / backward on identical weights.  Afterwards Adam (whose update is ~lr * sign(g) wherever |g| is
    # small) amplifies rounding differences step by step, and the reference's own trajectory has chaotic spikes (cfg1: steps 6 and 16,
    # r50: step 5 -- a loss that jumps UP by 10 %): around a spike any two fp32 implementations differ by O(10 %), before and after
    # they agree again.  Measured on MI355X: fp32-grade mode 1e-7 / 6e-5 / 5e-4 on steps 0 / 1 / 2, median 2e-3 .. 6e-3 over the 21
    # losses; bf16 training path 1e-5 / 4e-4 / 3e-3, median 2e-3 .. 1.5e-2.
    assert rep["precise_rel"][0] <= 2e-5 and max(rep["precise_rel"][:3]) <= 2e-3, rep["precise_rel"]
    assert float(np.median(rep["precise_rel"])) <= 1.5e-2 and max(rep["precise_rel"]) <= 0.25, rep["precise_rel"]
    assert rep["bf16_rel"][0] <= 1e-3 and max(rep["bf16_rel"][:3]) <= 1.5e-2, rep["bf16_rel"]
    assert float(np.median(rep["bf16_rel"])) <= 8e-2 and max(rep["bf16_rel"]) <= 0.3, rep["bf16_rel"]        # (run to run: median 0.002 .. 0.045)
    assert bf[-1] < 0.9 * bf[0] and got[-1] < 0.9 * got[0]


# ---- the bf16 training path against the fp32-grade mode at a trained state -----------------------------------------------------------
# (layers, image, J, D, batch) -> floors (min, 5th percentile, median cosine over all parameter gradients; loss rtol).  "bench" IS the bench
# configuration (BASELINE.json configs[1]: ResNet-50, 256 x 256, batch 32, D = 64); "small" is the round-3 shape.
# Measured on MI355X in deterministic mode (the same figures in every run of one build); ranges over the builds of round 4 (see tests/test_hip_network.py),
# (min, 5th percentile, median) over all 161 tensors | (min, 5th percentile) over the weight tensors:
#   small (ResNet-50, 128 x 128, batch 8, 10 steps)           0.79-0.90 / 0.90-0.943 / 0.977-0.993 | 0.907-0.919 / 0.916-0.924      losses equal to 4e-6
#   bench (ResNet-50, 256 x 256, batch 32, D = 64, 30 steps)  0.784-0.869 / 0.859-0.899 / 0.975-0.990 | 0.873-0.904 / 0.878-0.906   losses equal to 5e-5
# At batch 32 the network leaves its ill-conditioned initial state more slowly (loss 0.98 -> 0.91 in 10 steps: cosines 0.75 / 0.81 / 0.95 then), so the
# bench case trains 30 steps first; its minima are the stem convolution and layer-1 BatchNorm biases, the head is at 0.9998.  (This case is what found the
# fp32-result planning defect of rounds 2-3: csrc/head_gemm.hip gemm_plan.)  Floors = lowest seen - 0.05; (min, p05, median, loss rtol, min weights, p05 weights).
TRAINED_CASES = {
    "small_r50_128_b8": ((50, 128, 17, 32, 8, 10), (0.72, 0.84, 0.92, 1e-3, 0.85, 0.86)),
    "bench_r50_256_b32": ((50, 256, 17, 64, 32, 30), (0.72, 0.80, 0.92, 1e-3, 0.82, 0.82)),
}


@pytest.mark.parametrize("case", sorted(TRAINED_CASES))
def test_bf16_training_path_vs_precise_at_trained_state(case):
    """every parameter gradient of the bf16 product path against the fp32-grade mode after 10 Adam
    steps (tests/trained_state.py) -- no stock-bf16 escape clause -- at the round-3 shape AND at the bench configuration itself."""
    from trained_state import bf16_vs_precise_at_trained_state
    (layers, image, j, d, b, steps), (min_floor, p05_floor, med_floor, loss_rtol, wmin_floor, wp05_floor) = TRAINED_CASES[case]
    rep = bf16_vs_precise_at_trained_state(layers, image, j, d, b, steps=steps)
    REPORT["bf16_vs_precise_trained/" + case] = rep
    assert rep["n_params"] >= 150, rep["n_params"]
    assert abs(rep["loss_bf16"] - rep["loss_precise"]) <= loss_rtol * rep["loss_precise"], rep
    assert rep["head_min_cos"] >= 0.995, rep["head_min_cos"]
    # (round 3 ran this with atomics and a state that varied run to run: minimum 0.78 .. 0.87, 5th percentile 0.88 .. 0.90 -- the figures
    #  tools/probe_conditioning.py gets for stock bf16 autocast on the reference network)
    assert rep["min_cos"] >= min_floor, rep["worst"]
    assert rep["p05_cos"] >= p05_floor and rep["median_cos"] >= med_floor, (rep["p05_cos"], rep["median_cos"])
    assert rep["min_cos_weights"] >= wmin_floor and rep["p05_cos_weights"] >= wp05_floor, rep["worst_weights"]
