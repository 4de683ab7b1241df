"""FusedAdam.enable_step_in_backward: the update of the deep part of a network issued from a tensor hook INSIDE the backward pass, on
the second HIP stream of the C++ glue, must leave exactly the parameters a plain ``step()`` after the pass leaves."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
REPORT = {}


@pytest.fixture(autouse=True)
def deterministic_sums():
    """Every test of this file runs in the library's deterministic mode (ordered BatchNorm sums, hip.set_deterministic): two runs of the same code are then
    bit-identical, so the comparisons below need no run-to-run yardstick (rounds 2-5 judged them against the noise of the fp32 atomics)."""
    import json
    import os
    from epipolarpose_amd import hip
    was = hip.set_deterministic(True)
    yield
    hip.set_deterministic(was)
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "step_in_backward.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)


def _train(model, data, steps, early, boundary=None, late=None):
    from epipolarpose_amd.optim import FusedAdam
    opt = FusedAdam(model, lr=1e-2)
    if early:
        opt.enable_step_in_backward(boundary(model), late(model))
    fired = []
    for s in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = model(data[s]).float().square().mean()
        loss.backward()
        fired.append(opt._early_done)
        opt.step()
        assert not opt._early_done
    torch.cuda.synchronize()
    return opt, fired


def _conditioned_pose_net(seed, size=128, batch=16, steps=8, warm_up=None):
    """A small PoseResNet-18 in a WELL-CONDITIONED state, and its data.  At random initialisation with a handful of samples per BatchNorm
    channel in layer4 the network is chaotic: one bf16 rounding flip in the stem (the order of the statistics' atomics decides it) moves
    the output by 2.6 % and deep-layer gradients by 50 %, so two runs of the SAME code end up 0.3 of a four-step update apart and
    whether two runs agree is a coin toss (tools/debug_bucket_flake.py, tools/debug_fwd_bimodal.py; the run-to-run spread of the
    four-step trajectory is 0.06-0.45 of the update there against 0.02-0.04 from the state made here).  128 x 128 images, 16 of them, and eight
    plain steps first."""
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    dev = torch.device("cuda:0")
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.EXTRA.NUM_LAYERS = 18
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = 4, 16, [size, size]
    torch.manual_seed(seed)
    base = get_pose_net(cfg, is_train=False).to(dev).train()
    data = [torch.randn(batch, 3, size, size, device=dev) for _ in range(4)]
    warm = copy.deepcopy(base)
    if warm_up is not None:
        warm_up(warm, [data[i % 4] for i in range(steps)])
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            _train(warm, [data[i % 4] for i in range(steps)], steps, early=False)
    base.load_state_dict(warm.state_dict())          # (the copy keeps the bf16 training copies; ``base`` stays a plain module)
    return base, data


def test_step_in_backward_is_bit_identical_on_a_deterministic_network():
    """Linear layers only (deterministic GEMMs, no atomics): the early update of layers 2 and 3 from the hook on layer 1's output +
    the late update of layer 1 in step() == one update of everything in step(), bit for bit, moments and step count included."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                              torch.nn.Linear(256, 32)).to(dev)
    alt = copy.deepcopy(ref)
    data = [torch.randn(128, 64, device=dev) for _ in range(4)]
    o_ref, fired_ref = _train(ref, data, 4, early=False)
    o_alt, fired_alt = _train(alt, data, 4, early=True, boundary=lambda m: m[0], late=lambda m: [m[0]])
    assert fired_ref == [False] * 4 and fired_alt == [True] * 4
    for (k, a), (_, b) in zip(ref.state_dict().items(), alt.state_dict().items()):
        assert torch.equal(a, b), k
    sa, sb = o_ref.state_dict()["state"], o_alt.state_dict()["state"]
    for i in sa:
        assert float(sa[i]["step"]) == float(sb[i]["step"]) == 4.0
        assert torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and torch.equal(sa[i]["exp_avg_sq"], sb[i]["exp_avg_sq"])


def test_step_in_backward_on_the_pose_network():
    """PoseResNet-18 on the hand-written kernels (bf16 training copies, packed backward operands, deferred slab sums, weight gradients
    on the second stream): three steps with the update of layers 2-4 + head inside the backward pass against three plain steps.
    Deterministic mode: two plain runs are bit-identical, and the early update is the same arithmetic on the same gradients -- bit-identical too."""
    base, data = _conditioned_pose_net(seed=1)

    def run(early):
        m = copy.deepcopy(base)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            opt, fired = _train(m, data, 3, early, boundary=lambda mm: mm.step_in_backward_split()[0],
                                late=lambda mm: mm.step_in_backward_split()[1])
        assert fired == [early] * 3
        # the packed backward operands follow the bf16 training copies in both modes
        for mod in m.modules():
            if getattr(mod, "weight_bwd", None) is not None and mod.weight.shape[2] == 1 and getattr(mod, "epi_geometry", (0, 0, 0))[1] == 1:
                assert torch.equal(mod.weight_bwd.view(mod.weight.shape[1], mod.weight.shape[0]), mod.weight_lp.flatten(1).t())
        return {k: v.detach().float().clone() for k, v in m.named_parameters()}, m
    a, _ = run(False)
    a2, _ = run(False)
    b, mb = run(True)
    _, late_modules = base.step_in_backward_split()
    late_names = {n for n, p_ in base.named_parameters() if any(p_ is q for m_ in late_modules for q in m_.parameters())}
    start = {k: v.detach().float() for k, v in base.named_parameters()}
    assert all(torch.equal(a[k], a2[k]) for k in a)                      # the plain path repeats itself
    for part in ("late", "early"):
        keys = [k for k in a if (k in late_names) == (part == "late")]
        assert keys
        diff = sum(float((a[k] - b[k]).abs().sum()) for k in keys)
        step = sum(float((a[k] - start[k]).abs().sum()) for k in keys)
        REPORT["pose_network/" + part] = {"diff": diff, "step": step, "not_identical": [k for k in keys if not torch.equal(a[k], b[k])][:8]}
        # a part that was NOT updated, or updated twice, differs by the whole step in (almost) every element
        assert step > 0 and diff == 0.0, (part, diff, step, REPORT["pose_network/" + part]["not_identical"])
    for k in a:
        assert float((b[k] - start[k]).abs().max()) > 0 or float((a[k] - start[k]).abs().max()) == 0, k
    # the bf16 training copies follow their masters
    for mod in mb.modules():
        if getattr(mod, "weight_lp", None) is not None:
            assert torch.equal(mod.weight_lp.detach(), mod.weight.detach().to(torch.bfloat16)), type(mod)


# Deterministic mode: the figure is the same in every run of one build (measured 0.0094 of the four-step update, call r06a).  It is not zero because the
# bucket path keeps the gradients of the bf16 training copies in a bf16 bucket (one more rounding before Adam, which moves an element by ~lr whatever the
# size of its gradient); a bucket that left before its gradient was final, or a part updated twice, is the whole step.
BUCKETED_BAR = 0.03


def test_bucketed_grad_sync_with_second_stream_and_deferred_sums():
    """The N > 1 gradient path on one GPU (world size 1: buckets, hooks, flat gradient views -- no collective): from the second step on
    only the last gradient of each bucket keeps its hook, every other weight gradient runs on the second stream with deferred slab sums,
    and the hook must see finished gradients (it joins the stream and sums the slabs first).  Four steps with the bucket path against
    four plain steps in deterministic mode: the plain path repeats itself bit for bit; the bucket path sums the same weight-gradient slabs at another moment
    (at the bucket's hook instead of after the pass) and holds part of the gradients in bf16 (BUCKETED_BAR)."""
    from epipolarpose_amd.core.function import train_step
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.distributed import BucketedGradSync
    from epipolarpose_amd.optim import FusedAdam
    dev = torch.device("cuda:0")
    j = 4
    torch.manual_seed(3)
    gt = (torch.rand(16, 3 * j, device=dev) - 0.5) * 0.4
    vis = torch.ones(16, 3 * j, device=dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)

    def warm_up(m, batches):
        opt = FusedAdam(m, lr=1e-2)
        for x in batches:
            train_step(m, crit, opt, x, gt, vis)
        torch.cuda.synchronize()
    base, data = _conditioned_pose_net(seed=2, batch=16, warm_up=warm_up)

    def run(bucketed):
        m = copy.deepcopy(base)
        opt = FusedAdam(m, lr=1e-2)
        sync = BucketedGradSync(m, optimizer=opt, bucket_bytes=4 << 20) if bucketed else None
        if bucketed:
            assert len(sync.buckets) >= 4
        for x in data:
            train_step(m, crit, opt, x, gt, vis, grad_sync=sync)
        if bucketed:
            assert not sync._learning and len(sync._hooks) == len(sync.buckets)      # learned: one hook per bucket
        torch.cuda.synchronize()
        return {k: v.detach().float().clone() for k, v in m.named_parameters()}
    a, a2, b = run(False), run(False), run(True)
    start = {k: v.detach().float() for k, v in base.named_parameters()}
    assert all(torch.equal(a[k], a2[k]) for k in a)
    diff = sum(float((a[k] - b[k]).abs().sum()) for k in a)
    step = sum(float((a[k] - start[k]).abs().sum()) for k in a)
    REPORT["bucketed"] = {"diff": diff, "step": step, "not_identical": [k for k in a if not torch.equal(a[k], b[k])][:8]}
    assert step > 0 and diff <= BUCKETED_BAR * step, (diff, step, REPORT["bucketed"]["not_identical"])
