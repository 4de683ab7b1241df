"""GPU: the backbone convolutions on the hand-written implicit-GEMM kernels (epi_conv2d_fwd / _bwd_data / _bwd_weight) against
torch's fp32 ``F.conv2d`` and its autograd on bf16-rounded inputs, at every convolution geometry of ResNet-18/50/152
(pose3d_resnet.py:21-88,130-136: 3x3 stride 1|2 pad 1, 1x1 stride 1|2) -- reduced batch / spatial size, full channel counts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (Cin, Cout, k, stride, H) -- H is the input extent; channel counts of ResNet-50's four stages + the BasicBlock shapes
GEOMETRIES = [
    (64, 64, 1, 1, 16), (64, 64, 3, 1, 16), (64, 256, 1, 1, 16), (256, 64, 1, 1, 16),          # layer1
    (256, 128, 1, 1, 16), (128, 128, 3, 2, 16), (128, 512, 1, 1, 8), (256, 512, 1, 2, 16),      # layer2.0 (+ downsample)
    (512, 128, 1, 1, 8), (128, 128, 3, 1, 8),
    (512, 256, 1, 1, 8), (256, 256, 3, 2, 8), (256, 1024, 1, 1, 4), (512, 1024, 1, 2, 8),       # layer3.0
    (1024, 256, 1, 1, 4), (256, 256, 3, 1, 4),
    (1024, 512, 1, 1, 4), (512, 512, 3, 2, 4), (512, 2048, 1, 1, 2), (1024, 2048, 1, 2, 4),     # layer4.0
    (2048, 512, 1, 1, 2), (512, 512, 3, 1, 2),
    (64, 128, 3, 2, 16), (64, 128, 1, 2, 16),                                                   # BasicBlock (ResNet-18/34) stage entries
]


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("geo", GEOMETRIES, ids=["%dto%d_k%ds%d_h%d" % g for g in GEOMETRIES])
def test_conv2d_fwd_bwd_vs_torch_fp32(geo):
    from epipolarpose_amd import hip
    cin, cout, k, stride, h = geo
    pad = k // 2
    b = 3
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin * 7 + cout + k + stride)
    x = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = _rand((cout, cin, k, k), gen, scale=(2.0 / (cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    ref = F.conv2d(xf, wf, stride=stride, padding=pad)
    y = hip.conv2d_fwd(x, w, stride, pad)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    tol = 2 ** -7 * ref.abs().max().item()                   # one bf16 rounding of the fp32-accumulated result
    assert (y.float() - ref).abs().max().item() <= tol
    dy = _rand(tuple(ref.shape), gen).to(dev).contiguous(memory_format=torch.channels_last)
    ref.backward(dy.float())
    w_bwd = hip.conv2d_pack_weight_bwd(w, stride, pad)
    dx = hip.conv2d_bwd_data(dy, w_bwd, tuple(x.shape), k, stride, pad)
    assert dx.shape == x.shape
    assert (dx.float() - xf.grad).abs().max().item() <= 2 ** -7 * xf.grad.abs().max().item() + 1e-6
    dw = hip.conv2d_bwd_weight(x, dy, k, stride, pad, dtype=torch.float32)
    assert (dw - wf.grad).abs().max().item() <= 2e-3 * wf.grad.abs().max().item() + 1e-5     # fp32 out: split-K summation order only


def test_conv2d_full_size_bench_shapes_spot_check():
    """The bench configuration's largest shapes (batch 32): a sampled comparison so that the kernels' big-tile / split-K /
    A-stationary paths -- selected by shape -- are exercised exactly as the training step selects them."""
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    for cin, cout, k, stride, h in ((256, 64, 1, 1, 64), (64, 64, 3, 1, 64), (64, 256, 1, 1, 64), (128, 128, 3, 2, 64), (512, 512, 3, 1, 8),
                                    (1024, 2048, 1, 2, 16), (256, 256, 3, 1, 16)):
        pad = k // 2
        x = _rand((32, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
        w = _rand((cout, cin, k, k), gen, scale=(2.0 / (cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
        y = hip.conv2d_fwd(x, w, stride, pad)
        sl = slice(0, 32, 13)                                 # images 0, 13, 26 in fp32
        ref = F.conv2d(x[sl].float(), w.float(), stride=stride, padding=pad)
        assert (y[sl].float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
        dy = _rand(tuple(y.shape), gen).to(dev).contiguous(memory_format=torch.channels_last)
        dx = hip.conv2d_bwd_data(dy, hip.conv2d_pack_weight_bwd(w, stride, pad), tuple(x.shape), k, stride, pad)
        refdx = torch.nn.grad.conv2d_input((3, cin, h, h), w.float(), dy[sl].float(), stride=stride, padding=pad)
        assert (dx[sl].float() - refdx).abs().max().item() <= 2 ** -7 * refdx.abs().max().item() + 1e-6
        dw = hip.conv2d_bwd_weight(x, dy, k, stride, pad, dtype=torch.float32)
        refdw = torch.nn.grad.conv2d_weight(x.float(), (cout, cin, k, k), dy.float(), stride=stride, padding=pad)
        assert (dw - refdw).abs().max().item() <= 3e-3 * refdw.abs().max().item() + 1e-5


# (Cin, Cout, B, H, W): the patch-stationary 3x3 kernel -- narrow (Cout 64) / wide tiles, unsplit (many tiles) / split over channel
# chunks (few tiles), ragged last tile (B*H*W % 256 != 0), several images per tile (small H*W), H != W, a three-deep and (W = 96)
# a two-deep weight ring, Cout that does not fill the last 128-wide tile
PATCH_CASES = [(64, 64, 32, 64, 64), (64, 64, 3, 10, 10), (128, 128, 32, 32, 32), (128, 192, 5, 12, 20), (256, 256, 7, 16, 16), (512, 512, 9, 8, 8),
               (64, 128, 2, 96, 96), (256, 64, 3, 6, 4), (128, 128, 1, 3, 3)]


@pytest.fixture
def patch_kernel_always():
    from epipolarpose_amd import hip
    before = hip.conv3x3_patch_mode(2)
    yield
    hip.conv3x3_patch_mode(before)


@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "%dto%d_b%d_%dx%d" % c)
def test_conv3x3_patch_kernel_vs_torch_fp32(case, patch_kernel_always):
    """3x3 / stride 1 / pad 1 forward and backward-data (the patch-stationary kernel: every filter tap from one staged pixel patch,
    border taps masked per row) against fp32 F.conv2d on the same bf16 operands -- every image border, image-to-image transitions
    inside one tile and the zero tail of the last tile included."""
    from epipolarpose_amd import hip
    cin, cout, b, h, w_ = case
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin + cout + b + h * w_)
    x = _rand((b, cin, h, w_), gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = _rand((cout, cin, 3, 3), gen, scale=(2.0 / (cin * 9)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    sl = slice(0, b, max(1, b // 3))                        # a few images in fp32 (first, middle, last third)
    y = hip.conv2d_fwd(x, w, 1, 1)
    ref = F.conv2d(x[sl].float(), w.float(), padding=1)
    assert (y[sl].float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    dy = _rand(tuple(y.shape), gen).to(dev).contiguous(memory_format=torch.channels_last)
    r = _rand(tuple(x.shape), gen).to(dev).contiguous(memory_format=torch.channels_last)
    wb = hip.conv2d_pack_weight_bwd(w, 1, 1)
    dx = hip.conv2d_bwd_data(dy, wb, tuple(x.shape), 3, 1, 1)
    n = len(range(*sl.indices(b)))
    refdx = torch.nn.grad.conv2d_input((n, cin, h, w_), w.float(), dy[sl].float(), padding=1)
    assert (dx[sl].float() - refdx).abs().max().item() <= 2 ** -7 * refdx.abs().max().item() + 1e-6
    fused = hip.conv2d_bwd_data(dy, wb, tuple(x.shape), 3, 1, 1, addend=r)
    assert torch.equal(fused, (dx.float() + r.float()).to(torch.bfloat16))
    # BatchNorm statistics from the epilogue (unsplit launches) are the column sums of what was written
    sums = torch.zeros(hip.bn_sum_copies(cout) * 2 * cout, dtype=torch.float32, device=dev)
    y2, done = hip.conv2d_fwd(x, w, 1, 1, bn_sums=sums)
    assert torch.equal(y2, y)
    if done:
        yf = y.float()
        got = sums.view(-1, 2 * cout).sum(0)
        torch.testing.assert_close(got[:cout], yf.sum(dim=(0, 2, 3)), rtol=2e-4, atol=2e-3 * float(yf.abs().max()) * (yf.numel() / cout) ** 0.5)
        torch.testing.assert_close(got[cout:], (yf * yf).sum(dim=(0, 2, 3)), rtol=2e-4, atol=1e-3)


def test_conv2d_unsupported_geometry_is_refused():
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    x = torch.zeros(1, 3, 16, 16, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(64, 3, 7, 7, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match="epi_conv2d_fwd"):
        hip.conv2d_fwd(x, w, 2, 3)                            # the 7x7 stem (49 taps, 3 channels) stays with the library


def _ref_conv_bn(x, w, gamma, beta, residual, stride, pad, relu, eps=1e-5):
    """fp32 reference of one conv -> BatchNorm(training) (+ residual) (+ ReLU) stage (pose3d_resnet.py:68-88)."""
    raw = F.conv2d(x, w, stride=stride, padding=pad).to(torch.bfloat16).float()      # our kernel rounds the conv output to bf16
    y = F.batch_norm(raw, None, None, gamma, beta, True, 0.1, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


@pytest.mark.parametrize("geo", [(64, 64, 3, 1, 16, False), (256, 64, 1, 1, 16, False), (128, 128, 3, 2, 16, False), (64, 256, 1, 1, 16, True),
                                 (256, 512, 1, 2, 16, False), (512, 512, 3, 1, 4, True)],
                         ids=lambda g: "%dto%d_k%ds%d_h%d_res%d" % g)
def test_conv_bn_act_node_vs_torch_fp32(geo):
    """The fused autograd node (glue conv_bn_act: conv + BatchNorm + residual + ReLU forward, and its whole backward) against
    torch fp32 autograd of the same composition; fp32-master and bf16-copy weight inputs."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.models.fused import FusedBatchNormAct
    cin, cout, k, stride, h, with_res = geo
    pad, b = k // 2, 4
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin + 3 * cout + k)
    x = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    w32 = (_rand((cout, cin, k, k), gen, scale=(2.0 / (cin * k * k)) ** 0.5).float()).to(dev).contiguous(memory_format=torch.channels_last)
    ho = (h + 2 * pad - k) // stride + 1
    res = _rand((b, cout, ho, ho), gen).to(dev).contiguous(memory_format=torch.channels_last) if with_res else None
    dy = _rand((b, cout, ho, ho), gen).to(dev).contiguous(memory_format=torch.channels_last)
    bn = FusedBatchNormAct(cout, relu=True).to(dev)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(cout, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=gen) * 0.1)
    # reference
    xr, wr = x.float().requires_grad_(True), w32.clone().requires_grad_(True)
    gr, br = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    rr = res.float().requires_grad_(True) if with_res else None
    yr = _ref_conv_bn(xr, wr, gr, br, rr, stride, pad, True)
    yr.backward(dy.float())
    for wdtype in (torch.float32, torch.bfloat16):
        bn.zero_grad()
        xo = x.clone().requires_grad_(True)
        wo = w32.detach().clone().to(wdtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ro = res.clone().requires_grad_(True) if with_res else None
        bn.train()
        y = hip.glue().conv_bn_act(xo, wo, None, stride, pad, *bn._tensors()[:2], ro, *bn._tensors()[2:], bn._flags, True, bn.momentum,
                                   bn.eps, True)
        assert y.dtype == torch.bfloat16 and y.shape == yr.shape
        assert (y.float() - yr).abs().max().item() <= 2 ** -6 * max(1.0, yr.abs().max().item())
        y.backward(dy)
        assert wo.grad.dtype == wdtype and wo.grad.shape == wo.shape
        scale = lambda t: t.abs().max().item() + 1e-6
        assert (xo.grad.float() - xr.grad).abs().max().item() <= 3e-2 * scale(xr.grad)
        assert (wo.grad.float() - wr.grad).abs().max().item() <= 3e-2 * scale(wr.grad)
        assert (bn.weight.grad - gr.grad).abs().max().item() <= 3e-2 * scale(gr.grad)
        assert (bn.bias.grad - br.grad).abs().max().item() <= 3e-2 * scale(br.grad)
        if with_res:
            assert (ro.grad.float() - rr.grad).abs().max().item() <= 2e-2 * scale(rr.grad)


def test_bn_parameter_gradients_survive_a_forward_and_accumulate():
    """BatchNorm parameter gradients must not alias the accumulator the next forward clears: with gradient
    accumulation (two forward/backward rounds before the optimizer step) weight.grad must be g1 + g2."""
    from epipolarpose_amd.models.fused import FusedBatchNormAct
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(2)
    bn = FusedBatchNormAct(64, relu=True).to(dev)
    xs = [_rand((2, 64, 8, 8), gen).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    dys = [_rand((2, 64, 8, 8), gen).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    singles = []
    for x, dy in zip(xs, dys):
        bn.zero_grad()
        bn(x).backward(dy)
        singles.append((bn.weight.grad.clone(), bn.bias.grad.clone()))
    bn.zero_grad()
    bn(xs[0]).backward(dys[0])
    g1 = bn.weight.grad.clone()
    y2 = bn(xs[1])                                   # a forward between backward and step: clears the internal accumulator
    torch.cuda.synchronize()
    torch.testing.assert_close(bn.weight.grad, g1)   # ... but not the gradient
    y2.backward(dys[1])
    torch.testing.assert_close(bn.weight.grad, singles[0][0] + singles[1][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(bn.bias.grad, singles[0][1] + singles[1][1], rtol=1e-5, atol=1e-5)


def test_fused_adam_maintains_packed_backward_weights():
    """FusedAdam keeps every fused convolution's backward-data operand (weight_bwd) equal to a fresh pack of the updated bf16
    copy, through steps and through an external load_state_dict."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd.optim import FusedAdam
    from tests_util_cfg import make_cfg
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = get_pose_net(make_cfg(18, 64, 3, 8), is_train=True).to(dev)
    opt = FusedAdam(model, lr=1e-3)
    convs = [m for m in model.modules() if getattr(m, "epi_geometry", None) is not None]
    assert len(convs) == 19 and all(getattr(m, "weight_bwd", None) is not None for m in convs)        # 16 3x3 + 3 downsample 1x1

    def check():
        for m in convs:
            k, s, p = m.epi_geometry
            assert torch.equal(m.weight_lp.detach().float(), m.weight.detach().to(torch.bfloat16).float())
            assert torch.equal(m.weight_bwd, hip.conv2d_pack_weight_bwd(m.weight_lp.detach(), s, p))
    check()
    crit = SmoothL1JointLocationLoss(num_joints=3)
    x = torch.randn(4, 3, 64, 64, device=dev)
    gt = torch.rand(4, 9, device=dev) - 0.5
    for _ in range(2):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
        crit(out, gt, torch.ones_like(gt)).backward()
        opt.step()
    check()
    model.load_state_dict({k: (torch.randn_like(v) * 0.05 if v.dtype.is_floating_point and v.dim() == 4 else v) for k, v in model.state_dict().items()})
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        model(x)                                     # the forward notices the new masters and re-syncs copy + packed operand
    check()


@pytest.mark.parametrize("geo", [(256, 64, 1, 1, 64, 8), (64, 64, 3, 1, 64, 8), (64, 256, 1, 1, 64, 8), (128, 512, 1, 1, 32, 16), (128, 128, 3, 2, 64, 8),
                                 (512, 512, 3, 1, 8, 32), (256, 256, 3, 1, 16, 32), (64, 64, 1, 1, 16, 2)],
                         ids=lambda g: "%dto%d_k%ds%d_h%d_b%d" % g)
def test_conv2d_fwd_fused_bn_statistics(geo):
    """The BatchNorm batch sums accumulated by the convolution's own epilogue (tall / small / big / A-stationary kernels) equal the
    column sums of the bf16 tensor it wrote; launches that cannot do it (split-K) say so and leave the accumulator alone."""
    from epipolarpose_amd import hip
    cin, cout, k, stride, h, b = geo
    pad = k // 2
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin + cout + k + h)
    x = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = _rand((cout, cin, k, k), gen, scale=(2.0 / (cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    sums = torch.zeros(hip.bn_sum_copies(cout) * 2 * cout, dtype=torch.float32, device=dev)
    y, done = hip.conv2d_fwd(x, w, stride, pad, bn_sums=sums)
    sums = sums.view(-1, 2 * cout).sum(0)                 # the M tiles spread their atomics over the copies
    ref = hip.conv2d_fwd(x, w, stride, pad)
    assert torch.equal(y, ref)
    if done:
        yf = y.float()
        s1, s2 = yf.sum(dim=(0, 2, 3)), (yf * yf).sum(dim=(0, 2, 3))
        torch.testing.assert_close(sums[:cout], s1, rtol=2e-4, atol=2e-3 * float(yf.abs().max()) * (yf.numel() / cout) ** 0.5)
        torch.testing.assert_close(sums[cout:], s2, rtol=2e-4, atol=1e-3)
    else:
        assert float(sums.abs().max()) == 0.0


@pytest.mark.parametrize("geo", [(64, 64, 3, 1, 16, 3), (256, 64, 1, 1, 64, 8), (64, 256, 1, 1, 64, 8), (128, 128, 3, 2, 16, 3), (256, 512, 1, 2, 16, 3),
                                 (512, 512, 3, 1, 8, 32), (2048, 512, 1, 1, 8, 32)],
                         ids=lambda g: "%dto%d_k%ds%d_h%d_b%d" % g)
def test_conv2d_bwd_data_epilogue_addend(geo):
    """dx = bf16(backward-data) + addend in the GEMM epilogue (every kernel path: tall / small / A-stationary / split-K / phased)."""
    from epipolarpose_amd import hip
    cin, cout, k, stride, h, b = geo
    pad = k // 2
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin + 5 * cout + k)
    w = _rand((cout, cin, k, k), gen, scale=(2.0 / (cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    ho = (h + 2 * pad - k) // stride + 1
    dy = _rand((b, cout, ho, ho), gen).to(dev).contiguous(memory_format=torch.channels_last)
    r = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    wb = hip.conv2d_pack_weight_bwd(w, stride, pad)
    plain = hip.conv2d_bwd_data(dy, wb, (b, cin, h, h), k, stride, pad)
    fused = hip.conv2d_bwd_data(dy, wb, (b, cin, h, h), k, stride, pad, addend=r)
    assert torch.equal(fused, (plain.float() + r.float()).to(torch.bfloat16))


_UNIT_KINDS = {"bottleneck_proj": ("_BOTTLENECK", 64, 64, 1), "bottleneck_identity": ("_BOTTLENECK", 256, 64, 1),
               "bottleneck_stride2": ("_BOTTLENECK", 256, 128, 2), "basic_identity": ("_BASIC", 64, 64, 1), "basic_stride2": ("_BASIC", 64, 128, 2)}


def _torch_unit(unit, plan, inplanes, planes, stride):
    """Plain fp32 torch restatement of one residual unit (pose3d_resnet.py:31-47,68-88,130-136) on CPU, holding `unit`'s parameters and running
    estimates: the oracle of the unit tests below (nn.Conv2d / nn.BatchNorm2d and autograd, nothing of this package)."""
    import torch.nn as nn

    class Ref(nn.Module):
        def __init__(self):
            super().__init__()
            cin = inplanes
            for i, (k, mult, strided) in enumerate(plan, start=1):
                cout = planes * mult
                setattr(self, "conv%d" % i, nn.Conv2d(cin, cout, k, stride if strided else 1, k // 2, bias=False))
                setattr(self, "bn%d" % i, nn.BatchNorm2d(cout, momentum=0.1))
                cin = cout
            self.downsample = None
            if stride != 1 or inplanes != cin:
                self.downsample = nn.Sequential(nn.Conv2d(inplanes, cin, 1, stride, bias=False), nn.BatchNorm2d(cin, momentum=0.1))

        def forward(self, x):
            out = x
            for i in range(1, len(plan) + 1):
                out = getattr(self, "bn%d" % i)(getattr(self, "conv%d" % i)(out))
                if i < len(plan):
                    out = torch.relu(out)
            return torch.relu(out + (x if self.downsample is None else self.downsample(x)))

    ref = Ref()
    own = {k: v.detach().float().cpu() for k, v in unit.state_dict().items()}
    missing = ref.load_state_dict(own, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return ref.train()


def _run_unit(m, x, fp32=False):
    """One forward + backward of a residual unit (training mode) from a fixed input and a fixed output gradient; the unit sits behind a fixed
    elementwise scale so that it has to return an input gradient.  -> {"y", "dx", parameter gradients, running estimates after the step}."""
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.zero_grad(set_to_none=True)
    xin = x.clone().requires_grad_(True)
    h = xin * 0.5                                              # (exact in bf16: gives the unit's input a grad_fn)
    if not fp32:
        h = h.contiguous(memory_format=torch.channels_last)
    y = m(h)
    dy = _rand(tuple(y.shape), torch.Generator().manual_seed(9)).to(y.device)
    if fp32:
        dy = dy.float()
    else:
        dy = dy.contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    out = {"y": y.detach().clone(), "dx": xin.grad.detach().clone()}
    out.update({"grad:" + k: p.grad.detach().clone() for k, p in m.named_parameters()})
    out.update({"stat:" + k: v.detach().clone() for k, v in m.state_dict().items() if "running" in k})
    m.load_state_dict(state)                                   # running estimates back: every run starts from the same state
    return out


def _make_unit_pair(kind, batch=4, size=16):
    import copy
    from epipolarpose_amd.models import pose3d_resnet as P
    dev = torch.device("cuda:0")
    plan_name, inpl, planes, stride = _UNIT_KINDS[kind]
    plan = getattr(P, plan_name)
    torch.manual_seed(3)
    unit = P.ResidualUnit(inpl, planes, plan, stride)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, p in unit.named_parameters():
            if p.dim() == 4:
                p.copy_(p.to(torch.bfloat16).float())          # bf16-representable weights: both sides multiply the same numbers
            elif k.endswith("weight"):
                p.copy_(0.5 + torch.rand(p.shape, generator=gen))                  # BatchNorm gamma / beta away from (1, 0)
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=gen))
    unit = unit.to(dev).to(memory_format=torch.channels_last)
    assert unit._unit is not None
    staged = P.ResidualUnit(inpl, planes, plan, stride, unit_node=False).to(dev).to(memory_format=torch.channels_last)
    assert staged._unit is None and staged._fused
    staged.load_state_dict(copy.deepcopy(unit.state_dict()))
    x = _rand((batch, inpl, size, size), torch.Generator().manual_seed(4)).to(dev).contiguous(memory_format=torch.channels_last)
    return unit, staged, _torch_unit(unit, plan, inpl, planes, stride), x


# rel-L2 of the bf16 unit against the fp32 torch unit, per tensor class (measured on MI355X in deterministic mode -- the same figures in every run --
# tools/probe_unit_node.py -> profiles/r06_unit_node_probe.txt): a dropped shortcut gradient, a missing stage or a wrong statistic is O(1).
#   output 0.003-0.005 | running estimates <= 0.002 | input gradient 0.044-0.065 | convolution weight gradients 0.042-0.072 |
#   BatchNorm parameter gradients 0.020-0.094 (sums over 256 .. 1024 positions of terms of either sign that bf16 activations round one by one)
def _unit_oracle_bar(key):
    if key == "y":
        return 0.02
    if key.startswith("stat:"):
        return 0.01
    return 0.15 if ("bn" in key or "downsample.1" in key) else 0.12


@pytest.mark.parametrize("kind", list(_UNIT_KINDS))
def test_residual_unit_node_matches_per_stage_nodes(kind, request):
    """One autograd node per residual unit (shortcut gradient added in the first stage's dgrad epilogue) against one node per conv/bn stage with
    autograd's own accumulation, in the library's deterministic mode (ordered BatchNorm sums: no run-to-run noise): the same kernels and the same
    roundings, so the two forms are BIT-IDENTICAL -- output, input gradient, every parameter gradient, the running estimates -- and each form agrees
    with the fp32 torch restatement of the unit on the same weights to the bf16 yardstick."""
    from epipolarpose_amd import hip
    # (the per-stage chain rounds the projection's BatchNorm output to bf16 before the add; the unit node's one-pass form of the two
    #  BatchNorms does not: "same roundings" holds for the two-pass form, the one-pass form has its own test below)
    prev_dual = hip.glue().bn_dual_mode(0)
    request.addfinalizer(lambda: hip.glue().bn_dual_mode(prev_dual))
    was = hip.set_deterministic(True)
    request.addfinalizer(lambda: hip.set_deterministic(was))
    unit, staged, ref, x = _make_unit_pair(kind)
    a, b = _run_unit(unit, x), _run_unit(staged, x)
    r = _run_unit(ref, x.float().cpu(), fp32=True)
    assert a.keys() == b.keys() == r.keys()
    for k in a:
        # bit for bit; the fp32 tensors (BatchNorm parameter gradients, running estimates) may differ in their last bits where the unit node's
        # reduction adds the same partial sums in another grouping (stride-2 bottleneck: one tensor, 1.6e-7 relative)
        same = torch.equal(a[k], b[k]) or (a[k].dtype == torch.float32
                                           and float((a[k] - b[k]).abs().max()) <= 1e-6 * float(a[k].abs().max()))
        assert same, (k, a[k].dtype, float((a[k].float() - b[k].float()).abs().max()))
    for k in a:                                                # the oracle: both forms against fp32 torch
        for got in (a[k].float().cpu(), b[k].float().cpu()):
            want = r[k]
            assert float((got - want).norm()) <= _unit_oracle_bar(k) * max(float(want.norm()), 1e-6), (k, float((got - want).norm()), float(want.norm()))
    a2 = _run_unit(unit, x)                                    # and the unit node repeats itself bit for bit
    for k in a:
        assert torch.equal(a[k], a2[k]), k


@pytest.mark.parametrize("kind", ["bottleneck_proj", "bottleneck_stride2", "basic_stride2"])
@pytest.mark.parametrize("training", [True, False])
def test_projection_batchnorm_in_one_pass(kind, training):
    """epi_bn_act_fwd_dual: y = relu(bn3(z3) + bn_proj(z_proj)) in one pass over the two raw convolution outputs against the two passes
    (projection BatchNorm written in bf16, then bn3 + residual + ReLU): the output differs by at most the bf16 rounding of the shortcut
    that the one-pass form no longer performs, statistics / running estimates are identical, gradients agree to the noise of that rounding."""
    import copy
    from epipolarpose_amd import hip
    from epipolarpose_amd.models import pose3d_resnet as P
    dev = torch.device("cuda:0")
    plan, inpl, planes, stride = {"bottleneck_proj": (P._BOTTLENECK, 64, 64, 1), "bottleneck_stride2": (P._BOTTLENECK, 256, 128, 2),
                                  "basic_stride2": (P._BASIC, 64, 128, 2)}[kind]
    torch.manual_seed(13)
    base = P.ResidualUnit(inpl, planes, plan, stride).to(dev).to(memory_format=torch.channels_last)
    assert base._unit is not None
    for mod in base.modules():                               # non-trivial running statistics for the eval case
        if hasattr(mod, "running_mean"):
            mod.running_mean.normal_(0, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
    gen = torch.Generator().manual_seed(14)
    x = _rand((8, inpl, 16, 16), gen).to(dev).contiguous(memory_format=torch.channels_last)
    outs = []
    for dual in (1, 0):
        prev = hip.glue().bn_dual_mode(dual)
        try:
            m = copy.deepcopy(base)
            m.train(training)
            xin = x.clone().requires_grad_(training)
            y = m(xin)
            grads = {}
            if training:
                dy = _rand(tuple(y.shape), torch.Generator().manual_seed(15)).to(dev).contiguous(memory_format=torch.channels_last)
                y.backward(dy)
                grads = {k: p.grad.float().clone() for k, p in m.named_parameters()}
                grads["x"] = xin.grad.float().clone()
            outs.append((y.detach().float().clone(), grads, {k: v.detach().float().clone() for k, v in m.state_dict().items()}))
        finally:
            hip.glue().bn_dual_mode(prev)
    (ya, ga, sa), (yb, gb, sb) = outs
    # the shortcut term is at most a few units large: its bf16 rounding moves y by <= 2^-8 of it, and y's own rounding by one more bf16 unit
    tol = 2 ** -7 * float(yb.abs().max())
    assert float((ya - yb).abs().max()) <= tol, (float((ya - yb).abs().max()), tol)
    assert float(((ya - yb).abs() > 0).float().mean()) > 0.01 or not training        # the two forms ARE different roundings (not the same code path twice)
    for k in sa:                                              # statistics come from the raw convolution outputs: identical up to atomics order
        assert float((sa[k] - sb[k]).abs().max()) <= 1e-5 * max(1.0, float(sb[k].abs().max())), k
    for k in ga:
        a, b = ga[k].reshape(-1).double(), gb[k].reshape(-1).double()
        assert float((a - b).norm()) <= 0.12 * float(b.norm()) + 1e-12, (k, float((a - b).norm()), float(b.norm()))


@pytest.mark.parametrize("kind", ["bottleneck", "bottleneck_proj", "bottleneck_stride2", "bottleneck_wide"])
@pytest.mark.parametrize("group", [2, 0])
def test_bottleneck_interior_batchnorm_inside_the_consuming_convolution(kind, group):
    """Round 4 (epi_conv1x1_fwd_bn_in + EpiWgradItem::x_scale_shift): conv3 of a Bottleneck normalises its own input -- relu(bn2(z2)) is never written,
    the forward kernel applies it to its A fragments and the weight-gradient kernel re-applies it to its B fragments -- against the separate apply
    pass of rounds 1-3.  The arithmetic is the apply kernel's on the same values, so the output and every gradient agree to the run-to-run level of
    the statistics' atomics; the saved statistics / running estimates of bn2 are written by the convolution launch and must be the same numbers.
    group 2 / 0: the weight gradient through the grouped launch / through its own launch (epi_wgrad_item)."""
    import copy
    from epipolarpose_amd import hip
    from epipolarpose_amd.models import pose3d_resnet as P
    dev = torch.device("cuda:0")
    inpl, planes, stride, hw, b = {"bottleneck": (256, 64, 1, 16, 8), "bottleneck_proj": (64, 64, 1, 16, 8), "bottleneck_stride2": (256, 128, 2, 16, 8),
                                   "bottleneck_wide": (2048, 512, 1, 8, 4)}[kind]
    torch.manual_seed(21)
    base = P.ResidualUnit(inpl, planes, P._BOTTLENECK, stride).to(dev).to(memory_format=torch.channels_last)
    assert base._unit is not None
    for mod in base.modules():
        if hasattr(mod, "running_mean"):
            mod.running_mean.normal_(0, 0.2)
            mod.running_var.uniform_(0.5, 1.5)
            mod.weight.data.normal_(1.0, 0.3)               # (a few negative scales as well: the mask test must follow the sign)
            mod.bias.data.normal_(0.0, 0.3)
    x = _rand((b, inpl, hw, hw), torch.Generator().manual_seed(22)).to(dev).contiguous(memory_format=torch.channels_last)
    glue = hip.glue()
    outs = []
    group_before = glue.wgrad_group_mode(group)
    try:
        for fuse in (1, 0, 0):
            prev = glue.bn_in_fuse_mode(fuse)
            glue.bn_in_fuse_count(True)
            try:
                m = copy.deepcopy(base)
                m.train()
                xin = x.clone().requires_grad_(True)
                y = m(xin)
                dy = _rand(tuple(y.shape), torch.Generator().manual_seed(23)).to(dev).contiguous(memory_format=torch.channels_last)
                y.backward(dy)
                torch.cuda.synchronize()
                grads = {k: p.grad.float().clone() for k, p in m.named_parameters()}
                grads["x"] = xin.grad.float().clone()
                outs.append((y.detach().float().clone(), grads, {k: v.detach().float().clone() for k, v in m.state_dict().items()}, glue.bn_in_fuse_count(True)))
            finally:
                glue.bn_in_fuse_mode(prev)
    finally:
        glue.wgrad_group_mode(group_before)
    (ya, ga, sa, na), (yb, gb, sb, nb), (yc, gc, sc, nc) = outs
    assert na == 1 and nb == 0 and nc == 0, (na, nb, nc)                 # conv3 took its input raw exactly once per forward
    noise_y = float((yb - yc).abs().max())                               # two runs of the SAME (unfused) path: the level of the atomics
    assert float((ya - yb).abs().max()) <= max(4.0 * noise_y, 2 ** -7 * float(yb.abs().max())), (float((ya - yb).abs().max()), noise_y)
    for k in sa:
        assert float((sa[k] - sb[k]).abs().max()) <= 1e-5 * max(1.0, float(sb[k].abs().max())), k
    for k in ga:
        a, bb, c = ga[k].reshape(-1).double(), gb[k].reshape(-1).double(), gc[k].reshape(-1).double()
        noise = float((bb - c).norm())
        assert float((a - bb).norm()) <= max(4.0 * noise, 2e-2 * float(bb.norm())) + 1e-12, (k, float((a - bb).norm()), noise, float(bb.norm()))


def test_deferred_weight_gradient_reduction_matches_per_layer_reduction():
    """Split weight gradients summed by one launch at the end of backward() (the autograd engine's final callback) against one
    reduce per layer: the same slabs, fp32 sums in a different order; also when backward() runs twice in a row, when a gradient is
    consumed from a hook in the middle of the pass (flush_pending_reduces, the bucketed all-reduce path) and for a bare call
    outside any backward pass."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.models import pose3d_resnet as P
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = torch.nn.Sequential(P.ResidualUnit(64, 64, P._BOTTLENECK, 1), P.ResidualUnit(256, 128, P._BOTTLENECK, 2),
                              P.ResidualUnit(512, 128, P._BOTTLENECK, 1)).to(dev).to(memory_format=torch.channels_last)
    x = _rand((8, 64, 32, 32), torch.Generator().manual_seed(6)).to(dev).contiguous(memory_format=torch.channels_last)
    pre = torch.nn.Conv2d(64, 64, 1, bias=False).to(dev).to(torch.bfloat16)

    def grads(hooked=False):
        net.zero_grad()
        seen = {}
        handles = []
        if hooked:                                       # read one gradient in the middle of backward, as a bucket hook would
            name, prm = list(net.named_parameters())[-4]
            def hook(p_):
                hip.glue().flush_pending_reduces()
                seen[name] = p_.grad.detach().float().clone()
            handles.append(prm.register_post_accumulate_grad_hook(hook))
        y = net(pre(x).contiguous(memory_format=torch.channels_last))
        y.float().square().mean().backward()
        for h in handles:
            h.remove()
        out = {k: p_.grad.detach().float().clone() for k, p_ in net.named_parameters()}
        for k, v in seen.items():
            assert torch.equal(v, out[k]), k             # what the hook saw is the finished gradient
        return out

    before = hip.glue().defer_wgrad_reduce(False)
    stream_before = hip.glue().wgrad_stream_mode(0)
    group_before = hip.glue().wgrad_group_mode(0)
    try:
        ref = grads()
        # (deferred reduce, weight gradients on the second stream [0 off / 1 on / 2 on, lowest priority], gradient read from a hook,
        #  grouped weight-gradient launches [0 per layer / 1 per unit / 2 per stage])
        for defer, side, hooked, group in ((True, 0, False, 0), (True, 0, False, 0), (True, 0, True, 0), (False, 1, False, 0), (True, 1, False, 0),
                                           (True, 1, True, 0), (True, 2, False, 0), (True, 1, False, 0),
                                           (True, 0, False, 2), (True, 0, True, 2), (True, 1, False, 2), (True, 1, False, 2), (True, 1, True, 2),
                                           (True, 1, False, 1), (True, 1, True, 1), (False, 1, False, 2), (True, 2, False, 2)):
            hip.glue().defer_wgrad_reduce(defer)
            hip.glue().wgrad_stream_mode(side)
            hip.glue().wgrad_group_mode(group)
            got = grads(hooked)
            for k in ref:
                scale = float(ref[k].abs().max()) + 1e-12
                # run-to-run noise of this network (BatchNorm sums by atomics, bf16 activations: see the unit-node test) is ~5 % of the
                # largest element / of the norm; a reduce that did not run, or a gradient read before its launch finished, leaves O(1) garbage
                assert float((got[k] - ref[k]).abs().max()) <= 2e-1 * scale, (k, defer, side, hooked, group)
                assert float((got[k] - ref[k]).norm()) <= 1e-1 * float(ref[k].norm()) + 1e-12, (k, defer, side, hooked, group)
    finally:
        hip.glue().defer_wgrad_reduce(before)
        hip.glue().wgrad_stream_mode(stream_before)
        hip.glue().wgrad_group_mode(group_before)
    # outside a backward pass the reduce runs at once
    w = _rand((64, 64, 3, 3), torch.Generator().manual_seed(7)).to(dev).contiguous(memory_format=torch.channels_last)
    xx = _rand((32, 64, 32, 32), torch.Generator().manual_seed(8)).to(dev).contiguous(memory_format=torch.channels_last)
    dy = _rand((32, 64, 32, 32), torch.Generator().manual_seed(9)).to(dev).contiguous(memory_format=torch.channels_last)
    dw = hip.conv2d_bwd_weight(xx, dy, 3, 1, 1, dtype=torch.float32)
    refdw = torch.nn.grad.conv2d_weight(xx.float(), tuple(w.shape), dy.float(), padding=1)
    assert (dw - refdw).abs().max().item() <= 3e-3 * refdw.abs().max().item() + 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_wgrad_group_vs_torch_fp32(dtype):
    """epi_wgrad_group: MANY backward-weight GEMMs in one launch per tile class (a ResNet stage's worth: 1x1, 3x3 stride 1 | 2, the
    stride-2 projection, 64-output-channel layers on the narrow tile, a transposed convolution) against torch's fp32 weight gradients of the
    same bf16-rounded operands -- split and unsplit reductions side by side, results in the channels_last weight order."""
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(21)
    # (kind, Cin, Cout, k, stride, H, B)
    geos = [("conv", 256, 64, 1, 1, 16, 8), ("conv", 64, 64, 3, 1, 16, 8), ("conv", 64, 256, 1, 1, 16, 8), ("conv", 256, 128, 1, 1, 16, 8),
            ("conv", 128, 128, 3, 2, 16, 8), ("conv", 128, 512, 1, 1, 8, 8), ("conv", 256, 512, 1, 2, 16, 8), ("conv", 512, 128, 1, 1, 8, 8),
            ("conv", 128, 128, 3, 1, 8, 8), ("conv", 1024, 256, 1, 1, 4, 4), ("conv", 256, 256, 3, 1, 4, 4), ("conv", 64, 64, 1, 1, 32, 16),
            ("deconv", 128, 64, 4, 2, 8, 4), ("conv", 64, 128, 3, 2, 16, 4)]
    jobs, refs = [], []
    for kind, cin, cout, k, stride, h, b in geos:
        x = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
        if kind == "deconv":
            dy = _rand((b, cout, 2 * h, 2 * h), gen).to(dev).contiguous(memory_format=torch.channels_last)
            w = torch.zeros(cin, cout, 4, 4, device=dev, requires_grad=True)
            F.conv_transpose2d(x.float(), w, stride=2, padding=1).backward(dy.float())
            refs.append(w.grad)
            jobs.append((kind, x, dy, 4, 2, 1))
        else:
            pad = k // 2
            ho = (h + 2 * pad - k) // stride + 1
            dy = _rand((b, cout, ho, ho), gen).to(dev).contiguous(memory_format=torch.channels_last)
            refs.append(torch.nn.grad.conv2d_weight(x.float(), (cout, cin, k, k), dy.float(), stride=stride, padding=pad))
            jobs.append((kind, x, dy, k, stride, pad))
    assert len(jobs) <= hip.wgrad_group_max()
    slab, nsplit = hip.wgrad_group_plan([(kind, b, h, h, cin, cout, k, stride, (1 if kind == "deconv" else k // 2)) for kind, cin, cout, k, stride, h, b in geos])
    assert max(nsplit) > 1 and min(nsplit) == 1 and slab > 0, (slab, nsplit)      # the case mixes split and unsplit reductions
    outs = hip.wgrad_group(jobs, dtype=dtype)
    for (kind, cin, cout, k, stride, h, b), got, ref in zip(geos, outs, refs):
        assert got.shape == ref.shape and got.dtype == dtype, (kind, cin, cout, k)
        if kind == "conv":
            assert got.is_contiguous(memory_format=torch.channels_last) or k == 1
        tol = (2e-3 if dtype == torch.float32 else 2 ** -7) * ref.abs().max().item() + 1e-5
        assert (got.float() - ref).abs().max().item() <= tol, (kind, cin, cout, k, stride, h, float((got.float() - ref).abs().max()), tol)
    # one item alone == the single-GEMM entry point on the same operands (bitwise when neither splits its reduction)
    kind, x, dy, k, stride, pad = jobs[9]
    single = hip.conv2d_bwd_weight(x, dy, k, stride, pad, dtype=dtype)
    alone = hip.wgrad_group([jobs[9]], dtype=dtype)[0]
    assert (alone.float() - single.float()).abs().max().item() <= 2e-3 * single.float().abs().max().item()


@pytest.mark.parametrize("layout", ["nchw_f32", "nhwc_f32", "nhwc_bf16"])
def test_stem_convolution_vs_torch_fp32(layout):
    """The 7x7 / stride-2 / pad-3 convolution on 3 channels (pose3d_resnet.py:99) through the space-to-depth gather GEMM: forward and weight
    gradient against torch's fp32 convolution of the bf16-rounded operands; then the whole stem node (conv -> BatchNorm -> ReLU) against fp32 autograd."""
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(17)
    b, h, cout = 4, 64, 64
    x = torch.randn((b, 3, h, h), generator=gen)
    w = torch.randn((cout, 3, 7, 7), generator=gen) * (2.0 / 147) ** 0.5
    xin = x.to(dev)
    if layout.startswith("nhwc"):
        xin = xin.contiguous(memory_format=torch.channels_last)
    if layout.endswith("bf16"):
        xin = xin.to(torch.bfloat16)
    xr, wr = x.to(torch.bfloat16).float().to(dev), w.to(torch.bfloat16).float().to(dev).requires_grad_(True)
    ref = F.conv2d(xr, wr, stride=2, padding=3)
    y, s2d = hip.stem_conv_fwd(xin, w.to(dev))
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert (y.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    wcl = w.to(dev).contiguous(memory_format=torch.channels_last)                 # the model's weights are channels_last
    y2, _ = hip.stem_conv_fwd(xin, wcl)
    assert torch.equal(y2, y)
    dy = _rand(tuple(ref.shape), gen).to(dev).contiguous(memory_format=torch.channels_last)
    ref.backward(dy.float())
    dw = hip.stem_conv_bwd_weight(s2d, dy, (h, h))
    assert (dw - wr.grad).abs().max().item() <= 3e-3 * wr.grad.abs().max().item() + 1e-5
    # the node: conv -> BatchNorm (training) -> ReLU
    gamma, beta = (torch.rand(cout, generator=gen) + 0.5).to(dev), (torch.randn(cout, generator=gen) * 0.1).to(dev)
    bn = hip_bn_module(cout, gamma, beta, dev)
    wp = torch.nn.Parameter(wcl.clone())
    g_, b_, rm, rv, nbt, sums_ws, bwd_sums = bn._tensors()
    out = hip.glue().stem_conv_bn_act(xin, wp, g_, b_, rm, rv, nbt, sums_ws, bwd_sums, bn._flags, True, 0.1, 1e-5, True, False)
    wf, gf, bf_ = wr.detach().clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    refo = F.relu(F.batch_norm(F.conv2d(xr, wf, stride=2, padding=3), None, None, gf, bf_, True, 0.1, 1e-5))
    assert (out.float() - refo).abs().max().item() <= 3e-2 * refo.abs().max().item()
    dyo = _rand(tuple(refo.shape), gen).to(dev).contiguous(memory_format=torch.channels_last)
    out.backward(dyo)
    refo.backward(dyo.float())
    torch.cuda.synchronize()
    for got, want, what in ((wp.grad, wf.grad, "dw"), (bn.weight.grad, gf.grad, "dgamma"), (bn.bias.grad, bf_.grad, "dbeta")):
        got, want = got.float(), want.float()
        assert float((got - want).norm()) <= 4e-2 * float(want.norm()) + 1e-6, (what, float((got - want).norm()), float(want.norm()))


@pytest.mark.parametrize("training", [True, False])
def test_stem_node_with_the_pool_inside(training):
    """stem_conv_bn_act(..., pool=True): BatchNorm + ReLU applied on the way into the 3x3 stride-2 max-pool (epi_bn_finalize +
    epi_maxpool3x3s2_bn_relu_fwd; the normalised tensor is never written) against the two-pass form of the same kernels (stem node, then the
    max-pool node): the same values rounded at the same place -> bit-identical outputs, statistics and input-side gradients."""
    import copy
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(41)
    b, cout, h = 4, 64, 64
    x = _rand((b, 3, h, h), gen).float().to(dev)
    w0 = _rand((cout, 3, 7, 7), gen, scale=(2.0 / 147) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    gamma, beta = (torch.rand(cout, generator=gen) + 0.5).to(dev), (torch.randn(cout, generator=gen) * 0.1).to(dev)
    outs = []
    for pool in (True, False):
        bn = hip_bn_module(cout, gamma, beta, dev)
        with torch.no_grad():
            bn.running_mean.normal_(0, 0.1, generator=None)
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, cout))
            bn.running_var.copy_(torch.linspace(0.5, 1.5, cout))
        bn.train(training)
        wp = torch.nn.Parameter(w0.clone())
        g_, b_, rm, rv, nbt, sums_ws, bwd_sums = bn._tensors()
        y = hip.glue().stem_conv_bn_act(x, wp, g_, b_, rm, rv, nbt, sums_ws, bwd_sums, bn._flags, training, 0.1, 1e-5, True, pool)
        if not pool:
            y = hip.glue().maxpool3x3s2(y)
        grads = None
        if training:
            dy = _rand(tuple(y.shape), torch.Generator().manual_seed(42)).to(dev).contiguous(memory_format=torch.channels_last)
            y.backward(dy)
            torch.cuda.synchronize()
            grads = (wp.grad.float().clone(), bn.weight.grad.clone(), bn.bias.grad.clone())
        outs.append((y.detach().clone(), grads, bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked)))
    (ya, ga, ma, va, na), (yb, gb, mb, vb, nb) = outs
    assert ya.shape == (b, cout, h // 4, h // 4) and torch.equal(ya, yb)
    assert na == nb
    if training:
        # the batch statistics come from the same epilogue atomics in both forms (their order varies run to run: last-bit differences)
        assert float((ma - mb).abs().max()) <= 1e-6 and float((va - vb).abs().max()) <= 1e-6
        for u, v in zip(ga, gb):
            assert float((u - v).norm()) <= 2e-3 * float(v.norm()) + 1e-9
    else:
        assert torch.equal(ma, mb) and torch.equal(va, vb)


def hip_bn_module(c, gamma, beta, dev):
    from epipolarpose_amd.models.fused import FusedBatchNormAct
    bn = FusedBatchNormAct(c, momentum=0.1, relu=True).to(dev)
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    bn.train()
    return bn


# (Cin = channels of dx, Cout, k, stride, H of dx, batch, mask from y, addend): the launch each takes is noted
BNRED_CASES = [
    (64, 256, 1, 1, 16, 4, False, False),      # plain 1x1, 64 columns (tall tile)
    (256, 64, 1, 1, 64, 4, True, True),        # K = 64, N = 256, 16 384 rows (a unit's first stage, shortcut added): A-stationary shape, kept on the generic kernel
    (512, 128, 1, 1, 32, 16, True, True),      # the same with K = 128
    (64, 64, 3, 1, 32, 4, False, False),       # patch-stationary 3x3
    (128, 128, 3, 1, 16, 8, False, False),     # patch-stationary 3x3, 128 columns, too few rows: split launch, no fusion
    (128, 128, 3, 1, 32, 32, False, False),    # patch-stationary 3x3, 128 columns (ResNet-50 layer2 at the bench shape)
    (128, 128, 3, 2, 32, 4, False, False),     # stride 2: four parity phases, scattered rows
    (256, 512, 1, 2, 32, 4, True, True),       # the downsample projection's geometry (three phases without taps)
    (256, 1088, 1, 1, 32, 16, False, False),   # the final layer's backward-data (K = 1088): 128-column tiles
]


BNRED_MAY_SPLIT = {c for c in BNRED_CASES if c[:6] == (128, 128, 3, 1, 16, 8)}       # 2 048 rows x 1 152 of K: channel-split patch launch


@pytest.mark.parametrize("case", BNRED_CASES, ids=["%dfrom%d_k%ds%d_h%d_b%d_y%d_add%d" % c for c in BNRED_CASES])
def test_conv2d_bwd_data_with_fused_batchnorm_backward_reduction(case):
    """epi_conv2d_bwd_data_bnred: the epilogue masks the gradient (dz = dx * [bn output > 0]) and accumulates sum(dz), sum(dz * xhat)
    per channel.  Against the SAME launch without the fusion (the bf16 dx it would have written) masked in PyTorch, and float64 sums."""
    from epipolarpose_amd import hip
    cin, cout, k, stride, h, b, from_y, with_addend = case
    pad = k // 2
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin + 3 * cout + k + stride)
    ho = (h + 2 * pad - k) // stride + 1
    dy = _rand((b, cout, ho, ho), gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = _rand((cout, cin, k, k), gen, scale=(2.0 / (cin * k * k)) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    z = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    addend = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last) if with_addend else None
    mean = torch.randn(cin, generator=gen) * 0.2
    rstd = torch.rand(cin, generator=gen) + 0.5
    gamma, beta = torch.randn(cin, generator=gen), torch.randn(cin, generator=gen) * 0.3
    scale = gamma * rstd
    shift = beta - mean * scale
    bn = torch.stack([mean, rstd, scale, shift]).to(dev).contiguous()
    y = None
    if from_y:          # residual + ReLU layer: the mask is the saved output, NOT a function of z alone
        y = torch.relu(z.float() * scale.to(dev).view(1, -1, 1, 1) + shift.to(dev).view(1, -1, 1, 1) + _rand((b, cin, h, h), gen).to(dev).float())
        y = y.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wb = hip.conv2d_pack_weight_bwd(w, stride, pad)
    plain = hip.conv2d_bwd_data(dy, wb, (b, cin, h, h), k, stride, pad, addend=addend)
    dz, sums, fused = hip.conv2d_bwd_data_bnred(dy, wb, (b, cin, h, h), k, stride, pad, z, bn, relu=True, y=y, addend=addend)
    if not fused:       # a split-K launch (few rows): the contract is "plain gradient, accumulator untouched" -- and only where that is expected
        assert case in BNRED_MAY_SPLIT, "this geometry is expected on a launch that carries the fused reduction"
        assert torch.equal(dz, plain) and float(sums.abs().max()) == 0.0
        return
    zf = z.float()
    if from_y:
        mask = y.float() > 0
    else:
        t = zf * bn[2].view(1, -1, 1, 1) + bn[3].view(1, -1, 1, 1)
        mask = t > 0
        near = t.abs() <= 1e-6 * (zf.abs() * bn[2].abs().view(1, -1, 1, 1) + bn[3].abs().view(1, -1, 1, 1))   # a fused multiply-add may decide these differently
    ref = torch.where(mask, plain, torch.zeros_like(plain))
    same = dz == ref
    if not from_y:
        same = same | near
    assert bool(same.all()), int((~same).sum())
    d64 = dz.double()
    xhat = (zf.double() - bn[0].double().view(1, -1, 1, 1)) * bn[1].double().view(1, -1, 1, 1)
    ref_s = torch.stack([d64.sum(dim=(0, 2, 3)), (d64 * xhat).sum(dim=(0, 2, 3))])
    scale_s = torch.stack([d64.abs().sum(dim=(0, 2, 3)), (d64 * xhat).abs().sum(dim=(0, 2, 3))]) + 1e-12
    assert float(((sums.double() - ref_s).abs() / scale_s).max()) <= 2e-6          # fp32 accumulation of a few thousand terms per tile, then atomics
    # no ReLU: nothing is masked, the sums still come out
    dz2, sums2, fused2 = hip.conv2d_bwd_data_bnred(dy, wb, (b, cin, h, h), k, stride, pad, z, bn, relu=False, y=None, addend=addend)
    assert fused2 and torch.equal(dz2, plain)
    p64 = plain.double()
    ref2 = torch.stack([p64.sum(dim=(0, 2, 3)), (p64 * xhat).sum(dim=(0, 2, 3))])
    scale2 = torch.stack([p64.abs().sum(dim=(0, 2, 3)), (p64 * xhat).abs().sum(dim=(0, 2, 3))]) + 1e-12
    assert float(((sums2.double() - ref2).abs() / scale2).max()) <= 2e-6


@pytest.mark.parametrize("with_red", [False, True])
def test_conv2d_bwd_data_half_resolution_addend(with_red):
    """addend_step 2: the shortcut gradient through a 1x1 stride-2 projection handed over at half resolution (the even pixels) must give exactly
    what the zero-expanded full-resolution addend gives -- with and without the fused BatchNorm-backward reduction in the same epilogue."""
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(77)
    b, cin, cout, h = 8, 256, 128, 32                     # a stride-2 Bottleneck's first stage: dx [8, 256, 32, 32] from dy [8, 128, 32, 32]
    assert hip.load().epi_conv2d_bwd_data_half_addend_ok(b, h, h, cin, cout) == 1
    assert hip.load().epi_conv2d_bwd_data_half_addend_ok(1, 8, 8, cin, 2048) == 0            # few rows, deep K: a split launch
    dy = _rand((b, cout, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    w = _rand((cout, cin, 1, 1), gen, scale=(2.0 / cin) ** 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    half = _rand((b, cin, h // 2, h // 2), gen).to(dev).contiguous(memory_format=torch.channels_last)
    full = torch.zeros((b, cin, h, h), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    full[:, :, ::2, ::2] = half
    z = _rand((b, cin, h, h), gen).to(dev).contiguous(memory_format=torch.channels_last)
    y = torch.relu(z.float() + _rand((b, cin, h, h), gen).to(dev).float()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    bn = torch.stack([torch.randn(cin, generator=gen) * 0.2, torch.rand(cin, generator=gen) + 0.5, torch.randn(cin, generator=gen),
                      torch.randn(cin, generator=gen) * 0.3]).to(dev).contiguous()
    wb = hip.conv2d_pack_weight_bwd(w, 1, 0)
    if with_red:
        ref, ref_s, f1 = hip.conv2d_bwd_data_bnred(dy, wb, (b, cin, h, h), 1, 1, 0, z, bn, relu=True, y=y, addend=full)
        got, got_s, f2 = hip.conv2d_bwd_data_bnred(dy, wb, (b, cin, h, h), 1, 1, 0, z, bn, relu=True, y=y, addend=half, addend_step=2)
        assert f1 and f2 and torch.equal(got, ref)
        assert float((got_s - ref_s).abs().max()) <= 1e-4 * float(ref_s.abs().max())          # (atomics order)
    else:
        ref = hip.conv2d_bwd_data(dy, wb, (b, cin, h, h), 1, 1, 0, addend=full)
        got, _, fused = hip.conv2d_bwd_data_bnred(dy, wb, (b, cin, h, h), 1, 1, 0, z, bn, relu=False, y=None, addend=half, addend_step=2)
        assert fused and torch.equal(got, ref)


def test_fused_batchnorm_backward_reduction_in_the_network():
    """The whole pose network, one backward pass with the reductions fused into the backward-data launches (across autograd nodes:
    unit -> unit, head -> backbone, final layer -> head) and one without: the fused pass must actually happen (counted by the glue),
    and every parameter gradient must agree with the unfused pass to the noise of two unfused passes (atomics)."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    import copy
    dev = torch.device("cuda:0")
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.EXTRA.NUM_LAYERS = 50
    j = 4
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, 16, [128, 128]
    torch.manual_seed(5)
    base = get_pose_net(cfg, is_train=False).to(dev).train()
    x = torch.randn(16, 3, 128, 128, device=dev)
    gt = (torch.rand(16, 3 * j, device=dev) - 0.5) * 0.4
    vis = torch.ones(16, 3 * j, device=dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)

    def grads(fuse):
        prev = hip.glue().bn_bwd_fuse_mode(1 if fuse else 0)
        try:
            m = copy.deepcopy(base)
            hip.glue().bn_bwd_fuse_counts(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = m(x)
            crit(out, gt, vis).backward()
            torch.cuda.synchronize()
            counts = hip.glue().bn_bwd_fuse_counts(True)
            return {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}, counts, out.detach().float()
        finally:
            hip.glue().bn_bwd_fuse_mode(prev)
    a, ca, oa = grads(False)
    b, cb, ob = grads(True)
    assert ca == [0, 0]
    # ResNet-50: 16 units x 3 BatchNorms + 3 head layers + stem + 4 projections = 56; the stem (max-pool in between), the projections
    # (fed by the shortcut gradient) and the split-K launches of the small late layers keep their own reduction
    assert cb[0] >= 20 and cb[1] == 0, cb              # (27 at this size; 128 x 128 images leave layer3 / layer4 to split launches)
    # the forward passes are two runs of the same code: where they already differ by an atomics-order rounding flip the gradients are
    # not comparable at random initialisation (tests/test_hip_step_in_backward.py); compare only like with like
    if float((oa - ob).abs().max()) <= 1e-3 * float(oa.abs().max()):
        for k in a:
            ca_, cb_ = a[k].reshape(-1).double(), b[k].reshape(-1).double()
            cos = float(ca_ @ cb_ / (ca_.norm() * cb_.norm() + 1e-300))
            assert cos >= 0.98, (k, cos)
