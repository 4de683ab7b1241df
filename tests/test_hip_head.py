"""GPU parity: MFMA head kernels (through the C ABI) vs. plain PyTorch fp32 references of the same ops on the
same bf16-rounded inputs.  Tolerance: bf16 output rounding (2^-8 relative) + fp32 accumulation-order noise."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run through gpurun)"
    from epipolarpose_amd import hip
    hip.load()
    return torch.device("cuda:0")


def rnd(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def close(got, ref, rel=1.2e-2, outliers=0.0):
    """max |got - ref| <= rel * max |ref|; ``outliers`` = fraction of elements allowed to miss it (ReLU-boundary flips:
    an activation within 1e-6 of zero can land on either side when batch statistics are summed in a different order)."""
    diff = (got.float() - ref).abs()
    tol = rel * ref.abs().max().item()
    if outliers > 0:
        bad = (diff > tol).float().mean().item()
        assert bad <= outliers, "%.3g of the elements exceed tol %.4g (max err %.4g)" % (bad, tol, diff.max().item())
        return
    err = diff.max().item()
    assert err <= tol, "max err %.4g > tol %.4g" % (err, tol)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 1088, 256), (1000, 200, 128), (4096, 256, 1088), (77, 36, 192), (512, 256, 24), (300, 40, 200),
                                   # 256x256-tile configuration: ragged M and N, K tail, split-K (few tiles) and no split (many tiles)
                                   (5164, 2528, 520), (2000, 2040, 4104), (65536, 256, 512), (2048, 1024, 2048), (700, 456, 576),
                                   # ... its tiles are ordered in groups of 4 tile rows: 21 (remainder group of 1), 6 (of 2) and 3 tile rows (a lone group of 3)
                                   (1500, 512, 4096),
                                   # A-stationary kernel (K in {64,128,256}, wide N, many rows): ragged M and N
                                   (16500, 1088, 256), (20000, 520, 128), (16384, 264, 64)])
def test_gemm_vs_torch(dev, m, n, k):
    from epipolarpose_amd import hip
    a = rnd((m, k), dev, 1).to(torch.bfloat16)
    bt = rnd((n, k), dev, 2).to(torch.bfloat16)
    ref = a.float() @ bt.float().t()          # fp32 matmul of the bf16-rounded values (asymmetric operands)
    close(hip.gemm_bf16(a, bt), ref)
    bias = rnd((n,), dev, 3)
    c32 = hip.gemm_bf16(a, bt, bias=bias, out_dtype=torch.float32)
    close(c32, ref + bias, rel=2e-5)
    close(hip.gemm_bf16(a, bt, bias=bias), ref + bias)         # bf16 output with the bias fused (both tile configurations)


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric B catches a transposed C write (HIP guide section 3)."""
    from epipolarpose_amd import hip
    a = torch.eye(128, 128, device=dev, dtype=torch.bfloat16)
    bt = (torch.arange(256 * 128, device=dev).reshape(256, 128) % 251).to(torch.bfloat16)
    c = hip.gemm_bf16(a, bt, out_dtype=torch.float32)
    assert torch.equal(c, bt.float().t().contiguous())
    # the 256x256-tile configuration (K >= 512, bf16 output through the LDS-transposed epilogue), ragged in M and N
    a = torch.eye(5200, 512, device=dev, dtype=torch.bfloat16)
    bt = (torch.arange(2528 * 512, device=dev).reshape(2528, 512) % 251).to(torch.bfloat16)
    c = hip.gemm_bf16(a, bt)
    want = torch.zeros(5200, 2528, device=dev)
    want[:512] = bt.float().t()
    assert c.dtype == torch.bfloat16 and torch.equal(c.float(), want)


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 8, 8, 128, 64), (3, 5, 7, 64, 32), (1, 16, 16, 256, 256), (2, 4, 4, 2048, 256),
                                            (4, 16, 16, 256, 256), (14, 32, 32, 128, 256), (4, 32, 32, 1024, 256)])
def test_deconv_fwd_bwd_data_vs_torch(dev, b, h, w, cin, cout):
    from epipolarpose_amd import hip
    x = rnd((b, cin, h, w), dev, 4).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = rnd((cin, cout, 4, 4), dev, 5, scale=(1.0 / cin) ** 0.5)
    wp, wb = hip.deconv_pack_weight(wt)
    w32 = wt.to(torch.bfloat16).float()
    y = hip.deconv4x4s2_fwd(x, wp)
    assert y.shape == (b, cout, 2 * h, 2 * w) and y.is_contiguous(memory_format=torch.channels_last)
    ref = F.conv_transpose2d(x.float(), w32, None, stride=2, padding=1)
    close(y, ref)
    dy = rnd((b, cout, 2 * h, 2 * w), dev, 6).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    if cout % 64 == 0:
        dx = hip.deconv4x4s2_bwd_data(dy, wb)
        xr = x.float().requires_grad_(True)
        F.conv_transpose2d(xr, w32, None, stride=2, padding=1).backward(dy.float())
        close(dx, xr.grad)


# (the last two: long reductions over large operands run on the 256 x 256 tile since round 4 -- the final layer's shape among them)
@pytest.mark.parametrize("r,i,j", [(4096, 128, 128), (5000, 1088, 256), (300, 72, 40), (131072, 256, 128), (32768, 1024, 256), (131072, 1088, 256)])
def test_gemm_tn_vs_torch(dev, r, i, j):
    from epipolarpose_amd import hip
    a = rnd((r, i), dev, 7).to(torch.bfloat16)
    b = rnd((r, j), dev, 8).to(torch.bfloat16)
    ref = a.float().t() @ b.float()
    got = hip.gemm_tn_bf16(a, b)
    assert got.dtype == torch.float32
    close(got, ref, rel=3e-5 * max(1.0, (r / 4096) ** 0.5))
    np_sum = hip.column_sum_bf16(a)
    close(np_sum, a.float().sum(0), rel=1e-5)


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 8, 8, 128, 64), (3, 5, 7, 64, 72), (32, 16, 16, 256, 256), (8, 64, 64, 256, 256)])
def test_deconv_bwd_weight_vs_torch(dev, b, h, w, cin, cout):
    from epipolarpose_amd import hip
    x = rnd((b, cin, h, w), dev, 9).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = rnd((b, cout, 2 * h, 2 * w), dev, 10).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.zeros((cin, cout, 4, 4), device=dev, requires_grad=True)
    F.conv_transpose2d(x.float(), wt, None, stride=2, padding=1).backward(dy.float())
    close(hip.deconv4x4s2_bwd_weight(x, dy), wt.grad, rel=1e-4)


@pytest.mark.parametrize("relu,res", [(True, False), (True, True), (False, False), (False, True)])
@pytest.mark.parametrize("shape", [(4, 64, 9, 7), (32, 256, 16, 16), (2, 2048, 2, 2), (8, 64, 128, 128)])
def test_fused_bn_act_vs_torch(dev, relu, res, shape):
    from epipolarpose_amd.models.fused import FusedBatchNormAct
    b, c, h, w = shape
    x = rnd(shape, dev, 11, 2.0).add_(0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = rnd(shape, dev, 12).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    m = FusedBatchNormAct(c, momentum=0.1, relu=relu).to(dev)
    with torch.no_grad():
        m.weight.copy_(rnd((c,), dev, 13).abs() + 0.5)
        m.bias.copy_(rnd((c,), dev, 14) * 0.3)
    ref = torch.nn.BatchNorm2d(c, momentum=0.1).to(dev)
    ref.load_state_dict({k: v for k, v in m.state_dict().items()})
    xg = x.clone().requires_grad_(True)
    rg = r.clone().requires_grad_(True) if res else None
    y = m(xg, residual=rg)
    x32 = x.float().requires_grad_(True)
    r32 = r.float().requires_grad_(True) if res else None
    t = ref(x32)
    if res:
        t = t + r32
    if relu:
        t = torch.relu(t)
    close(y, t, rel=1e-2)
    dy = rnd(shape, dev, 15).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y.backward(dy)
    # reference backward with the same activation mask as the fused kernel sees it
    t.backward(dy.float())
    close(xg.grad, x32.grad, rel=2e-2, outliers=1e-5 if relu else 0.0)
    if res:
        close(rg.grad, r32.grad, rel=1e-2, outliers=1e-5 if relu else 0.0)
    close(m.weight.grad, ref.weight.grad, rel=2e-2)
    close(m.bias.grad, ref.bias.grad, rel=2e-2)
    close(m.running_mean, ref.running_mean, rel=1e-3)
    close(m.running_var, ref.running_var, rel=1e-3)
    assert int(m.num_batches_tracked) == 1
    m.eval(); ref.eval()
    with torch.no_grad():
        ye = m(x, residual=r)
        te = ref(x.float())
        if res:
            te = te + r.float()
        if relu:
            te = torch.relu(te)
    close(ye, te, rel=1e-2)


def test_fused_bn_call_patterns(dev):
    """The forward / backward accumulators are cleared by each other's kernels in the steady fwd -> bwd pattern; every other
    pattern (forward only, the layer twice in one graph, repeated steps) must give the same numbers as nn.BatchNorm2d."""
    from epipolarpose_amd.models.fused import FusedBatchNormAct
    c, shape = 64, (4, 64, 6, 5)
    m = FusedBatchNormAct(c, relu=True).to(dev)
    ref = torch.nn.BatchNorm2d(c).to(dev)
    xs = [rnd(shape, dev, 40 + i, 1.5).add_(0.2 * i).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for i in range(4)]

    def ref_out(x):
        return torch.relu(ref(x.float()))
    # (1) three steady steps
    for i in range(3):
        xg, x32 = xs[i].clone().requires_grad_(True), xs[i].float().requires_grad_(True)
        y, t = m(xg), torch.relu(ref(x32))
        close(y, t, rel=1e-2)
        m.weight.grad = m.bias.grad = ref.weight.grad = ref.bias.grad = None
        y.float().sum().backward()
        t.sum().backward()
        close(m.weight.grad, ref.weight.grad, rel=2e-2)
        close(m.bias.grad, ref.bias.grad, rel=2e-2)
    # (2) training-mode forwards without a backward
    with torch.no_grad():
        for i in range(2):
            close(m(xs[i]), ref_out(xs[i]), rel=1e-2)
    # (3) the layer twice inside one autograd graph, then a normal step again
    for rounds in range(2):
        xa, xb = xs[2].clone().requires_grad_(True), xs[3].clone().requires_grad_(True)
        a32, b32 = xs[2].float().requires_grad_(True), xs[3].float().requires_grad_(True)
        m.weight.grad = m.bias.grad = ref.weight.grad = ref.bias.grad = None
        (m(xa).float().sum() + 2.0 * m(xb).float().sum()).backward()
        (torch.relu(ref(a32)).sum() + 2.0 * torch.relu(ref(b32)).sum()).backward()
        close(m.weight.grad, ref.weight.grad, rel=2e-2)
        close(m.bias.grad, ref.bias.grad, rel=2e-2)
        close(xa.grad, a32.grad, rel=2e-2, outliers=1e-3)
        close(xb.grad, b32.grad, rel=2e-2, outliers=1e-3)
    close(m.running_mean, ref.running_mean, rel=2e-3)
    close(m.running_var, ref.running_var, rel=2e-3)
    assert int(m.num_batches_tracked) == int(ref.num_batches_tracked) == 9


@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,pad", [(4, 16, 16, 64, 128, 1, 1, 0), (2, 12, 10, 64, 72, 3, 1, 1), (3, 9, 11, 128, 64, 3, 2, 1),
                                                         (32, 16, 16, 256, 256, 3, 1, 1), (8, 32, 32, 256, 512, 1, 1, 0)])
def test_conv2d_bwd_weight_vs_torch(dev, b, h, w, cin, cout, k, stride, pad):
    from epipolarpose_amd import hip
    x = rnd((b, cin, h, w), dev, 21).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = rnd((b, cout, ho, wo), dev, 22).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.zeros((cout, cin, k, k), device=dev, requires_grad=True)
    F.conv2d(x.float(), wt, None, stride=stride, padding=pad).backward(dy.float())
    got = hip.conv2d_bwd_weight(x, dy, k, stride, pad)
    assert got.shape == wt.shape and got.dtype == torch.float32 and got.is_contiguous(memory_format=torch.channels_last)
    close(got, wt.grad, rel=1e-4)
    got16 = hip.conv2d_bwd_weight(x, dy, k, stride, pad, dtype=torch.bfloat16)
    assert got16.dtype == torch.bfloat16
    close(got16, wt.grad, rel=1e-2)


@pytest.mark.parametrize("cin,cout,h", [(128, 64, 8), (2048, 256, 4), (256, 256, 16)])
def test_deconv_channels_last_weight_forms_and_module(dev, cin, cout, h):
    """The channels_last weight memory [Cin][kh][kw][Cout] is the backward-data operand as it stands; the packed forward operand
    equals the one produced from the contiguous layout; the module trains through a bf16 copy with a bf16 gradient in the weight's
    own layout, and FusedAdam keeps ``weight_phase`` in step with the updated copy."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.models.fused import Deconv4x4s2
    from epipolarpose_amd.optim import FusedAdam
    wt = rnd((cin, cout, 4, 4), dev, 21, scale=(1.0 / cin) ** 0.5)
    wp_ref, wb_ref = hip.deconv_pack_weight(wt)                                   # from the contiguous [Cin][Cout][4][4] layout
    wp, w16 = hip.deconv_weight_forms(wt.contiguous(memory_format=torch.channels_last))
    assert torch.equal(wp, wp_ref)
    assert torch.equal(torch.as_strided(w16, (cin, 16 * cout), (16 * cout, 1)), wb_ref)
    mod = Deconv4x4s2(cin, cout).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        mod.weight.copy_(wt)
    x = rnd((2, cin, h, h), dev, 22).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = rnd((2, cout, 2 * h, 2 * h), dev, 23).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_(True)
    wr = wt.to(torch.bfloat16).float().requires_grad_(True)
    ref = F.conv_transpose2d(xr, wr, None, stride=2, padding=1)
    ref.backward(dy.float())
    xg = x.clone().requires_grad_(True)
    y = mod(xg)                                                                    # fp32 master path: converted + packed per call
    close(y, ref)
    y.backward(dy)
    close(xg.grad, xr.grad)
    assert mod.weight.grad.dtype == torch.float32 and mod.weight.grad.stride() == mod.weight.stride()
    close(mod.weight.grad, wr.grad, rel=2e-3)
    opt = FusedAdam(mod, lr=1e-3)                                                  # bf16 training copy + optimizer-maintained weight_phase
    assert mod.weight_lp.dtype == torch.bfloat16 and mod.weight_phase.shape == (4, cout, 4 * cin)
    assert torch.equal(mod.weight_phase, wp_ref)
    xg2 = x.clone().requires_grad_(True)
    y2 = mod(xg2)
    close(y2, ref)
    y2.backward(dy)
    g = mod.weight_lp.grad
    assert g.dtype == torch.bfloat16 and g.stride() == mod.weight.stride()
    close(g, wr.grad, rel=1.5e-2)
    opt.step()
    assert torch.equal(mod.weight_lp.detach().float(), mod.weight.detach().to(torch.bfloat16).float())
    assert torch.equal(mod.weight_phase, hip.deconv_pack_weight(mod.weight.detach())[0])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 64, 32, 32), (3, 8, 17, 23), (1, 64, 128, 128), (2, 16, 9, 8)])
def test_maxpool3x3s2_matches_torch(shape):
    """epi_maxpool3x3s2_fwd / _bwd against nn.MaxPool2d(3, 2, 1) (pose3d_resnet.py:104) in fp32: the selected values are exact, the
    gradient goes to the library's arg-max (first of equal maxima in scan order -- post-ReLU inputs are full of ties) and a pixel
    selected by several windows gets their fp32 sum rounded once.  Odd sizes, a NaN and a window of -inf included."""
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=gen).clamp_min(0.0)              # ReLU output: ~half the entries are exact zeros
    x[0, 0, 0, 0] = float("nan")
    x[-1, :, -3:, -3:] = float("-inf")
    x = x.to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    dy_shape = (shape[0], shape[1], (shape[2] - 1) // 2 + 1, (shape[3] - 1) // 2 + 1)
    dy = torch.randn(dy_shape, generator=gen).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    y, pos = hip.maxpool3x3s2_fwd(x)
    dx = hip.maxpool3x3s2_bwd(dy, pos, shape[2:])
    # reference on an NCHW-contiguous tensor: the library's channels_last kernel leaves the gradient of an all -inf window at pixel
    # (0, 0) of the image (its index buffer starts at 0), its NCHW kernel at the window's first pixel inside the image, as here
    xr = x.float().contiguous().requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 3, 2, 1)
    yr.backward(dy.float().contiguous())
    assert y.shape == yr.shape and dx.shape == xr.shape
    assert torch.equal(torch.nan_to_num(y.float(), nan=12345.0), torch.nan_to_num(yr.detach(), nan=12345.0))
    ref_dx = xr.grad.to(torch.bfloat16).float()
    bad = (dx.float() != ref_dx).nonzero()
    assert bad.numel() == 0, (bad[:8].tolist(), dx.float()[tuple(bad[0])].item(), ref_dx[tuple(bad[0])].item())
    # through the autograd node of the C++ glue
    xg = x.clone().requires_grad_(True)
    yg = hip.glue().maxpool3x3s2(xg)
    yg.backward(dy)
    assert torch.equal(torch.nan_to_num(yg.float(), nan=12345.0), torch.nan_to_num(y.float(), nan=12345.0)) and torch.equal(xg.grad, dx)


@pytest.mark.gpu
@pytest.mark.parametrize("geo", [(32, 256, 256, 32), (32, 256, 256, 16), (8, 2048, 256, 8), (2, 64, 64, 8)], ids=lambda g: "b%d_%dto%d_h%d" % g)
def test_deconv_fwd_fused_bn_statistics(geo):
    """BatchNorm batch sums from the transposed convolution's epilogue (four output-parity phases in one launch): equal to the column
    sums of the bf16 tensor it wrote; a launch that cannot do it (split-K) says so and leaves the accumulator alone."""
    from epipolarpose_amd import hip
    b, cin, cout, h = geo
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(b + cin + h)
    x = torch.randn((b, cin, h, h), generator=gen).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((cin, cout, 4, 4), generator=gen) * (1.0 / (4 * cin)) ** 0.5).to(dev)
    wp, _ = hip.deconv_pack_weight(w)
    sums = torch.zeros(hip.bn_sum_copies(cout) * 2 * cout, dtype=torch.float32, device=dev)
    y, done = hip.deconv4x4s2_fwd(x, wp, bn_sums=sums)
    assert torch.equal(y, hip.deconv4x4s2_fwd(x, wp))
    sums = sums.view(-1, 2 * cout).sum(0)
    if done:
        yf = y.float()
        s1, s2 = yf.sum(dim=(0, 2, 3)), (yf * yf).sum(dim=(0, 2, 3))
        torch.testing.assert_close(sums[:cout], s1, rtol=2e-4, atol=2e-3 * float(yf.abs().max()) * (yf.numel() / cout) ** 0.5)
        torch.testing.assert_close(sums[cout:], s2, rtol=2e-4, atol=1e-3 * max(1.0, float(s2.max())))
    else:
        assert float(sums.abs().max()) == 0.0
    if geo[0] == 32 and h >= 16:
        assert done                         # the two large head layers of the bench configuration take the fused path
