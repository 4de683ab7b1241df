"""GPU parity: soft-argmax / joint loss / arg-max HIP kernels (through the C ABI) vs. the oracle and the
reference-generated golden vectors.  Tolerances: soft-argmax coordinates <= 1e-5 abs
(fp32), loss rel 1e-5, arg-max indices exact."""
import zlib

import numpy as np
import pytest
import torch

from det_weights import seeded_array
from make_golden_cases import dlogits_stride, INTEGRAL_CASES
from oracle import inference as o_inf
from oracle import integral as o_int

pytestmark = pytest.mark.gpu

COORD_ATOL = 1e-5
KIND_CLASS = {"l1": "L1JointLocationLoss", "smoothl1": "SmoothL1JointLocationLoss", "l2": "L2JointLocationLoss"}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU (run through gpurun)"
    from epipolarpose_amd import hip
    hip.load()
    return torch.device("cuda:0")


def case_logits(g, name, b, j, d, h, w, sc):
    if name + "/logits" in g:
        return g[name + "/logits"]
    logits = seeded_array("logits/" + name, (b, j * d, h, w), scale=sc)
    assert np.uint32(zlib.crc32(logits.tobytes())) == g[name + "/logits_crc"]
    return logits


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("case", INTEGRAL_CASES, ids=[c[0] for c in INTEGRAL_CASES])
def test_golden_losses_and_gradients(golden, dev, case, layout):
    from epipolarpose_amd.core import integral_loss as il
    g = golden("integral")
    name, b, j, d, h, w, sc = case
    logits = case_logits(g, *case)
    gt = torch.from_numpy(g[name + "/gt"]).to(dev)
    wt = torch.from_numpy(g[name + "/wt"]).to(dev)
    base = torch.from_numpy(logits).to(dev)
    if layout == "nhwc":
        base = base.contiguous(memory_format=torch.channels_last)
    xyz = il.softmax_integral_tensor(base, j, True, w, h, d)
    # against the reference's code on float64 tensors at the stated tolerance; against its float32 run at that tolerance or the float32 run's own
    # distance from the float64 one (1.8e-5 at the configuration's shape "cfg", 2e-6 below), whichever is larger
    np.testing.assert_allclose(xyz.cpu().numpy(), g[name + "/xyz64"], atol=COORD_ATOL)
    np.testing.assert_allclose(xyz.cpu().numpy(), g[name + "/xyz"], atol=max(COORD_ATOL, 1.5 * float(np.abs(g[name + "/xyz"] - g[name + "/xyz64"]).max())))
    for kind, cls in KIND_CLASS.items():
        for norm in (False, True):
            key = "%s/%s/norm%d" % (name, kind, int(norm))
            t = base.clone().requires_grad_(True)
            loss = getattr(il, cls)(num_joints=j, norm=norm)(t, gt, wt)
            loss.backward()
            np.testing.assert_allclose(loss.item(), g[key + "/loss"], rtol=2e-5, atol=1e-7)
            ref = g[key + "/dlogits"]
            got = t.grad.contiguous().cpu().numpy()       # logical NCHW order either way
            if ref.ndim == 1:
                got = got.reshape(-1)[::dlogits_stride(got.size)]
            scale = np.abs(ref).max()
            own = 0.0
            if key + "/loss64" in g:
                np.testing.assert_allclose(loss.item(), g[key + "/loss64"], rtol=2e-5, atol=1e-7)
                np.testing.assert_allclose(got, g[key + "/dlogits64"], atol=3e-5 * scale + 1e-8)
                own = float(np.abs(ref - g[key + "/dlogits64"]).max())
            np.testing.assert_allclose(got, ref, atol=1.5 * own + 3e-5 * scale + 1e-8)
    if name + "/decode256" in g:
        dec = il.get_joint_location_result(256, 256, base)
        own_xyz = float(np.abs(g[name + "/xyz"] - g[name + "/xyz64"]).max())
        np.testing.assert_allclose(dec, g[name + "/decode256"], atol=256 * max(COORD_ATOL, 1.5 * own_xyz))


@pytest.mark.parametrize("shape", [(2, 3, 5, 7, 9), (1, 2, 6, 10, 12), (3, 1, 1, 1, 1), (1, 1, 64, 64, 64), (2, 17, 8, 24, 20)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_ragged_shapes_vs_oracle(dev, shape, layout):
    """Odd extents take the scalar fallback; 1x1x1 rows and single rows are the degenerate edges."""
    from epipolarpose_amd import hip
    b, j, d, h, w = shape
    logits = seeded_array("ragged/%s" % (shape,), (b, j * d, h, w), scale=3.0)
    t = torch.from_numpy(logits).to(dev)
    if layout == "nhwc":
        t = t.contiguous(memory_format=torch.channels_last)
    xyz, rmax, rsum = hip.softargmax3d_fwd(t, j)
    np.testing.assert_allclose(xyz.cpu().numpy(), o_int.softmax_integral(logits, j, w, h, d), atol=COORD_ATOL)
    np.testing.assert_allclose(rmax.cpu().numpy(), logits.reshape(b * j, -1).max(axis=1), rtol=0, atol=0)
    gx = seeded_array("ragged/g", (b, 3 * j))
    dl = hip.softargmax3d_bwd(t, j, rmax, rsum, xyz, torch.from_numpy(gx).to(dev))
    ref = o_int.softmax_integral_backward(logits, j, w, h, d, gx)
    np.testing.assert_allclose(dl.contiguous().cpu().numpy(), ref, atol=3e-5 * np.abs(ref).max() + 1e-9)


def test_bf16_logits(dev):
    from epipolarpose_amd import hip
    b, j, d, h, w = 2, 5, 16, 16, 16
    logits32 = seeded_array("bf16", (b, j * d, h, w), scale=4.0)
    tb = torch.from_numpy(logits32).to(dev).to(torch.bfloat16)
    exact = tb.float().cpu().numpy()                    # the values the kernel actually sees
    for t in (tb, tb.contiguous(memory_format=torch.channels_last)):
        xyz, rmax, rsum = hip.softargmax3d_fwd(t, j)
        np.testing.assert_allclose(xyz.cpu().numpy(), o_int.softmax_integral(exact, j, w, h, d), atol=COORD_ATOL)
        gx = seeded_array("bf16/g", (b, 3 * j))
        dl = hip.softargmax3d_bwd(t, j, rmax, rsum, xyz, torch.from_numpy(gx).to(dev))
        assert dl.dtype == torch.bfloat16
        ref = o_int.softmax_integral_backward(exact, j, w, h, d, gx)
        np.testing.assert_allclose(dl.float().contiguous().cpu().numpy(), ref, atol=1e-2 * np.abs(ref).max())


@pytest.mark.parametrize("shape", [(2, 5, 16, 16, 16), (3, 4, 64, 8, 8), (2, 3, 32, 6, 10), (32, 17, 64, 64, 64)], ids=["d16", "d64", "d32_ragged", "bench"])
@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_backward_delivers_the_column_sums_of_the_gradient_it_writes(dev, shape, dtype):
    """epi_softargmax3d_bwd_colsums on channels-last logits: the per-channel sums (the bias gradient of the final 1x1 convolution,
    pose3d_resnet.py:116-122) equal the sums of the gradient AS STORED -- fp32 accumulation of the same numbers in another order, so the bar is
    1e-5 of the sum of magnitudes -- and the gradient itself is bit-identical to the plain entry point's."""
    from epipolarpose_amd import hip
    b, j, d, h, w = shape
    gen = torch.Generator(device="cpu").manual_seed(b * 131 + d)
    logits = (torch.randn((b, j * d, h, w), generator=gen) * 3).to(dev)
    if dtype == "bf16":
        logits = logits.to(torch.bfloat16)
    logits = logits.contiguous(memory_format=torch.channels_last)
    xyz, rmax, rsum = hip.softargmax3d_fwd(logits, j)
    g = torch.randn((b, 3 * j), generator=gen).to(dev)
    scale = torch.tensor([0.37], device=dev)
    plain = hip.softargmax3d_bwd(logits, j, rmax, rsum, xyz, g, scale)
    sums = torch.zeros(j * d, dtype=torch.float32, device=dev)
    dl, delivered = hip.softargmax3d_bwd(logits, j, rmax, rsum, xyz, g, scale, col_sums=sums)
    assert delivered, "channels-last logits with a power-of-two depth extent: the sums must come from the kernel"
    assert torch.equal(dl, plain) and dl.is_contiguous(memory_format=torch.channels_last)
    want = dl.double().sum(dim=(0, 2, 3))
    mag = dl.double().abs().sum(dim=(0, 2, 3))
    assert ((sums.double() - want).abs() <= 1e-5 * mag + 1e-12).all(), float(((sums.double() - want).abs() / (mag + 1e-30)).max())
    # NCHW logits / deterministic mode: not delivered, buffer untouched
    sums2 = torch.zeros_like(sums)
    _, delivered2 = hip.softargmax3d_bwd(logits.contiguous(), j, rmax, rsum, xyz, g, scale, col_sums=sums2)
    assert not delivered2 and not sums2.any()
    hip.set_deterministic(True)
    try:
        _, delivered3 = hip.softargmax3d_bwd(logits, j, rmax, rsum, xyz, g, scale, col_sums=sums2)
    finally:
        hip.set_deterministic(False)
    assert not delivered3 and not sums2.any()


def test_bench_shape_criterion_elementwise_vs_oracle(dev):
    """The criterion exactly as the bench step runs it -- B = 32, J = 17, D = 64, 64 x 64 heat-map, bf16 channels-last logits, SmoothL1 (BASELINE.json
    configs[1]; the softargmax_partial / softargmax_bwd_kernel<bf16, 8, NHWC> instances) -- against the oracle ON THE SAME bf16-rounded logits: every
    coordinate, the loss, and the gradient of three whole images element by element (integral_loss.py:49-86,33-47,140-160).  The oracle runs image by
    image (the soft-argmax is separable over the batch; the loss's 1 / B is applied here)."""
    from epipolarpose_amd.core import integral_loss as il
    b, j, d, hm = 32, 17, 64, 64
    gen = torch.Generator(device="cpu").manual_seed(21)
    logits = (torch.randn((b, j * d, hm, hm), generator=gen) * 3).to(torch.bfloat16)
    gt = (torch.rand((b, 3 * j), generator=gen) - 0.5) * 0.6
    wt = (torch.rand((b, 3 * j), generator=gen) > 0.1).float()
    t = logits.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xyz = il.softmax_integral_tensor(t.detach(), j, True, hm, hm, d)
    loss = il.SmoothL1JointLocationLoss(num_joints=j)(t, gt.to(dev), wt.to(dev))
    loss.backward()
    assert t.grad.dtype == torch.bfloat16 and t.grad.is_contiguous(memory_format=torch.channels_last)
    pred = np.concatenate([o_int.softmax_integral(logits[i:i + 1].float().numpy(), j, hm, hm, d) for i in range(b)])
    np.testing.assert_allclose(xyz.cpu().numpy(), pred, atol=COORD_ATOL)
    np.testing.assert_allclose(loss.item(), o_int.joint_loss(pred, gt.numpy(), wt.numpy(), "smoothl1"), rtol=1e-5, atol=1e-8)
    gxyz = o_int.joint_loss_grad(pred, gt.numpy(), wt.numpy(), "smoothl1")            # [B, 3J], carries the 1 / B
    for i in (0, 13, 31):
        ref = o_int.softmax_integral_backward(logits[i:i + 1].float().numpy(), j, hm, hm, d, gxyz[i:i + 1])[0]
        got = t.grad[i].float().cpu().numpy()                                        # logical [C, H, W] order
        # stored in bf16: half a unit in the last place of each element (2^-9 relative) on top of the fp32 arithmetic (1e-5 of the row's largest element)
        tol = 2.0 ** -8 * np.abs(ref) + 2e-5 * np.abs(ref).max()
        bad = np.abs(got - ref) > tol
        assert not bad.any(), (i, int(bad.sum()), float(np.abs(got - ref).max()), float(np.abs(ref).max()))


def test_full_size_properties(dev):
    """BASELINE size (B=32, J=17, D=H=W=64): shift invariance, peak recovery, gradient rows sum to zero."""
    from epipolarpose_amd import hip
    from epipolarpose_amd.synthetic import SyntheticScenes
    b, j, d = 32, 17, 64
    gen = torch.Generator(device="cpu").manual_seed(0)
    logits = torch.randn((b, j * d, 64, 64), generator=gen).to(dev)
    xyz, rmax, rsum = hip.softargmax3d_fwd(logits, j)
    xyz2, _, _ = hip.softargmax3d_fwd(logits + 7.5, j)
    assert (xyz - xyz2).abs().max().item() <= 2e-6                      # softmax(x + c) == softmax(x)
    xyz3, _, _ = hip.softargmax3d_fwd(logits.contiguous(memory_format=torch.channels_last), j)
    assert (xyz - xyz3).abs().max().item() <= 2e-6                      # layout independent
    # fp32 torch restatement on the device for the full tensor (plain PyTorch ops)
    p = torch.softmax(logits.reshape(b, j, -1).double(), dim=2).reshape(b, j, d, 64, 64)
    ar = torch.arange(64, device=dev, dtype=torch.float64)
    ex = (p.sum(dim=(2, 3)) * ar).sum(-1) / 64 - 0.5
    ey = (p.sum(dim=(2, 4)) * ar).sum(-1) / 64 - 0.5
    ez = (p.sum(dim=(3, 4)) * ar).sum(-1) / 64 - 0.5
    ref = torch.stack([ex, ey, ez], dim=2).reshape(b, 3 * j)
    assert (xyz.double() - ref).abs().max().item() <= COORD_ATOL
    g = torch.randn((b, 3 * j), generator=gen).to(dev)
    dl = hip.softargmax3d_bwd(logits, j, rmax, rsum, xyz, g)
    rows = dl.reshape(b * j, -1).double().sum(dim=1)
    assert rows.abs().max().item() <= 1e-5                              # softmax Jacobian annihilates constants
    # peaked logits: soft-argmax agrees with the hard arg-max voxel within half a voxel, arg-max exact
    sc = SyntheticScenes(n_group=2, n_view=4, num_joints=j, seed=5)
    peaked = torch.from_numpy(sc.peaked_logits(depth=d, hm=64)).to(dev)
    xyz_p, _, _ = hip.softargmax3d_fwd(peaked, j)
    idx, val = hip.argmax_rows(peaked.reshape(8 * j, -1))
    ref_idx = peaked.reshape(8 * j, -1).argmax(dim=1)
    assert torch.equal(idx, ref_idx)
    vox = torch.stack([(idx % 64), (idx // 64) % 64, idx // 4096], dim=1).float().reshape(8, j, 3)
    soft = (xyz_p.reshape(8, j, 3) + 0.5) * 64
    inside = (torch.from_numpy(np.abs(sc.label.reshape(8, j, 3))).to(dev) < 0.4).all(dim=2)
    assert ((soft - vox).abs().max(dim=2).values[inside] < 0.75).all()


def test_argmax_golden_and_ties(golden, dev):
    from epipolarpose_amd.core import inference
    g = golden("maxpreds")
    preds, maxvals = inference.get_max_preds(g["heatmaps"])
    np.testing.assert_array_equal(preds, g["preds"])
    np.testing.assert_array_equal(maxvals, g["maxvals"])
    # long rows with planted ties, NaN and -inf: must match numpy.argmax exactly
    rng = np.random.default_rng(9)
    x = rng.standard_normal((7, 70001)).astype(np.float32)
    x[0, [5, 60000]] = 99.0
    x[1, 69999] = np.nan
    x[1, 123] = np.nan
    x[2, :] = -np.inf
    x[3, :] = 1.5
    from epipolarpose_amd import hip
    idx, val = hip.argmax_rows(torch.from_numpy(x).to(dev))
    np.testing.assert_array_equal(idx.cpu().numpy(), np.argmax(x, axis=1))
    oi, ov = o_inf.argmax_rows(x[None])
    np.testing.assert_array_equal(val.cpu().numpy(), ov[0])


def test_error_behaviour(dev):
    from epipolarpose_amd import hip
    from epipolarpose_amd.core import integral_loss as il
    with pytest.raises(RuntimeError):
        hip.softargmax3d_fwd(torch.zeros(1, 8, 2, 2), 2)               # CPU tensor: no fallback
    with pytest.raises(TypeError):
        hip.softargmax3d_fwd(torch.zeros(1, 8, 2, 2, device=dev, dtype=torch.float16), 2)
    crit = il.L1JointLocationLoss(num_joints=3)
    with pytest.raises(ValueError):
        crit(torch.zeros(1, 8, 2, 2, device=dev), torch.zeros(1, 9, device=dev), torch.ones(1, 9, device=dev))
    with pytest.raises(AssertionError):
        crit(torch.zeros(1, 9, 2, 2, device=dev), torch.zeros(1, 9, device=dev, requires_grad=True),
             torch.ones(1, 9, device=dev))
