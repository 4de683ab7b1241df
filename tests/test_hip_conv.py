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


def test_conv2d_unsupported_geometry_is_refused():
    from epipolarpose_amd import hip
    dev = torch.device("cuda:0")
    x = torch.zeros(1, 3, 16, 16, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.zeros(64, 3, 7, 7, dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    with pytest.raises(RuntimeError, match="epi_conv2d_fwd"):
        hip.conv2d_fwd(x, w, 2, 3)                            # the 7x7 stem (49 taps, 3 channels) stays with the library
