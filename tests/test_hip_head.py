"""GPU parity: MFMA head kernels (through the C ABI) vs. plain PyTorch fp32 references of the same ops on the
same bf16-rounded inputs.  Tolerance: bf16 output rounding (2^-8 relative) + fp32 accumulation-order noise."""
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


def close(got, ref, rel=1.2e-2):
    err = (got.float() - ref).abs().max().item()
    tol = rel * ref.abs().max().item()
    assert err <= tol, "max err %.4g > tol %.4g" % (err, tol)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 1088, 256), (1000, 200, 128), (4096, 256, 1088), (77, 36, 192)])
def test_gemm_vs_torch(dev, m, n, k):
    from epipolarpose_amd import hip
    a = rnd((m, k), dev, 1).to(torch.bfloat16)
    bt = rnd((n, k), dev, 2).to(torch.bfloat16)
    ref = a.float() @ bt.float().t()          # fp32 matmul of the bf16-rounded values (asymmetric operands)
    close(hip.gemm_bf16(a, bt), ref)
    bias = rnd((n,), dev, 3)
    c32 = hip.gemm_bf16(a, bt, bias=bias, out_dtype=torch.float32)
    close(c32, ref + bias, rel=2e-5)


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric B catches a transposed C write (HIP guide section 3)."""
    from epipolarpose_amd import hip
    a = torch.eye(128, 128, device=dev, dtype=torch.bfloat16)
    bt = (torch.arange(256 * 128, device=dev).reshape(256, 128) % 251).to(torch.bfloat16)
    c = hip.gemm_bf16(a, bt, out_dtype=torch.float32)
    assert torch.equal(c, bt.float().t().contiguous())


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 8, 8, 128, 64), (3, 5, 7, 64, 32), (1, 16, 16, 256, 256), (2, 4, 4, 2048, 256)])
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
