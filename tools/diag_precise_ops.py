"""Diagnostic (GPU): every operator of the fp32-grade mode at the BENCH sizes (batch 32) against stock fp32 PyTorch (TF32 off) -- which one
breaks between batch 8 and batch 32?  Prints max-relative errors of y, dx, dw (db)."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def main():
    from epipolarpose_amd.models import precise
    torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(1)
    for b in (8, 32):
        print("== batch %d" % b, flush=True)
        # (kind, Cin, Cout, k, stride, H)
        cases = [("conv", 64, 64, 1, 1, 64), ("conv", 64, 64, 3, 1, 64), ("conv", 64, 256, 1, 1, 64), ("conv", 256, 64, 1, 1, 64), ("conv", 256, 512, 1, 2, 64),
                 ("conv", 128, 128, 3, 2, 64), ("conv", 512, 128, 1, 1, 32), ("conv", 256, 256, 3, 1, 16), ("conv", 1024, 256, 1, 1, 16), ("conv", 512, 512, 3, 1, 8),
                 ("conv", 2048, 512, 1, 1, 8), ("deconv", 2048, 256, 4, 2, 8), ("deconv", 256, 256, 4, 2, 16), ("deconv", 256, 256, 4, 2, 32), ("final", 256, 1088, 1, 1, 64)]
        for kind, cin, cout, k, stride, h in cases:
            x = torch.randn((b, cin, h, h), generator=gen).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            if kind == "conv":
                w = (torch.randn((cout, cin, k, k), generator=gen) * (2.0 / (cin * k * k)) ** 0.5).to(dev).requires_grad_(True)
                y = precise.conv2d(x, w, stride, k // 2)
                yr = F.conv2d(x, w, stride=stride, padding=k // 2)
                bias = None
            elif kind == "deconv":
                w = (torch.randn((cin, cout, 4, 4), generator=gen) * (1.0 / (cin * 4)) ** 0.5).to(dev).requires_grad_(True)
                y = precise.deconv4x4s2(x, w)
                yr = F.conv_transpose2d(x, w, stride=2, padding=1)
                bias = None
            else:
                w = (torch.randn((cout, cin, 1, 1), generator=gen) * (1.0 / cin) ** 0.5).to(dev).requires_grad_(True)
                bias = torch.randn(cout, generator=gen).to(dev).requires_grad_(True)
                y = precise.conv1x1_bias(x, w, bias)
                yr = F.conv2d(x, w, bias)
            dy = torch.randn(y.shape, generator=gen).to(dev).contiguous(memory_format=torch.channels_last)
            gs = torch.autograd.grad(y, [x, w] + ([bias] if bias is not None else []), dy)
            gr = torch.autograd.grad(yr, [x, w] + ([bias] if bias is not None else []), dy)
            torch.cuda.synchronize()
            print("%-7s %4d->%4d k%d s%d H%-3d  y %.2e  dx %.2e  dw %.2e%s" % (kind, cin, cout, k, stride, h, rel(y, yr), rel(gs[0], gr[0]), rel(gs[1], gr[1]),
                                                                            "  db %.2e" % rel(gs[2], gr[2]) if bias is not None else ""), flush=True)
            del x, w, y, yr, dy, gs, gr


if __name__ == "__main__":
    main()
