#!/usr/bin/env python
"""Launch every hand-written kernel a few times on bench-sized inputs (B=32) -- the target of the rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE are collected in separate runs: tools/gpu_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402
from epipolarpose_amd.models.fused import FusedBatchNormAct  # noqa: E402

DEV = torch.device("cuda:0")
REPS = 3


def cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def main():
    hip.load()
    b, j, d, hm = 32, 17, 64, 64
    logits = cl(torch.randn(b, j * d, hm, hm, device=DEV).to(torch.bfloat16))
    g = torch.randn(b, 3 * j, device=DEV)
    for _ in range(REPS):
        xyz, rmax, rsum = hip.softargmax3d_fwd(logits, j)
        hip.softargmax3d_bwd(logits, j, rmax, rsum, xyz, g)
    for h, cin in ((8, 2048), (16, 256), (32, 256)):
        x = cl(torch.randn(b, cin, h, h, device=DEV).to(torch.bfloat16))
        w = torch.randn(cin, 256, 4, 4, device=DEV) * 0.02
        wp, wb = hip.deconv_pack_weight(w)
        dy = cl(torch.randn(b, 256, 2 * h, 2 * h, device=DEV).to(torch.bfloat16))
        for _ in range(REPS):
            hip.deconv4x4s2_fwd(x, wp)
            hip.deconv4x4s2_bwd_data(dy, wb)
            hip.deconv4x4s2_bwd_weight(x, dy)
    a = torch.randn(b * hm * hm, 256, device=DEV).to(torch.bfloat16)
    wt = torch.randn(j * d, 256, device=DEV).to(torch.bfloat16)
    dl = torch.randn(b * hm * hm, j * d, device=DEV).to(torch.bfloat16)
    wtt = wt.t().contiguous()
    for _ in range(REPS):
        hip.gemm_bf16(a, wt)
        hip.gemm_bf16(dl, wtt)
        hip.gemm_tn_bf16(dl, a)
    for shape, res in (((32, 64, 128, 128), False), ((32, 256, 64, 64), True), ((32, 1024, 16, 16), True)):
        x = cl(torch.randn(shape, device=DEV).to(torch.bfloat16)).requires_grad_(True)
        r = cl(torch.randn(shape, device=DEV).to(torch.bfloat16)).requires_grad_(True) if res else None
        dy = cl(torch.randn(shape, device=DEV).to(torch.bfloat16))
        m = FusedBatchNormAct(shape[1]).to(DEV)
        for _ in range(REPS):
            y = m(x, residual=r)
            y.backward(dy)
    # backbone convolutions (round 2): one 1x1, one 3x3 and one stride-2 layer of ResNet-50, forward / backward-data / backward-weight
    for cin, cout, k, s, h in ((256, 64, 1, 1, 64), (64, 256, 1, 1, 64), (128, 128, 3, 1, 32), (256, 256, 3, 2, 32), (1024, 256, 1, 1, 16), (512, 2048, 1, 1, 8)):
        pad = k // 2
        x = cl(torch.randn(b, cin, h, h, device=DEV).to(torch.bfloat16))
        w = cl((torch.randn(cout, cin, k, k, device=DEV) * 0.05).to(torch.bfloat16))
        y = hip.conv2d_fwd(x, w, s, pad)
        dy = cl(torch.randn_like(y.float()).to(torch.bfloat16))
        wb = hip.conv2d_pack_weight_bwd(w, s, pad)
        for _ in range(REPS):
            hip.conv2d_fwd(x, w, s, pad)
            hip.conv2d_bwd_data(dy, wb, tuple(x.shape), k, s, pad)
            hip.conv2d_bwd_weight(x, dy, k, s, pad, dtype=torch.bfloat16)
    x = cl(torch.randn(b, 64, 128, 128, device=DEV).clamp_min(0).to(torch.bfloat16))
    for _ in range(REPS):
        y, pos = hip.maxpool3x3s2_fwd(x)
        hip.maxpool3x3s2_bwd(y, pos, (128, 128))
    torch.cuda.synchronize()
    print("pmc target done")


if __name__ == "__main__":
    main()
