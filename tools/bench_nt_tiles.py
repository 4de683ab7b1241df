"""Head GEMMs (final 1x1 convolution forward / backward-data, deconvolution forward / backward-data) under the planner's tile choice against forced tiles
(epi_gemm_tune: 1 = 128 x 128, 2 = 256 x 256, 3 = 64 x 128).  python tools/bench_nt_tiles.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402
from bench_tn_tiles import timed  # noqa: E402


def main():
    lib = hip.load()
    dev = torch.device("cuda:0")
    tiles = ((0, "planner"), (1, "128x128"), (2, "256x256"), (3, "64x128"))
    for name, m, n, k in (("final fwd", 131072, 1088, 256), ("final dX", 131072, 256, 1088)):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        bt = torch.randn(n, k, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        for tile, label in tiles:
            lib.epi_gemm_tune(tile, -1)
            try:
                us = timed(lambda: hip.gemm_bf16(a, bt, out=out))
                print("%-12s M %6d N %4d K %4d  %-8s %7.1f us  %6.1f TF" % (name, m, n, k, label, us, 2.0 * m * n * k / us * 1e-6))
            except RuntimeError as e:
                print("%-12s %-8s %s" % (name, label, e))
        lib.epi_gemm_tune(0, -1)
    for name, b, h, cin, cout in (("deconv1", 32, 8, 2048, 256), ("deconv2", 32, 16, 256, 256), ("deconv3", 32, 32, 256, 256)):
        x = torch.randn(b, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(b, cout, 2 * h, 2 * h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wp, wb = hip.deconv_pack_weight(torch.randn(cin, cout, 4, 4, device=dev) * (1.0 / cin) ** 0.5)
        flop = 2.0 * b * h * h * cin * 16 * cout
        for tile, label in tiles:
            lib.epi_gemm_tune(tile, -1)
            try:
                f = timed(lambda: hip.deconv4x4s2_fwd(x, wp))
                d = timed(lambda: hip.deconv4x4s2_bwd_data(dy, wb))
                print("%-12s B %2d H %2d Cin %4d  %-8s fwd %7.1f us %6.1f TF   bwd-data %7.1f us %6.1f TF" % (name, b, h, cin, label, f, flop / f * 1e-6, d, flop / d * 1e-6))
            except RuntimeError as e:
                print("%-12s %-8s %s" % (name, label, e))
        lib.epi_gemm_tune(0, -1)


if __name__ == "__main__":
    main()
