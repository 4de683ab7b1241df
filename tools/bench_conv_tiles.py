"""Backbone convolutions that are far from their bound (the stride-2 3x3 layers, layer 4) under the planner's tile choice against forced tiles
(epi_gemm_tune: 1 = 128 x 128, 2 = 256 x 256, 3 = 64 x 128, 4 = 64 x 64), forward (with BatchNorm sums) and backward-data.  python tools/bench_conv_tiles.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402
from bench_tn_tiles import timed  # noqa: E402


def main():
    lib = hip.load()
    dev = torch.device("cuda:0")
    b = 32
    layers = [("l2.0.c2 s2", 128, 128, 3, 2, 64), ("l3.0.c2 s2", 256, 256, 3, 2, 32), ("l4.0.c2 s2", 512, 512, 3, 2, 16), ("l4.c1", 2048, 512, 1, 1, 8), ("l4.c2", 512, 512, 3, 1, 8),
              ("l4.c3", 512, 2048, 1, 1, 8), ("l3.c2", 256, 256, 3, 1, 16), ("l3.c1", 1024, 256, 1, 1, 16)]
    for name, cin, cout, k, s, h in layers:
        pad = k // 2
        ho = (h + 2 * pad - k) // s + 1
        nset = 4
        xs = [torch.randn(b, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nset)]
        dys = [torch.randn(b, cout, ho, ho, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(nset)]
        w = (torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = hip.conv2d_pack_weight_bwd(w, s, pad)
        sums = torch.zeros(hip.load().epi_bn_sum_copies(cout) * 2 * cout, device=dev)
        flop = 2.0 * b * ho * ho * cout * cin * k * k
        i = [0]

        def fwd():
            i[0] += 1
            return hip.conv2d_fwd(xs[i[0] % nset], w, s, pad, bn_sums=sums)

        def bwd():
            i[0] += 1
            return hip.conv2d_bwd_data(dys[i[0] % nset], wb, (b, cin, h, h), k, s, pad)
        for tile, label in ((0, "planner"), (1, "128x128"), (2, "256x256"), (3, "64x128"), (4, "64x64")):
            lib.epi_gemm_tune(tile, -1)
            try:
                f, d = timed(fwd, 30), timed(bwd, 30)
                print("%-11s %4d->%4d k%d s%d H%2d  %-8s fwd %6.1f us %5.0f TF   bwd-data %6.1f us %5.0f TF" % (name, cin, cout, k, s, h, label, f, flop / f * 1e-6, d, flop / d * 1e-6))
            except RuntimeError as e:
                print("%-11s %-8s %s" % (name, label, e))
        lib.epi_gemm_tune(0, -1)


if __name__ == "__main__":
    main()
