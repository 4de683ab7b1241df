#!/usr/bin/env python
"""Per-layer timing of the backbone convolutions (ResNet-50 @ 256x256, batch 32): the hand-written implicit-GEMM kernels
(epi_conv2d_fwd / _bwd_data / _bwd_weight) beside MIOpen (torch, bf16 channels_last, cudnn.benchmark) for every distinct
convolution shape, with the roofline bound of each (max of MFMA time at 2.5 PF and HBM time at 8 TB/s on the algorithmic bytes).
    python tools/bench_conv.py [--batch 32] [--iters 20] > gpurun_out/conv_layers.txt"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402


def r50_shapes(image=256):
    """(name, count, Cin, Cout, k, stride, H_in) for every distinct conv of ResNet-50 behind the stem."""
    out, h, inpl = {}, image // 4, 64
    for st, (pl, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), 1):
        for u in range(n):
            s = 2 if (u == 0 and st > 1) else 1
            for name, cin, cout, k, ss, hh in (("c1", inpl, pl, 1, 1, h), ("c2", pl, pl, 3, s, h), ("c3", pl, pl * 4, 1, 1, h // s)):
                key = (cin, cout, k, ss, hh)
                out.setdefault(key, ["l%d.%s" % (st, name), 0])[1] += 1
            if s != 1 or inpl != pl * 4:
                key = (inpl, pl * 4, 1, s, h)
                out.setdefault(key, ["l%d.ds" % st, 0])[1] += 1
            inpl, h = pl * 4, h // s
    return [(v[0], v[1]) + k for k, v in out.items()]


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--image", type=int, default=256)
    ap.add_argument("--no-miopen", action="store_true", help="skip the library timings (quick A/B of our kernels)")
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda:0")
    hip.load()
    b = args.batch
    tot = {"ours": [0.0, 0.0, 0.0], "miopen": [0.0, 0.0, 0.0], "bound": [0.0, 0.0, 0.0], "flops": 0.0}
    print("# ResNet-50 convolutions behind the stem, batch %d, %dx%d input; times in us per launch; TF = algorithmic TFLOP/s" % (b, args.image, args.image))
    print("%-7s %2s %5s %5s %1s %1s %3s | %7s | %-23s | %-23s | %-23s" % ("layer", "n", "Cin", "Cout", "k", "s", "H", "GFLOP", "fwd ours/miopen/bound", "dgrad ours/miopen/bound", "wgrad ours/miopen/bound"))
    for name, count, cin, cout, k, s, h in r50_shapes(args.image):
        pad = k // 2
        x = torch.randn(b, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, device=dev) * (2.0 / (cin * k * k)) ** 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        y = hip.conv2d_fwd(x, w, s, pad)
        dy = torch.randn_like(y.float()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = hip.conv2d_pack_weight_bwd(w, s, pad)
        ho = y.shape[2]
        flops = 2.0 * b * ho * ho * cout * cin * k * k
        bytes_f = 2.0 * (x.numel() + y.numel() + w.numel())
        bound = max(flops / 2.5e15, bytes_f / 8e12) * 1e6
        ours = [timeit(lambda: hip.conv2d_fwd(x, w, s, pad), args.iters),
                timeit(lambda: hip.conv2d_bwd_data(dy, wb, tuple(x.shape), k, s, pad), args.iters),
                timeit(lambda: hip.conv2d_bwd_weight(x, dy, k, s, pad, dtype=torch.bfloat16), args.iters)]
        mi = [0.0, 0.0, 0.0] if args.no_miopen else [timeit(lambda: F.conv2d(x, w, stride=s, padding=pad), args.iters),
              timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1, (True, False, False)), args.iters),
              timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (s, s), (pad, pad), (1, 1), False, (0, 0), 1, (False, True, False)), args.iters)]
        cells = ["%6.1f/%6.1f/%5.1f" % (ours[i], mi[i], bound) for i in range(3)]
        print("%-7s %2d %5d %5d %1d %1d %3d | %7.2f | %s | %s | %s   fwd %4.0f TF (%.2f of bound)" % (name, count, cin, cout, k, s, h, flops / 1e9, cells[0], cells[1], cells[2], flops / ours[0] / 1e6, bound / ours[0]))
        for i in range(3):
            tot["ours"][i] += count * ours[i]
            tot["miopen"][i] += count * mi[i]
            tot["bound"][i] += count * bound
        tot["flops"] += count * flops
        sys.stdout.flush()
    for kname in ("ours", "miopen", "bound"):
        t = tot[kname]
        print("# total %-6s fwd %7.1f us  dgrad %7.1f us  wgrad %7.1f us  sum %7.1f us  -> %6.1f TFLOP/s over 3 x %.1f GFLOP" %
              (kname, t[0], t[1], t[2], sum(t), 3 * tot["flops"] / sum(t) / 1e6, tot["flops"] / 1e9))


if __name__ == "__main__":
    main()
