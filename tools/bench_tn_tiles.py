"""Weight-gradient (TN) GEMM shapes of the head under the planner's tile choice against forced tiles (epi_gemm_tune: 1 = 128 x 128, 2 = 256 x 256).
python tools/bench_tn_tiles.py   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


def main():
    lib = hip.load()
    dev = torch.device("cuda:0")
    shapes = [("final dW", 131072, 1088, 256), ("l4-like", 2048, 2048, 512), ("l3-like", 8192, 1024, 256), ("l2-like", 32768, 512, 128)]
    for name, r, i, j in shapes:
        # operands rotate through a few sets so that nothing stays cache resident between launches
        nset = max(1, min(8, (1 << 30) // ((r * (i + j)) * 2)))
        a = [torch.randn(r, i, device=dev).to(torch.bfloat16) for _ in range(nset)]
        b = [torch.randn(r, j, device=dev).to(torch.bfloat16) for _ in range(nset)]
        k = [0]

        def run():
            k[0] += 1
            return hip.gemm_tn_bf16(a[k[0] % nset], b[k[0] % nset])
        for tile, label in ((0, "planner"), (1, "128x128"), (2, "256x256")):
            lib.epi_gemm_tune(tile, -1)
            plan = hip.gemm_tn_plan(r, i, j, 1)
            us = timed(run)
            print("%-9s R %6d I %4d J %4d  %-8s cfg %d tiles %3d splits %3d   %7.1f us  %6.1f TF" % (name, r, i, j, label, plan["cfg"], plan["tiles"], plan["nsplit"], us,
                                                                                                       2.0 * r * i * j / us * 1e-6))
        lib.epi_gemm_tune(0, -1)
    # the deconvolution head's weight gradients (gathered B operand): x [B, Cin, H, W], dy [B, Cout, 2H, 2W]
    for name, b_, h, cin, cout in (("deconv1 dW", 32, 8, 2048, 256), ("deconv2 dW", 32, 16, 256, 256), ("deconv3 dW", 32, 32, 256, 256)):
        x = torch.randn(b_, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(b_, cout, 2 * h, 2 * h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        for tile, label in ((0, "planner"), (1, "128x128"), (2, "256x256")):
            lib.epi_gemm_tune(tile, -1)
            plan = hip.gemm_tn_plan(b_ * h * h, cin, cout, 16)
            us = timed(lambda: hip.deconv4x4s2_bwd_weight(x, dy, torch.bfloat16))
            print("%-10s R %6d I %4d J %5d  %-8s cfg %d tiles %3d splits %3d   %7.1f us  %6.1f TF" % (name, b_ * h * h, cin, 16 * cout, label, plan["cfg"], plan["tiles"], plan["nsplit"],
                                                                                                        us, 2.0 * b_ * h * h * cin * 16 * cout / us * 1e-6))
        lib.epi_gemm_tune(0, -1)


if __name__ == "__main__":
    main()
