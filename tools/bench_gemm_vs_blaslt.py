#!/usr/bin/env python
"""8192^3 / 4096^3 bf16: epi_gemm_bf16 against torch.matmul (hipBLASLt) on the SAME box, the same operands and the same timing loop, for several
operand fills -- the rate of a large GEMM on this chip depends on what the operands toggle."""
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip          # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def operands(kind, m, n, k):
    g = torch.Generator(device=DEV).manual_seed(7)
    if kind == "randn":
        a, bt = torch.randn(m, k, device=DEV, generator=g), torch.randn(n, k, device=DEV, generator=g)
    elif kind == "uniform":
        a, bt = torch.rand(m, k, device=DEV, generator=g) - 0.5, torch.rand(n, k, device=DEV, generator=g) - 0.5
    elif kind == "ternary":
        a = torch.randint(-1, 2, (m, k), device=DEV, generator=g).float()
        bt = torch.randint(-1, 2, (n, k), device=DEV, generator=g).float()
    elif kind == "network":          # relu(normal) activations x 0.05 * normal weights
        a, bt = torch.randn(m, k, device=DEV, generator=g).clamp_min(0), 0.05 * torch.randn(n, k, device=DEV, generator=g)
    else:
        a, bt = torch.ones(m, k, device=DEV), torch.ones(n, k, device=DEV)
    return a.to(torch.bfloat16), bt.to(torch.bfloat16)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for m, n, k in ((8192, 8192, 8192), (4096, 4096, 4096)):
        for kind in ("randn", "uniform", "network", "ternary", "ones"):
            a, bt = operands(kind, m, n, k)
            out = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
            out2 = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
            fl = 2.0 * m * n * k
            rows = []
            for rnd in range(3):
                t = timeit(lambda: hip.gemm_bf16(a, bt, out=out), reps)
                tt = timeit(lambda: torch.matmul(a, bt.t(), out=out2), reps)
                rows.append((fl / t / 1e9, fl / tt / 1e9))
            same = (out.float() - out2.float()).abs().max().item()
            print("gemm %5d^3 %-8s reps %3d  ours %s TF   torch(hipBLASLt) %s TF   max |ours - theirs| %.4g" % (
                m, kind, reps, " / ".join("%6.1f" % r[0] for r in rows), " / ".join("%6.1f" % r[1] for r in rows), same), flush=True)


if __name__ == "__main__":
    main()
