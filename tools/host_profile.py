"""Host-side (Python / dispatcher) cost of one training step: cProfile over a few steps, GPU queue drained before each so
that only enqueue work is measured.  python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    sys.argv = [sys.argv[0], "--no-cpu-baseline"]
    args = bench.parse_args()
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.function import train_step
    device = torch.device("cuda:0")
    hip.load()
    torch.backends.cudnn.benchmark = True
    cfg, model, criterion, optimizer, images, label, weight, meta, scenes = bench.build_problem(args, device, 0)

    class Ctx:
        @staticmethod
        def step():
            return train_step(model, criterion, optimizer, images, label, weight, meta=None, n_view=None, autocast=True)
    ctx = Ctx()
    for _ in range(5):
        ctx.step()
    torch.cuda.synchronize()
    t = []
    for _ in range(steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.step()
        t.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
    print("host enqueue ms/step (empty queue):", ["%.2f" % v for v in t])
    pr = cProfile.Profile()
    for _ in range(steps):
        torch.cuda.synchronize()
        pr.enable()
        ctx.step()
        pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumtime").print_stats(30)


if __name__ == "__main__":
    main()
