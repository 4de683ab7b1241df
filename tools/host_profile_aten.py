"""Where the host's enqueue time goes, by ATen / autograd operator: torch.profiler (CPU activity only) over a few training steps into an empty queue.
python tools/host_profile_aten.py [steps]   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    sys.argv = [sys.argv[0], "--no-cpu-baseline"]
    args = bench.parse_args()
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.function import train_step
    device = torch.device("cuda:0")
    hip.load()
    cfg, model, criterion, optimizer, images, label, weight, meta, scenes = bench.build_problem(args, device, 0)
    for _ in range(5):
        train_step(model, criterion, optimizer, images, label, weight, meta=None, n_view=None, autocast=True)
    torch.cuda.synchronize()
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        for _ in range(steps):
            torch.cuda.synchronize()
            train_step(model, criterion, optimizer, images, label, weight, meta=None, n_view=None, autocast=True)
        torch.cuda.synchronize()
    print("# %d steps; times are totals over all steps (divide by %d)" % (steps, steps))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))


if __name__ == "__main__":
    main()
