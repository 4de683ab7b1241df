"""Host enqueue time and GPU time of the forward / backward / optimizer phases of one training step (queue drained
between phases), with and without the gradient-bucket path.  python tools/phase_times.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0], "--no-cpu-baseline"]
    args = bench.parse_args()
    from epipolarpose_amd import distributed as epd
    from epipolarpose_amd import hip
    device = torch.device("cuda:0")
    hip.load()
    torch.backends.cudnn.benchmark = True
    cfg, model, criterion, optimizer, images, label, weight, meta, scenes = bench.build_problem(args, device, 0)
    for mode in ("plain", "buckets"):
        sync = epd.BucketedGradSync(model, optimizer=optimizer) if mode == "buckets" else None
        acc = {}
        for it in range(8):
            def phase(name, fn):
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                out = fn()
                e1.record()
                host = (time.perf_counter() - t0) * 1e3
                torch.cuda.synchronize()
                if it >= 3:
                    a = acc.setdefault(name, [0.0, 0.0])
                    a[0] += host / 5
                    a[1] += e0.elapsed_time(e1) / 5
                return out

            def zero():
                if sync is not None:
                    sync.zero_grad()
                else:
                    optimizer.zero_grad(set_to_none=True)

            def fwd():
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    return criterion(model(images), label, weight)
            phase("zero_grad", zero)
            loss = phase("forward+criterion", fwd)
            phase("backward", loss.backward)
            if sync is not None:
                phase("finish", sync.finish)
            phase("optimizer", optimizer.step)
        print(mode, {k: ("host %.2f ms" % v[0], "gpu %.2f ms" % v[1]) for k, v in acc.items()}, flush=True)
        if sync is not None:
            for h in sync._hooks:
                h.remove()


if __name__ == "__main__":
    main()
