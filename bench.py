#!/usr/bin/env python
"""Benchmark of the EpipolarPose training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank/GPU)

A "step" is one full optimisation step (forward, soft-argmax criterion, backward, gradient all-reduce when N > 1,
Adam) of the ResNet-50 volumetric-heat-map network on one synthetic 4-view 256x256 batch that is already resident
in HBM.  Default workload = BASELINE.json configs[1] (fully-supervised SmoothL1 loss, 32 images = 8 groups x 4 views
per GPU); ``--workload ss`` = configs[2] (pseudo labels from 4-view triangulation inside the step).
Rank 0 prints ONE JSON line (see the task contract) carrying ``roofline`` and ``cpu_baseline`` objects.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("fs", "ss"), default="fs")
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (groups x 4 views)")
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--image", type=int, default=256)
    ap.add_argument("--joints", type=int, default=17)
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--fp32", action="store_true", help="disable bf16 autocast (diagnostic; not the bench line)")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the step as one hipGraph (N=1 only); 0: eager (default: "
                    "the step is GPU-bound and hipGraph replay measured 6 %% slower than eager launches on ROCm 7.2)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL); 'gloo' + --same-device lets "
                    "the N>1 code path be exercised on a single-GPU box")
    ap.add_argument("--same-device", action="store_true", help="all ranks use cuda:0 (functional testing only)")
    ap.add_argument("--force-grad-sync", action="store_true", help="diagnostic: run the N>1 gradient-bucket path at N=1 (copies "
                    "into the flat buckets, no collective) to price its overhead on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    return ap.parse_args()


def build_problem(args, device, rank, capturable=False):
    from epipolarpose_amd.core import integral_loss
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.hip import DeviceMeta
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd.synthetic import SyntheticScenes
    from epipolarpose_amd.utils.utils import get_optimizer

    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False               # no checkpoint on the box: random-init weights of the architecture
    cfg.MODEL.NUM_JOINTS = args.joints
    cfg.MODEL.DEPTH_RES = args.depth
    cfg.MODEL.IMAGE_SIZE = [args.image, args.image]
    cfg.MODEL.EXTRA.NUM_LAYERS = args.layers
    cfg.LOSS.FN = "SmoothL1JointLocationLoss"    # experiments/h36m/train.yaml:44
    torch.manual_seed(1234)                      # identical initial weights on every rank
    model = get_pose_net(cfg, is_train=True).to(device)
    model.train()
    criterion = getattr(integral_loss, cfg.LOSS.FN)(num_joints=cfg.MODEL.NUM_JOINTS, norm=cfg.LOSS.NORM).to(device)
    optimizer = get_optimizer(cfg, model, capturable=capturable)        # Adam, lr 1e-3 (train.yaml)

    n_group = args.batch // args.views
    scenes = SyntheticScenes(n_group=n_group, n_view=args.views, num_joints=args.joints, patch=256, seed=100 + rank)
    gen = torch.Generator(device="cpu").manual_seed(100 + rank)
    images = torch.randn((args.batch, 3, args.image, args.image), generator=gen).to(device)
    images = images.contiguous(memory_format=torch.channels_last)
    label = torch.from_numpy(scenes.label).to(device)
    weight = torch.from_numpy(scenes.weight).to(device)
    meta = DeviceMeta(scenes.meta, device) if args.workload == "ss" else None
    return cfg, model, criterion, optimizer, images, label, weight, meta, scenes


def cpu_baseline(args, scenes):
    """cpu_baseline leg: the oracle (fp32 torch-CPU restatement of the reference's model + criterion, float64 NumPy
    restatement of its self-supervision) timed on this host's cores on a bounded sample of the same workload."""
    import numpy as np
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from oracle import geometry as o_geo
    from oracle import network as o_net

    cores = min(os.cpu_count() or 1, 32)     # more threads than this only thrash on a 4-image batch
    torch.set_num_threads(cores)
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.EXTRA.NUM_LAYERS = args.joints, args.depth, args.layers
    torch.manual_seed(1234)
    sd = {k: v.detach().clone().contiguous() for k, v in get_pose_net(cfg, True).state_dict().items()}
    params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
    sd.update(params)
    opt = torch.optim.Adam(list(params.values()), lr=1e-3)
    b = args.cpu_batch
    x = torch.randn(b, 3, args.image, args.image)
    gt = torch.from_numpy(scenes.label[:b].copy())
    wt = torch.ones_like(gt)

    def step():
        opt.zero_grad()
        logits = o_net.forward(sd, x, args.layers, training=True, new_stats={})
        loss = o_net.joint_location_loss(logits, gt, wt, args.joints, "smoothl1")
        loss.backward()
        opt.step()
    t0 = time.perf_counter()
    step()                                   # warm-up (also bounds the leg: a slow host gets a 1-step sample)
    warm = time.perf_counter() - t0
    n, dt = 1, warm
    if warm < 15.0:
        t0 = time.perf_counter()
        n = 0
        while n < 1 or (time.perf_counter() - t0 < 10.0 and n < 8):
            step()
            n += 1
        dt = time.perf_counter() - t0
    out = {"value": round(b * n / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": "%d steps of batch %d (ResNet-%d, %dx%d, J=%d, D=%d; fwd + SmoothL1 soft-argmax loss + bwd + Adam), "
                     "fp32 torch-CPU oracle, %d threads" % (n, b, args.layers, args.image, args.image, args.joints,
                                                            args.depth, cores)}
    # self-supervision leg of the CPU path (float64 NumPy restatement, 1 core) + "MPJPE vs ref" on identical inputs
    cp = scenes.patch_coords(noise_px=1.0, seed=3)
    t0 = time.perf_counter()
    _, _, xw_ref, _ = o_geo.self_supervision(None, scenes.meta, n_view=scenes.n_view, coords_patch=cp)
    out["ss_groups_per_s_1core"] = round(scenes.n_group / (time.perf_counter() - t0), 2)
    if torch.cuda.is_available():
        from epipolarpose_amd import hip
        xyz = np.stack([cp[:, :, 0] / 256 - 0.5, cp[:, :, 1] / 256 - 0.5, cp[:, :, 2] / 256], 2).reshape(cp.shape[0], -1)
        dev = torch.device("cuda", torch.cuda.current_device())
        _, _, xw = hip.self_supervision(torch.from_numpy(xyz.astype(np.float32)).to(dev), hip.DeviceMeta(scenes.meta, dev),
                                        scenes.n_view, want_world=True)
        err = np.linalg.norm(xw.cpu().numpy() - xw_ref[:scenes.n_group], axis=2)
        out["mpjpe_vs_ref_mm"] = float(err.mean())
    return out


def main():
    args = parse_args()
    from epipolarpose_amd import distributed as epd
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.function import GraphedTrainStep, train_step

    if args.same_device:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    rank, world, local = epd.init_from_env(backend=args.backend, set_device=not args.same_device)
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)"
                         % (args.gpus, world, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    hip.load()
    torch.backends.cudnn.benchmark = True        # reference CUDNN.BENCHMARK: true -> MIOpen find mode

    use_graph = bool(args.graph)
    cfg, model, criterion, optimizer, images, label, weight, meta, scenes = build_problem(args, device, rank, capturable=use_graph)
    grad_sync = None
    if world > 1 or args.force_grad_sync:
        epd.broadcast_module(model, optimizer=optimizer)
        grad_sync = epd.BucketedGradSync(model, optimizer=optimizer)
    n_view = args.views if args.workload == "ss" else None
    # 4-view SS uses the V-view generalisation of the reference's iterative LS solver (V=2 is the reference itself)

    def eager_step():
        return train_step(model, criterion, optimizer, images, label, weight, meta=meta, n_view=n_view,
                          autocast=not args.fp32, grad_sync=grad_sync)
    step = eager_step
    if use_graph:
        if world > 1:
            raise SystemExit("--graph 1 is single-GPU only (the gradient all-reduce is issued eagerly)")
        step = GraphedTrainStep(model, criterion, optimizer, images, label, weight, meta=meta, n_view=n_view,
                                autocast=not args.fp32)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    th = time.perf_counter()
    step()                                           # diagnostic (untimed, part of warm-up): host cost of enqueueing ONE step
    host_one = time.perf_counter() - th              # into an empty queue, i.e. without back-pressure from the GPU
    torch.cuda.synchronize()
    barrier()
    hip.timer.reset()
    hip.timer.enabled = not use_graph            # events cannot be recorded inside a replayed graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_elapsed = time.perf_counter() - t0          # host-side enqueue time (diagnostic: how far the CPU runs ahead)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    hip.timer.enabled = False
    if use_graph:
        # per-kernel durations for the roofline: the same criterion kernels on the same resident logits-sized tensor,
        # launched eagerly with HIP events on the launch stream right after the timed region
        hip.timer.reset()
        hip.timer.enabled = True
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
            probe_logits = model(images)
        probe_logits = probe_logits.detach().requires_grad_(True)
        for _ in range(args.steps):
            criterion(probe_logits, label, weight).backward()
        torch.cuda.synchronize()
        hip.timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        global_batch = args.batch * world
        elem = 4 if args.fp32 else 2
        vox = args.joints * args.depth * (args.image // 4) ** 2
        ksum = hip.timer.summary()
        MFMA_PEAK = 2500.0      # dense bf16 TFLOP/s (MI355X_MICROARCH.md)
        b, hm, cd, jd = args.batch, args.image // 4, 256, args.joints * args.depth
        # algorithmic FLOPs (2 x MACs, BASELINE.md section 3 / SURVEY 2.2), B = per-GPU batch
        deconv_macs = [2048 * 256 * 16 * (hm // 8) ** 2, 256 * 256 * 16 * (hm // 4) ** 2, 256 * 256 * 16 * (hm // 2) ** 2]
        final_macs = cd * jd * hm * hm
        per_step_flops = {"epi_deconv4x4s2_fwd": 2.0 * b * sum(deconv_macs), "epi_deconv4x4s2_bwd_data": 2.0 * b * sum(deconv_macs),
                          "epi_deconv4x4s2_bwd_weight": 2.0 * b * sum(deconv_macs), "epi_gemm_bf16": 2.0 * b * final_macs * 2,
                          "epi_gemm_tn_bf16": 2.0 * b * final_macs}
        per_kernel = {}
        steps_timed = args.steps
        for name, fl in per_step_flops.items():
            if name in ksum:
                n, ms = ksum[name]
                step_ms = ms * n / steps_timed
                per_kernel[name] = {"launches_per_step": n / steps_timed, "ms_per_step": round(step_ms, 4),
                                    "achieved_tflops": round(fl / (step_ms * 1e-3) / 1e12, 1)}
        # dominant hand-written kernel: head_gemm_kernel = the 8 NT launches per step (3 deconv fwd, 3 deconv bwd-data,
        # final 1x1 fwd + bwd-data); "achieved" = their algorithmic FLOPs / their summed HIP-event durations
        nt = ("epi_deconv4x4s2_fwd", "epi_deconv4x4s2_bwd_data", "epi_gemm_bf16")
        head_ms = sum(per_kernel[k]["ms_per_step"] for k in nt if k in per_kernel)
        head_flops = sum(per_step_flops[k] for k in nt if k in per_kernel)
        head_launches = sum(per_kernel[k]["launches_per_step"] for k in nt if k in per_kernel)
        n_b, ms_b = ksum["epi_softargmax3d_bwd"]
        n_f, ms_f = ksum["epi_softargmax3d_fwd"]
        bytes_bwd = 2.0 * args.batch * vox * elem          # 1 read of the logits + 1 write of dlogits
        bytes_fwd = 1.0 * args.batch * vox * elem          # 1 read of the logits
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("head_gemm_kernel/b%d" % args.batch)
            except Exception:
                traffic = None
        hbm = {"softargmax_bwd_kernel (epi_softargmax3d_bwd)": {
                   "bound": "hbm", "achieved": round(bytes_bwd / (ms_b * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(bytes_bwd / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": bytes_bwd,
                   "avg_ms": round(ms_b, 5), "launches": n_b},
               "softargmax_partial+combine (epi_softargmax3d_fwd)": {
                   "bound": "hbm", "achieved": round(bytes_fwd / (ms_f * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": round(bytes_fwd / (ms_f * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": bytes_fwd,
                   "avg_ms": round(ms_f, 5), "launches": n_f}}
        if head_ms > 0:
            ach = head_flops / (head_ms * 1e-3) / 1e12
            roofline = {"kernel": "head_gemm_kernel + head_gemm_astat_kernel (deconvolution head fwd + bwd-data, final 1x1 conv bwd-data; "
                                  "final 1x1 conv fwd on the A-stationary variant: %d launches per step; the hand-written kernel "
                                  "family with the largest share of the step)" % round(head_launches),
                        "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK, 4),
                        "traffic": traffic, "algorithmic_flops_per_launch": head_flops / max(head_launches, 1),
                        "avg_ms": round(head_ms / max(head_launches, 1), 5), "launches_per_step": head_launches,
                        "entry_points": per_kernel, "other": hbm}
        else:
            k = "softargmax_bwd_kernel (epi_softargmax3d_bwd)"
            roofline = dict(hbm[k], kernel=k, traffic=None, other={kk: v for kk, v in hbm.items() if kk != k})
        line = {
            "metric": "images/sec (4-view 256x256, ResNet-50) at 1/2/4/8 MI355X; MPJPE vs ref",
            "value": round(global_batch * args.steps / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": ("configs[1]: ResNet-%d Integral-pose, 4-view %dx%d synthetic, batch=%d/GPU, fully-supervised "
                                    "SmoothL1 loss" if args.workload == "fs" else
                                    "configs[2]: ResNet-%d self-supervised, 4-view %dx%d epipolar-triangulation pseudo-labels, "
                                    "batch=%d/GPU") % (args.layers, args.image, args.image, args.batch),
                       "global_batch": global_batch, "joints": args.joints, "depth_res": args.depth,
                       "optimizer": "adam", "parallelism": "dp%d" % world, "final_loss": round(final_loss, 6),
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "host_enqueue_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
                       "host_enqueue_ms_one_step_empty_queue": round(host_one * 1e3, 3)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, scenes)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
