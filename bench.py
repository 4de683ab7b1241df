#!/usr/bin/env python
"""Benchmark of the EpipolarPose training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU -- under torch.distributed.run when RANK / WORLD_SIZE are
                                                            set, else bench.py starts its own N ranks under it)

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
    ap.add_argument("--allreduce", choices=["default", "ring", "direct"], default="default",
                    help="gradient collective at N > 1 (SURVEY section 5: xGMI is point-to-point): default = one all-reduce per bucket, algorithm and "
                         "channels chosen by RCCL; ring = the same with NCCL_ALGO=Ring pinned before the communicator exists; direct = reduce-scatter + "
                         "all-gather as grouped point-to-point transfers, one per peer / xGMI link (epipolarpose_amd/distributed.py) -- EXPERIMENTAL: tested on "
                         "gloo with CPU tensors only, it has never run on RCCL (no multi-GPU node has been available)")
    ap.add_argument("--force-grad-sync", action="store_true", help="diagnostic: run the N>1 gradient-bucket path at N=1 (copies "
                    "into the flat buckets, no collective) to price its overhead on one GPU")
    ap.add_argument("--refiner-leg", action="store_true", help="also time the refiner MLP beside the step (configs[4] names it: post-lift refinement of "
                    "the batch's poses in inference mode, and one refiner training step at the reference's batch size 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ss-leg", action="store_true", help="skip the extra configs[2] (self-supervised) measurement that rides along")
    ap.add_argument("--no-loader-leg", action="store_true", help="skip the extra measurement with the GPU input pipeline in the step "
                    "(synthetic uint8 frames in HBM -> augmentation draws, crop, occlusion, normalisation -> the same training step)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-baseline-only", action="store_true", help="run ONLY the cpu_baseline leg (no GPU needed) and print it as JSON: in the build "
                    "container this times the reference itself (kind 'reference'), which the GPU box cannot (no /root/reference there)")
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
    b = args.cpu_batch
    x = torch.randn(b, 3, args.image, args.image)
    gt = torch.from_numpy(scenes.label[:b].copy())
    wt = torch.ones_like(gt)
    # kind "reference": the reference's OWN lib/models/pose3d_resnet.py + lib/core/integral_loss.py, imported from /root/reference with the
    # third-party shims of tests/golden/ref_shims.py (easydict; torch.cuda.comm.broadcast -> identity) -- only where that tree exists
    # (the build container; never on the GPU box).  Otherwise kind "port": the oracle's restatement of the same model + criterion.
    ref = None
    if os.path.isdir("/root/reference/lib"):
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
            import ref_shims
            ref = ref_shims.load_reference()
        except Exception as e:               # (a broken shim must not cost the bench line its baseline)
            print("bench.py: reference import failed (%s): cpu_baseline falls back to the port" % e, file=sys.stderr)
            ref = None
    if ref is not None:
        import copy
        rcfg = copy.deepcopy(ref.config.config)
        rcfg.MODEL.NUM_JOINTS, rcfg.MODEL.DEPTH_RES, rcfg.MODEL.IMAGE_SIZE = args.joints, args.depth, [args.image, args.image]
        rcfg.MODEL.INIT_WEIGHTS = False
        rcfg.MODEL.EXTRA.NUM_LAYERS = args.layers
        torch.manual_seed(1234)
        rmodel = ref.pose3d_resnet.get_pose_net(rcfg, is_train=True)
        rmodel.train()
        rcrit = ref.integral_loss.SmoothL1JointLocationLoss(num_joints=args.joints)
        opt = torch.optim.Adam(rmodel.parameters(), lr=1e-3)

        def step():
            opt.zero_grad()
            loss = rcrit(rmodel(x), gt, wt)
            loss.backward()
            opt.step()
    else:
        cfg = default_config()
        cfg.MODEL.INIT_WEIGHTS = False
        cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.EXTRA.NUM_LAYERS = args.joints, args.depth, args.layers
        torch.manual_seed(1234)
        sd = {k: v.detach().clone().contiguous() for k, v in get_pose_net(cfg, True).state_dict().items()}
        params = {k: v.requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point and "running" not in k}
        sd.update(params)
        opt = torch.optim.Adam(list(params.values()), lr=1e-3)

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
    out = {"value": round(b * n / dt, 3), "unit": "images/s", "cores": cores, "kind": "reference" if ref is not None else "port",
           "sample": "%d steps of batch %d (ResNet-%d, %dx%d, J=%d, D=%d; fwd + SmoothL1 soft-argmax loss + bwd + Adam), %s, %d threads"
                     % (n, b, args.layers, args.image, args.image, args.joints, args.depth,
                        "the reference's own pose3d_resnet.py + integral_loss.py (fp32 torch-CPU, /root/reference)" if ref is not None else
                        "fp32 torch-CPU oracle port (/root/reference is absent on this host)", cores)}
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
    if ref is None:
        # The reference itself cannot travel to the GPU box; its timing on the BUILD container's cores (bench.py --cpu-baseline-only there, committed
        # under profiles/) rides in the same record, so that the port's figure on this host and the reference's own figure are never apart.
        path = os.path.join(ROOT, "profiles", "r05_cpu_baseline_reference_build_container.json")
        if not os.path.isfile(path):
            path = os.path.join(ROOT, "profiles", "r04_cpu_baseline_reference_build_container.json")
        try:
            with open(path) as f:
                rec = json.load(f)
            out["reference_build_container"] = dict(rec["cpu_baseline"], host_cpus=rec.get("host", {}).get("cpus"), file=os.path.relpath(path, ROOT))
        except (OSError, ValueError, KeyError):
            pass
    return out


MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 (MI355X_MICROARCH.md)


def stem_conv_ms(model, images, reps=5):
    """The 7x7 stem is the one convolution left to the library (3 input channels): timed on its own after the timed region
    (forward + weight gradient; it has no data gradient), so that the conv-stack figure covers EVERY convolution."""
    conv = model.conv1
    x = images.to(torch.bfloat16)
    w = (getattr(conv, "weight_lp", None) if getattr(conv, "weight_lp", None) is not None else conv.weight).detach().to(torch.bfloat16)
    y = torch.nn.functional.conv2d(x, w, stride=2, padding=3)
    dy = torch.randn_like(y)

    def run():
        torch.nn.functional.conv2d(x, w, stride=2, padding=3)
        torch.ops.aten.convolution_backward(dy, x, w, None, (2, 2), (3, 3), (1, 1), False, (0, 0), 1, (False, True, False))
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        run()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, 2.0 * 2 * x.shape[0] * y.shape[2] * y.shape[3] * 64 * 147


def event_pair_overhead_ms(device, reps=256):
    """What a HIP event pair adds to the duration it reports for ONE launch of a busy queue, measured in this run: the same `reps` kernels (a 64 MB fill,
    ~11 us: the size of a typical launch of the step) are enqueued twice behind a few milliseconds of queued fills (the host runs ahead, the GPU sets
    the pace) -- once with a pair around every launch (sum of the pairs' elapsed times), once back to back between ONE outer pair.  The back-to-back arm
    is what rocprofv3's kernel table reports (its durations tile a busy queue: 6.01 of 6.08 ms of a step have a kernel running); the difference per
    launch is the pair's own cost.  tools/probe_event_pair_overhead.py: 2.2 us for kernels of 7 us and more, rising to 4.8 us under a 1.7 us kernel
    (round 5's BatchNorm block: 1.47 ms from event pairs beside 1.23 ms in the rocprofv3 table of the same command = 106 pairs x 2.3 us)."""
    buf = torch.empty(64 << 20, dtype=torch.uint8, device=device)
    big = torch.empty(1 << 29, dtype=torch.uint8, device=device)

    def head():                                   # ~8 ms of queued work: the 3 x reps enqueue calls below take the host 2-4 ms
        for _ in range(80):
            big.fill_(0)
    for _ in range(20):
        buf.fill_(1)
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    head()
    for a, b in pairs:
        a.record()
        buf.fill_(1)
        b.record()
    torch.cuda.synchronize()
    with_pairs = sum(a.elapsed_time(b) for a, b in pairs)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    head()
    s.record()
    for _ in range(reps):
        buf.fill_(1)
    e.record()
    torch.cuda.synchronize()
    return max(0.0, (with_pairs - s.elapsed_time(e)) / reps)


def build_roofline(args, ksum, glue_times, model, images, pair_ms=0.0):
    """``roofline`` object of the JSON line.  Every duration is a HIP-event measurement on the launch stream taken INSIDE the timed
    region: the C++ glue brackets its epi_* launches (backbone convolutions, BatchNorm), ``hip.timer`` the ctypes entry points
    (deconvolution head, final 1x1 convolution, soft-argmax criterion, Adam).  FLOPs / bytes are the algorithmic ones of
    BASELINE.md section 3 (2 x MACs; one read + one write of what the operation must touch)."""
    steps = float(args.steps)
    b, hm, cd, jd = args.batch, args.image // 4, 256, args.joints * args.depth
    elem = 4 if args.fp32 else 2
    vox = args.joints * args.depth * hm * hm
    fam = {}

    def add(name, bound, n, ms_total, flops=0.0, nbytes=0.0):
        if n <= 0 or ms_total <= 0:
            return
        raw = ms_total
        # every timed entry is one event pair: its own overhead (event_pair_overhead_ms) comes off, so that short launches are not inflated
        # (never below a third of the raw figure: an entry that is a handful of back-to-back kernels under one pair carries the overhead once)
        ms_total = max(ms_total - n * pair_ms, raw / 3.0)
        e = {"bound": bound, "launches_per_step": round(n / steps, 2), "ms_per_step": round(ms_total / steps, 4)}
        if pair_ms > 0:
            e["ms_per_step_with_event_overhead"] = round(raw / steps, 4)
        if bound == "mfma":
            e["achieved_tflops"] = round(flops / (ms_total * 1e-3) / 1e12, 1)
            e["frac"] = round(e["achieved_tflops"] / MFMA_PEAK_TFLOPS, 4)
            e["flops_per_step"] = flops / steps
        else:
            e["achieved_gbs"] = round(nbytes / (ms_total * 1e-3) / 1e9, 1)
            e["frac"] = round(e["achieved_gbs"] / HBM_PEAK_GBS, 4)
            e["algorithmic_bytes_per_step"] = nbytes / steps
        fam[name] = e
    for name, (n, ms, flops, nbytes) in glue_times.items():         # everything routed through the C++ glue: convolutions, head, BatchNorm, max-pool
        if name.startswith("conv_"):
            add("backbone_" + name, "mfma", n, ms, flops=flops)
        elif name.startswith("stem_conv"):
            add(name, "mfma", n, ms, flops=flops)
        elif name.startswith("head_"):
            add(name, "mfma", n, ms, flops=flops)
        else:
            add(name, "hbm", n, ms, nbytes=nbytes)
    deconv_macs = [2048 * 256 * 16 * (hm // 8) ** 2, 256 * 256 * 16 * (hm // 4) ** 2, 256 * 256 * 16 * (hm // 2) ** 2]
    final_macs = cd * jd * hm * hm
    head = {"epi_deconv4x4s2_fwd": 2.0 * b * sum(deconv_macs), "epi_deconv4x4s2_bwd_data": 2.0 * b * sum(deconv_macs),
            "epi_deconv4x4s2_bwd_weight": 2.0 * b * sum(deconv_macs), "epi_gemm_bf16": 2.0 * b * final_macs * 2,
            "epi_gemm_tn_bf16": 2.0 * b * final_macs}
    for name, per_step in head.items():
        if name in ksum:
            n, ms = ksum[name]
            add("head_" + name[4:], "mfma", n, ms * n, flops=per_step * steps)
    for name, factor in (("epi_softargmax3d_fwd", 1.0), ("epi_softargmax3d_bwd", 2.0)):
        if name in ksum:
            n, ms = ksum[name]
            add(name[4:], "hbm", n, ms * n, nbytes=factor * b * vox * elem * n)
    if "epi_adam_step" in ksum:
        n, ms = ksum["epi_adam_step"]
        nparam = sum(p.numel() for p in model.parameters())
        add("adam_step", "hbm", n, ms * n, nbytes=30.0 * nparam * n)
    gemm = {k: v for k, v in fam.items() if v["bound"] == "mfma"}
    bn = {k: v for k, v in fam.items() if k.startswith("bn_")}
    out = {}
    if gemm:
        ms = sum(v["ms_per_step"] for v in gemm.values())
        flops = sum(v["flops_per_step"] for v in gemm.values())
        launches = sum(v["launches_per_step"] for v in gemm.values())
        ach = flops / (ms * 1e-3) / 1e12
        out = {"kernel": "head_gemm_kernel / head_gemm_tn_kernel / head_gemm_astat_kernel -- the implicit-GEMM family that runs EVERY convolution "
                         "of the network behind the 7x7 stem (backbone conv1..3 + downsample forward / backward-data / backward-weight, the "
                         "deconvolution head, the final 1x1 convolution; split-K finish kernels included in the time): the largest "
                         "share of the step",
               "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
               "traffic": None, "algorithmic_flops_per_launch": flops / max(launches, 1e-9), "avg_ms": round(ms / max(launches, 1e-9), 5),
               "launches_per_step": round(launches, 1), "ms_per_step": round(ms, 4),
               # context, not the denominator of `frac`: back-to-back MFMAs ALONE on random bf16 operand values hold 1.81 GHz on this chip
               # (tools/w4_lab.hip, profiles/r05_w4_lab.txt) -- the rate no bf16 GEMM on such operands exceeds here; hipBLASLt's 8192^3 reaches 1613-1662
               "peak_sustained_mfma_only_random_bf16": 1668.0}
        own_stem = any(k.startswith("stem_conv") for k in fam)       # round 3: the stem runs on the implicit-GEMM kernels and is in the family already
        stem_ms, stem_flops = (0.0, 0.0) if own_stem else stem_conv_ms(model, images)
        out["conv_stack"] = {"what": "every convolution of the network" + ("" if own_stem else " incl. the library 7x7 stem (timed separately after the timed region)"),
                             "flops_per_step": flops + stem_flops, "ms_per_step": round(ms + stem_ms, 4),
                             "achieved_tflops": round((flops + stem_flops) / ((ms + stem_ms) * 1e-3) / 1e12, 1),
                             "frac": round((flops + stem_flops) / ((ms + stem_ms) * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    if bn:
        ms = sum(v["ms_per_step"] for v in bn.values())
        nbytes = sum(v["algorithmic_bytes_per_step"] for v in bn.values())
        out["batchnorm"] = {"what": "bn_stats / bn_apply / bn_bwd_reduce / bn_bwd_apply: the dominant HBM-bound family (fused BatchNorm + "
                                    "residual + ReLU, forward and backward)", "bound": "hbm", "ms_per_step": round(ms, 4),
                            # (a "+" entry is two kernels -- its own statistics / reduction pass and the apply pass --, the others one: the
                            #  statistics came from the producing GEMM's epilogue, the backward reduction from the backward-data GEMM's)
                            "launches_per_step": round(sum(v["launches_per_step"] * (2 if "+" in k else 1) for k, v in bn.items()), 1),
                            "achieved": round(nbytes / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    if not out:          # hipGraph replay: only the criterion probe was timed
        k = "softargmax3d_bwd"
        e = fam.get(k, {})
        out = {"kernel": "softargmax_bwd_kernel", "bound": "hbm", "achieved": e.get("achieved_gbs"), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": e.get("frac"), "traffic": None}
    out["families"] = fam
    out["measured"] = ("HIP events on the launch stream around every launch, over `steps` additional steps of the same workload right after "
                       "the timed region (recording them inside it makes the step host-bound and would falsify `value`); in these steps the "
                       "weight gradients run on the main stream too, so that every figure is the kernel's own duration; every pair's own overhead "
                       "(event_pair_overhead_us, measured in this run: 64 MB fills in a busy queue with a pair around each against the same "
                       "kernels back to back) is subtracted per launch, which is what makes the figures agree with the rocprofv3 kernel table of the "
                       "same command (profiles/)")
    out["event_pair_overhead_us"] = round(pair_ms * 1e3, 3)
    out["note"] = "traffic: not measured in this run (PMC passes are separate rocprofv3 runs: profiles/*pmc*)"
    default_workload = (args.workload == "fs" and args.layers == 50 and args.image == 256 and args.batch == 32 and args.views == 4 and
                        args.joints == 17 and args.depth == 64 and not args.fp32 and not args.graph)
    pmc = pmc_traffic(default_workload)
    if pmc and "traffic" in out and out.get("launches_per_step"):
        g = pmc["families"].get("implicit_gemm")
        if g:
            # per launch of the family as the PMC passes counted them (their launch set includes the split finishes and slab sums: 141.6 per step;
            # the timed `launches_per_step` counts entry-point calls) -- both figures of the quotient from ONE source
            out["traffic"] = round(g.get("hbm_bytes_per_launch", g["hbm_bytes_per_step"] / out["launches_per_step"]))
            out["traffic_per_step"] = round(g["hbm_bytes_per_step"])
            out["traffic_launches_per_step"] = g.get("launches_per_step")
            if g.get("mfma_busy") is not None:
                # share of the family's own kernel cycles a SIMD's matrix pipe was busy: (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs),
                # summed over the family's launches of the profiled steps (profiles/r05_pmc_sq_step.csv holds it per kernel class)
                out["mfma_busy"] = g["mfma_busy"]
        bnp = pmc["families"].get("batchnorm")
        if bnp and "batchnorm" in out:
            out["batchnorm"]["traffic_per_step"] = round(bnp["hbm_bytes_per_step"])
            out["batchnorm"]["algorithmic_bytes_per_step"] = round(sum(v["algorithmic_bytes_per_step"] for v in bn.values()))
        out["note"] = ("traffic: HBM bytes per launch of the family (per step / launches_per_step) from the committed PMC passes of this same "
                       "command and workload -- %s; %s; %s" % (PMC_FILE, pmc["source"], pmc["correction"]))
    return out


def refiner_leg(args, device):
    """configs[4]'s "refiner MLP post-lift" (refiner/model.py:74-143, refiner/main.py:31-60), timed beside the pose step: (a) the refinement of one
    batch of lifted poses (15 joints x 3, inference mode, both heads), (b) one training step of the refiner at the reference's DataLoader batch size."""
    from epipolarpose_amd.refiner.main import TwoHeadMSE, make_optimizer
    from epipolarpose_amd.refiner.model import get_model, weight_init
    torch.manual_seed(7)
    model = get_model(weights=None).to(device)
    model.apply(weight_init)
    poses = torch.randn(args.batch, 45, device=device)
    model.eval()
    with torch.no_grad():
        for _ in range(5):
            model(poses)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            model(poses)
        torch.cuda.synchronize()
        infer_ms = (time.perf_counter() - t0) / 50 * 1e3
    model.train()
    criterion, optimizer = TwoHeadMSE(), make_optimizer(model, lr=1e-3)
    inp, tar = torch.randn(64, 45, device=device), torch.randn(64, 45, device=device)

    def rstep():
        optimizer.zero_grad()
        loss = criterion(model(inp), tar)
        loss.backward()
        optimizer.step()
        return loss
    for _ in range(5):
        rstep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        loss = rstep()
    torch.cuda.synchronize()
    train_ms = (time.perf_counter() - t0) / 50 * 1e3
    return {"workload": "refiner MLP (LinearModelPG, 1024 wide, 2 stages): post-lift refinement of %d poses in inference mode; one training step at batch 64" % args.batch,
            "post_lift_ms_per_batch": round(infer_ms, 4), "post_lift_poses_per_s": round(args.batch / (infer_ms * 1e-3), 1),
            "train_ms_per_step": round(train_ms, 4), "train_samples_per_s": round(64 / (train_ms * 1e-3), 1), "final_loss": round(float(loss.item()), 6)}


PMC_FILE = "profiles/r06_pmc_step_families.json"


def pmc_traffic(default_workload):
    """The committed summary of the three rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ + GRBM) over bench.py at the default workload
    (tools/gpu_pmc_step_r05.sh): counters cannot be read from inside the process, so `roofline.traffic` quotes that file -- and only for the
    workload it was taken on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_FILE)
    if not default_workload or not os.path.isfile(path):
        return None
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def self_launch(args):
    """``python bench.py --gpus N`` WITHOUT a launcher (no RANK / WORLD_SIZE in the environment): start N ranks of this same command under
    ``torch.distributed.run`` on this node -- one process per GPU, rendezvous on 127.0.0.1 at a free port -- and hand their output and exit
    code through.  (The reference starts its N replicas inside one process, ``torch.nn.DataParallel``: scripts/train.py:93-94,143.)"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # the host driver only supports dmabuf IPC (RCCL over xGMI needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without a launcher: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def rank_inventory(world, device):
    """Who took part, for the JSON line's `config`: backend, RCCL version, every rank's device."""
    inv = {"backend": None, "world_size": world, "rccl_version": None, "devices": [torch.cuda.get_device_name(device)],
           # what decides the all-reduce's shape on xGMI: the env RCCL reads (None = its own choice) -- `--allreduce` is recorded by the caller
           "NCCL_ALGO": os.environ.get("NCCL_ALGO"), "NCCL_PROTO": os.environ.get("NCCL_PROTO"), "NCCL_MIN_NCHANNELS": os.environ.get("NCCL_MIN_NCHANNELS")}
    if world > 1:
        inv["backend"] = torch.distributed.get_backend()
        names = [None] * world
        torch.distributed.all_gather_object(names, "rank %d: cuda:%d %s" % (torch.distributed.get_rank(), device.index,
                                                                             torch.cuda.get_device_name(device)))
        inv["devices"] = names
        if inv["backend"] == "nccl":
            try:
                inv["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:                    # (version query only: never fail the bench for it)
                inv["rccl_version"] = "unknown"
    return inv


def main():
    args = parse_args()
    if args.cpu_baseline_only:
        from epipolarpose_amd.synthetic import SyntheticScenes
        scenes = SyntheticScenes(n_group=args.batch // args.views, n_view=args.views, num_joints=args.joints, patch=256, seed=100)
        print(json.dumps({"cpu_baseline": cpu_baseline(args, scenes), "host": {"cpus": os.cpu_count()}}), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(args))
    from epipolarpose_amd import distributed as epd
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.function import GraphedTrainStep, train_step

    if args.same_device:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    if args.allreduce == "ring":
        os.environ["NCCL_ALGO"] = "Ring"               # (read by RCCL when the communicator is created, i.e. at the first collective)
    rank, world, local = epd.init_from_env(backend=args.backend, set_device=not args.same_device)
    if args.same_device:
        local = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or without any launcher: "
                         "bench.py then starts its own ranks)" % (args.gpus, world, args.gpus))
    if world > 1:
        print("bench.py: rank %d/%d up, backend %s" % (rank, world, torch.distributed.get_backend()), file=sys.stderr, flush=True)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    hip.load()
    torch.backends.cudnn.benchmark = True        # reference CUDNN.BENCHMARK: true -> MIOpen find mode

    use_graph = bool(args.graph)
    cfg, model, criterion, optimizer, images, label, weight, meta, scenes = build_problem(args, device, rank, capturable=use_graph)
    grad_sync = None
    if world == 1 and not args.force_grad_sync and not args.graph:
        from epipolarpose_amd.optim import enable_step_in_backward
        enable_step_in_backward(optimizer, model)       # EPI_STEP_IN_BACKWARD=1 only (measured: < 1 %)
    if world > 1 or args.force_grad_sync:
        epd.broadcast_module(model, optimizer=optimizer)
        grad_sync = epd.BucketedGradSync(model, optimizer=optimizer, collective="direct" if args.allreduce == "direct" else "allreduce")
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
    # ... and of a burst of three steps from an empty queue (untimed warm-up as well): the host's own enqueue cost per step with warm
    # caches and allocator, still without back-pressure.  The steady-state figure below (host_enqueue_ms_per_step, over `steps` steps)
    # additionally contains the time the host spends BLOCKED inside HIP calls once it has run as far ahead of the GPU as the runtime's
    # queue lets it -- on a GPU-bound step it converges to the GPU's own step time whatever the host costs.
    th = time.perf_counter()
    for _ in range(3):
        step()
    host_burst = (time.perf_counter() - th) / 3.0
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_elapsed = time.perf_counter() - t0          # host-side enqueue time (diagnostic: how far the CPU runs ahead)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    # Per-kernel durations for the roofline: the SAME step, `steps` more times, now with a HIP event pair around every launch
    # (on the launch stream).  Kept out of the timed region above: ~660 hipEventRecord calls per step cost the host ~8 us each on
    # ROCm 7.2 and turn the step host-bound (10.9 ms instead of 8.9 ms measured), which would falsify `value`.
    glue_times = {}
    if not use_graph:
        hip.timer.reset()
        hip.timer.enabled = True
        hip.glue().timing_collect()              # (clears) -- the C++ glue's launches carry their own HIP events
        hip.glue().timing_enable(True)
        # every launch on ONE stream here: with the weight gradients on the second stream (the timed region above) an event pair would
        # also measure what the two streams take from each other, not the kernel
        stream_mode = hip.glue().wgrad_stream_mode(0)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        hip.glue().wgrad_stream_mode(stream_mode)
        hip.timer.enabled = False
        hip.glue().timing_enable(False)
        glue_times = hip.glue().timing_collect()
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
    inventory = rank_inventory(world, device)
    inventory["allreduce"] = args.allreduce
    if grad_sync is not None:
        inventory["gradient_buckets"] = len(grad_sync.buckets)
        inventory["gradient_bytes_per_step"] = grad_sync.total_bytes()
    # the other single-GPU workload of BASELINE.json (configs[2]: pseudo labels from multi-view triangulation inside the step)
    # rides along as an extra field of the same JSON line: same model / optimizer state, `steps` more steps, same timing rules
    ss_line = None
    if args.workload == "fs" and not use_graph and not args.no_ss_leg:
        from epipolarpose_amd.hip import DeviceMeta
        ss_meta = DeviceMeta(scenes.meta, device)

        def ss_step():
            return train_step(model, criterion, optimizer, images, label, weight, meta=ss_meta, n_view=args.views,
                              autocast=not args.fp32, grad_sync=grad_sync)
        for _ in range(2):
            ss_step()
        torch.cuda.synchronize()
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            ss_loss = ss_step()
        torch.cuda.synchronize()
        barrier()
        ss_elapsed = time.perf_counter() - ts
        if world > 1:
            t = torch.tensor([ss_elapsed], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ss_elapsed = float(t.item())
        ss_line = {"workload": "configs[2]: self-supervised, %d-view epipolar-triangulation pseudo-labels inside the step, batch=%d/GPU"
                               % (args.views, args.batch), "value": round(args.batch * world * args.steps / ss_elapsed, 2), "unit": "images/s",
                   "ms_per_step": round(ss_elapsed / args.steps * 1e3, 3), "final_loss": round(float(ss_loss.item()), 6)}
        # what this workload adds to the fully-supervised step is ONE launch (decode -> triangulate -> re-project fused: csrc/selfsup.hip); SURVEY 8d: at
        # the configuration's size its input is 7.5 KB -- launch-latency bound, so the figure that matters is microseconds per step; the HBM fraction of the
        # triangulation kernels is shown on a bulk batch (tools/bench_kernels.py tri -> profiles/r05_microbench_tri.txt)
        try:
            xyz_ss = torch.zeros(args.batch, 3 * args.joints, device=device)
            for _ in range(3):
                hip.self_supervision(xyz_ss, ss_meta, args.views)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 50
            e0.record()
            for _ in range(reps):
                hip.self_supervision(xyz_ss, ss_meta, args.views)
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            n_grp = args.batch // args.views
            nbytes = n_grp * ((args.views * args.joints * 2 + args.views * 12 + args.joints * 3) * 4 + (args.views * 22 + args.views * args.joints * 3) * 4)
            ss_line["roofline"] = {"kernel": "self_supervision_kernel (the launch this workload adds to configs[1]'s step; the rest of the step is configs[1]'s `roofline`)",
                                   "bound": "hbm", "achieved": round(nbytes / us * 1e-3, 4), "peak": 8000.0, "unit": "GB/s", "frac": round(nbytes / us * 1e-3 / 8000.0, 7),
                                   "traffic": None, "algorithmic_bytes_per_launch": nbytes, "us_per_launch_back_to_back": round(us, 2), "launches_per_step": 1,
                                   "note": "launch-latency bound at the configuration's size (SURVEY 8d: %d groups = %.1f KB per step); the fraction is reported for "
                                           "form, the bulk figure is in profiles/r05_microbench_tri.txt" % (n_grp, nbytes / 1e3)}
        except Exception as e:                   # (a diagnostic: never fail the bench line for it)
            ss_line["roofline"] = {"error": str(e)}

    # the same step fed by the GPU input pipeline: uint8 BGR frames resident in HBM, per batch the reference's augmentation
    # draws + label arithmetic on the host and ONE crop / occlusion / normalisation launch (dataset/synthetic_frames.py); rides along like the SS leg
    loader_line = None
    if args.workload == "fs" and not use_graph and not args.no_loader_leg and args.batch % args.views == 0:
        from epipolarpose_amd.dataset.synthetic_frames import FramePatchLoader, SyntheticFrames
        n_grp = args.batch // args.views
        frames = SyntheticFrames(n_group=2 * n_grp, n_view=args.views, num_joints=args.joints, seed=7 + rank, device=device)
        loader = FramePatchLoader(frames, groups_per_batch=n_grp, patch=args.image, augment=True, occlusion=True, seed=3 + rank,
                                  dtype=torch.float32 if args.fp32 else torch.bfloat16)
        groups = [list(range(n_grp)), list(range(n_grp, 2 * n_grp))]

        def loader_step(i):
            data, lab, wt, _ = loader.batch(groups[i % 2])
            return train_step(model, criterion, optimizer, data, lab, wt, autocast=not args.fp32, grad_sync=grad_sync)
        for i in range(2):
            loader_step(i)
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(3):
            loader.batch(groups[i % 2])
        host_batch = (time.perf_counter() - th) / 3.0
        torch.cuda.synchronize()
        barrier()
        tl = time.perf_counter()
        for i in range(args.steps):
            l_loss = loader_step(i)
        torch.cuda.synchronize()
        barrier()
        l_elapsed = time.perf_counter() - tl
        if world > 1:
            t = torch.tensor([l_elapsed], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            l_elapsed = float(t.item())
        loader_line = {"workload": "configs[1] fed by the GPU input pipeline: %d uint8 1000x1000 frames in HBM, augmentation + occlusion + crop + "
                                   "normalisation per step, batch=%d/GPU" % (2 * args.batch, args.batch),
                       "value": round(args.batch * world * args.steps / l_elapsed, 2), "unit": "images/s",
                       "ms_per_step": round(l_elapsed / args.steps * 1e3, 3), "host_ms_per_batch_pipeline_only": round(host_batch * 1e3, 3),
                       "final_loss": round(float(l_loss.item()), 6)}

    refiner_line = None
    if args.refiner_leg and not use_graph:
        refiner_line = refiner_leg(args, device)

    if rank == 0:
        global_batch = args.batch * world
        elem = 4 if args.fp32 else 2
        vox = args.joints * args.depth * (args.image // 4) ** 2
        ksum = hip.timer.summary()
        roofline = build_roofline(args, ksum, glue_times, model, images, pair_ms=event_pair_overhead_ms(device))
        line = {
            "metric": "images/sec (4-view 256x256, ResNet-50) at 1/2/4/8 MI355X; MPJPE vs ref",
            "value": round(global_batch * args.steps / elapsed, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
            # (the per-GPU share of configs[4] -- ResNet-152 at 384 x 384 -- is `--layers 152 --image 384`: any other shape than the default names itself)
            "config": {"workload": (("configs[4] per-GPU share: " if (args.layers, args.image) == (152, 384) else
                                     ("configs[1]: " if args.workload == "fs" else "configs[2]: ") if (args.layers, args.image) == (50, 256) else "") +
                                    ("ResNet-%d Integral-pose, 4-view %dx%d synthetic, batch=%d/GPU, fully-supervised SmoothL1 loss" if args.workload == "fs" else
                                     "ResNet-%d self-supervised, 4-view %dx%d epipolar-triangulation pseudo-labels, batch=%d/GPU")
                                    % (args.layers, args.image, args.image, args.batch)),
                       "global_batch": global_batch, "joints": args.joints, "depth_res": args.depth,
                       "optimizer": "adam", "parallelism": "dp%d" % world, "final_loss": round(final_loss, 6),
                       "ranks": inventory,
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "streams": 1 if use_graph or hip.glue().wgrad_stream_mode(-1) == 0 else 2,
                       "host_enqueue_ms_per_step": round(host_elapsed / args.steps * 1e3, 3),
                       "host_enqueue_ms_one_step_empty_queue": round(host_one * 1e3, 3),
                       "host_enqueue_ms_per_step_burst3": round(host_burst * 1e3, 3),
                       "host_blocked_in_hip_ms_per_step": round(max(0.0, host_elapsed / args.steps - host_burst) * 1e3, 3)},
            "roofline": roofline,
            "workload_ss": ss_line,
            "workload_loader": loader_line,
            "workload_refiner": refiner_line,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, scenes)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
