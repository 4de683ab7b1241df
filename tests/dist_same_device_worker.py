"""Worker of tests/test_hip_distributed.py: one of N ranks (2 or 8) that share cuda:0 (gloo process group), running the real N > 1 training
choreography on the hand-written kernels -- flat per-dtype gradient buckets, learned bucket hooks, grouped weight gradients on the second
stream, deferred slab sums, asynchronous all-reduce handles, FusedAdam on the bucket views.  Started by the test itself, one process per rank
(RANK / WORLD_SIZE in the environment, --init-file = a FileStore rendezvous: no master port to collide on, no launcher agent), or by torch.distributed.run.
Rank 0 also computes, in the same process, what the step must produce: the mean over the two shards of the single-process gradients
(per-shard BatchNorm statistics -- nn.DataParallel semantics, scripts/train.py:93-94) and writes the comparison to --out."""
import argparse
import copy
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--image", type=int, default=64)
    ap.add_argument("--pretrain", type=int, default=8)
    ap.add_argument("--batch", type=int, default=8, help="images per rank (whole 4-view groups)")
    ap.add_argument("--deterministic", type=int, default=1, help="1: the library's deterministic mode (ordered BatchNorm sums) on both ranks and for the "
                    "single-process reference: the comparison is then free of the run-to-run noise of the atomics and can be held tight")
    ap.add_argument("--init-file", default="", help="rendezvous through this file (torch FileStore) instead of MASTER_ADDR / MASTER_PORT")
    a = ap.parse_args()
    from epipolarpose_amd import distributed as epd
    from epipolarpose_amd import hip
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.function import train_step
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from epipolarpose_amd.optim import FusedAdam
    if a.init_file:
        dist.init_process_group("gloo", init_method="file://" + a.init_file, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world, _ = epd.init_from_env(backend="gloo", set_device=False)      # (joins through MASTER_* when no group exists yet)
    assert world >= 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    hip.load()
    if a.deterministic:
        hip.set_deterministic(True)
    j, d, image, b = 4, 16, a.image, a.batch            # per rank: whole 4-view groups
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, d, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = a.layers
    torch.manual_seed(100 + rank)                       # different initial weights per rank: the broadcast has to fix that
    model = get_pose_net(cfg, is_train=True).to(dev)
    model.train()
    crit = SmoothL1JointLocationLoss(num_joints=j)
    gen = torch.Generator().manual_seed(5)
    x_all = torch.randn((world * b, 3, image, image), generator=gen).to(dev)
    gt_all = ((torch.rand((world * b, 3 * j), generator=gen) - 0.5) * 0.4).to(dev)
    wt_all = torch.ones(world * b, 3 * j, device=dev)
    shard = slice(rank * b, (rank + 1) * b)             # whole multi-view groups per rank (epd.shard_groups)
    if rank == 0 and a.pretrain:
        # A few plain optimisation steps first (rank 0 only; the broadcast below hands the result to rank 1): at random initialisation
        # the early-layer gradients of this network are noise in bf16 (two runs of the SAME path agree to cosine ~0.3 on some
        # BatchNorm parameters of ResNet-50, tests/test_hip_precise.py), which would make the comparison below say nothing
        pre = torch.optim.Adam(model.parameters(), lr=1e-3)
        for _ in range(a.pretrain):
            pre.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(x_all)
            crit(out, gt_all, wt_all).backward()
            pre.step()
        del pre
        model.zero_grad(set_to_none=True)
    epd.broadcast_module(model)                         # before the optimizer exists, as scripts/train.py does
    twin = copy.deepcopy(model) if rank == 0 else None  # (before FusedAdam installs its bf16 training copies on the modules)
    opt = FusedAdam(model, lr=1e-3)
    sync = epd.BucketedGradSync(model, optimizer=opt)
    report = {"rank": rank, "buckets": len(sync.buckets), "bucket_dtypes": sorted({str(f.dtype) for f, _, _ in sync.buckets})}
    # reference for the FIRST step, computed by rank 0 before anything moves: mean over the shards of single-process gradients on the
    # per-layer, one-stream, undeferred path (EPI_WGRAD_GROUP=0 semantics)
    ref = None
    if rank == 0:
        g = hip.glue()
        modes = (g.wgrad_group_mode(0), g.wgrad_stream_mode(0), g.defer_wgrad_reduce(False))
        acc = None
        for r in range(world):
            twin.zero_grad(set_to_none=True)
            sl = slice(r * b, (r + 1) * b)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = twin(x_all[sl])
            crit(out, gt_all[sl], wt_all[sl]).backward()
            grads = [p.grad.detach().float().clone() for p in twin.parameters()]
            acc = grads if acc is None else [u + v for u, v in zip(acc, grads)]
        ref = [v / world for v in acc]
        g.wgrad_group_mode(modes[0]); g.wgrad_stream_mode(modes[1]); g.defer_wgrad_reduce(modes[2])
        del twin
    losses = []
    first = None
    for step in range(a.steps):
        if step == 0:
            # one step by hand to look at the synchronised gradients before Adam consumes them
            sync.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = model(x_all[shard])
            loss = crit(out, gt_all[shard], wt_all[shard])
            loss.backward()
            sync.finish()
            first = [p.grad.detach().float().clone() for p in sync.params]
            opt.step()
        else:
            loss = train_step(model, crit, opt, x_all[shard], gt_all[shard], wt_all[shard], grad_sync=sync)
        losses.append(float(loss))
    torch.cuda.synchronize()
    report["losses"] = losses
    report["learning_done"] = not sync._learning
    report["hooks_after"] = len(sync._hooks)
    # parameters must be bit-identical on every rank after the steps (same averaged gradients, same Adam)
    flat = torch.cat([p.detach().float().reshape(-1) for p in model.parameters()])
    mine = flat.cpu()
    other = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(other, mine)
    report["params_identical_across_ranks"] = bool(all(torch.equal(other[0], o) for o in other[1:]))
    if rank == 0:
        cos, nrm = {}, {}
        names = [n for n, p in model.named_parameters() if p.requires_grad]
        for n, got, want in zip(names, first, ref):
            a_, b_ = got.double().reshape(-1), want.double().reshape(-1).to(got.device)
            cos[n] = float(a_ @ b_ / (a_.norm() * b_.norm() + 1e-300))
            nrm[n] = float(a_.norm() / (b_.norm() + 1e-300))
        report["min_cos"] = min(cos.values())
        report["median_cos"] = sorted(cos.values())[len(cos) // 2]
        report["worst"] = sorted(cos.items(), key=lambda kv: kv[1])[:5]
        report["norm_ratio_range"] = [min(nrm.values()), max(nrm.values())]
    with open(a.out + ".rank%d" % rank, "w") as f:
        json.dump(report, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
