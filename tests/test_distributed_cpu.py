"""CPU, world_size 2 over gloo: the bucketed gradient all-reduce reproduces the single-process gradient of the
global batch (loss = sum / B_local per rank, then mean over ranks == sum / B_global)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(nn.Sequential):
    """The last module ("unused") takes no part in forward: its parameters receive no gradient (must come out as zeros)."""

    def forward(self, x):
        for m in list(self)[:-1]:
            x = m(x)
        return x


def _make_model():
    torch.manual_seed(0)
    # channels_last weights incl. a 1x1 convolution: its gradient may come back with the strides of the contiguous form
    return _Net(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.Flatten(),
                nn.Linear(8 * 6 * 6, 5), nn.Linear(4, 4)).to(memory_format=torch.channels_last)


def _loss(model, x, y):
    return ((model(x) - y) ** 2).sum() / x.shape[0]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from epipolarpose_amd import distributed as epd
    r, w, _ = epd.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    model = _make_model()
    if rank == 1:                                    # rank 1 starts from different weights: broadcast must fix it
        for p in model.parameters():
            p.data.add_(1.0)
    epd.broadcast_module(model)
    sync = epd.BucketedGradSync(model, bucket_bytes=2048)     # small buckets -> several collectives
    assert len(sync.buckets) > 1
    torch.manual_seed(42)
    x, y = torch.randn(8, 3, 6, 6), torch.randn(8, 5)
    start, n = epd.shard_groups(4, rank, world)               # 4 groups of 2 "views"
    xs, ys = x[2 * start:2 * (start + n)], y[2 * start:2 * (start + n)]
    for _ in range(3):                                        # three times: all hooks, then the learned per-bucket hooks
        sync.zero_grad()
        assert all(p.grad is None for p in model.parameters())          # autograd hands gradients over, no accumulate kernels
        _loss(model, xs, ys).backward()
        sync.finish()
    for flat, plist, _ in sync.buckets:                      # every .grad now lives inside its flat all-reduce bucket
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        assert all(lo <= p.grad.data_ptr() < hi for p in plist)
    torch.save([p.grad.clone() for p in model.parameters()], os.path.join(tmp, "g%d.pt" % rank))
    assert sync._learning                                     # the unused layer's bucket never completes: every hook stays
    # second model, every parameter used: after the first step only one hook per bucket remains
    model2 = nn.Sequential(*list(_make_model())[:-1])
    sync2 = epd.BucketedGradSync(model2, bucket_bytes=2048)
    n_hooks_first = len(sync2._hooks)
    for _ in range(3):
        sync2.zero_grad()
        _loss(model2, xs, ys).backward()
        sync2.finish()
    assert n_hooks_first == len(list(model2.parameters())) and not sync2._learning and len(sync2._hooks) == len(sync2.buckets)
    torch.save([p.grad.clone() for p in model2.parameters()], os.path.join(tmp, "h%d.pt" % rank))
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_global_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0 = torch.load(tmp_path / "g0.pt")
    g1 = torch.load(tmp_path / "g1.pt")
    model = _make_model()
    torch.manual_seed(42)
    x, y = torch.randn(8, 3, 6, 6), torch.randn(8, 5)
    _loss(model, x, y).backward()
    for a, b, p in zip(g0, g1, model.parameters()):
        assert torch.equal(a, b)
        want = p.grad if p.grad is not None else torch.zeros_like(p)      # the unused layer
        torch.testing.assert_close(a, want, rtol=1e-5, atol=1e-6)
    h0 = torch.load(tmp_path / "h0.pt")
    h1 = torch.load(tmp_path / "h1.pt")
    for a, b, p in zip(h0, h1, list(model.parameters())[:-2]):             # learned-hook mode: same gradients
        assert torch.equal(a, b)
        torch.testing.assert_close(a, p.grad, rtol=1e-5, atol=1e-6)


class _MixedNet(nn.Module):
    """fp32 / bf16 parameters alternating layer by layer, as FusedAdam's bf16 training copies make them in the real network."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.a = nn.Linear(6, 16)
        self.b = nn.Linear(16, 16).to(torch.bfloat16)
        self.c = nn.Linear(16, 16)
        self.d = nn.Linear(16, 16).to(torch.bfloat16)
        self.e = nn.Linear(16, 4)

    def forward(self, x):
        x = torch.relu(self.a(x))
        x = torch.relu(self.b(x.to(torch.bfloat16))).float()
        x = torch.relu(self.c(x))
        x = torch.relu(self.d(x.to(torch.bfloat16))).float()
        return self.e(x)


def _mixed_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from epipolarpose_amd import distributed as epd
    epd.init_from_env(backend="gloo")
    model = _MixedNet()
    sync = epd.BucketedGradSync(model, bucket_bytes=1 << 20)
    # one chain of buckets per dtype, NOT one bucket per dtype change (10 parameters alternate dtype 4 times)
    assert len(sync.buckets) == 2 and {f.dtype for f, _, _ in sync.buckets} == {torch.float32, torch.bfloat16}
    torch.manual_seed(7)
    x, y = torch.randn(8, 6), torch.randn(8, 4)
    xs, ys = x[4 * rank:4 * rank + 4], y[4 * rank:4 * rank + 4]
    for _ in range(3):
        sync.zero_grad()
        (((model(xs) - ys) ** 2).sum() / 4).backward()
        sync.finish()
    torch.save([p.grad.clone() for p in model.parameters()], os.path.join(tmp, "m%d.pt" % rank))
    dist.destroy_process_group()


def test_mixed_dtype_buckets_allreduce(tmp_path):
    """world 2, gloo: bf16 and fp32 gradients travel in their own bucket chains and both come out as the rank mean."""
    port = _free_port()
    mp.spawn(_mixed_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = torch.load(tmp_path / "m0.pt"), torch.load(tmp_path / "m1.pt")
    model = _MixedNet()
    torch.manual_seed(7)
    x, y = torch.randn(8, 6), torch.randn(8, 4)
    per_rank = []
    for r in range(2):
        model.zero_grad()
        (((model(x[4 * r:4 * r + 4]) - y[4 * r:4 * r + 4]) ** 2).sum() / 4).backward()
        per_rank.append([p.grad.clone() for p in model.parameters()])
    for a, b, w0, w1 in zip(g0, g1, *per_rank):
        assert torch.equal(a, b)
        want = (w0.float() + w1.float()) / 2
        tol = dict(rtol=2e-2, atol=2e-3) if a.dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(a.float(), want, **tol)


def test_resnet50_bucket_count_with_bf16_training_copies():
    """The real ResNet-50 with bf16 training copies of the convolution weights (FusedAdam's default): conv (bf16) and BatchNorm
    (fp32) parameters alternate through the whole network; the gradient path must still be a handful of large buckets
    (round-1 defect: 106 buckets, median 18.8 KB)."""
    from epipolarpose_amd import distributed as epd
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    model = get_pose_net(cfg, True)
    params = [p for p in model.parameters() if p.requires_grad]
    index = {id(p): i for i, p in enumerate(params)}

    class _Copies:
        def training_copies(self):
            out = {}
            for mod in model.modules():
                if (type(mod) is nn.Conv2d or getattr(mod, "supports_training_copy", False)) and mod.bias is None:
                    out[index[id(mod.weight)]] = mod.weight.detach().to(torch.bfloat16).requires_grad_(True)
            return out
    sync = epd.BucketedGradSync(model, optimizer=_Copies())
    sizes = [f.numel() * f.element_size() for f, _, _ in sync.buckets]
    assert len(sync.buckets) <= 6, sizes
    assert {f.dtype for f, _, _ in sync.buckets} == {torch.float32, torch.bfloat16}
    assert sum(len(pl) for _, pl, _ in sync.buckets) == len(params)
    assert len(epd.BucketedGradSync(model).buckets) <= 6          # pure fp32 gradients


def test_bf16_bucket_accumulation_error_at_eight_ranks():
    """Decision record for the dtype of the gradient buckets: the convolution-weight gradients travel in bf16 (they are
    produced in bf16, on the bf16 training copies) and RCCL's ring adds them hop by hop, rounding to bf16 after every hop.  Simulated here
    for N = 8 on gradients of mixed magnitude: the ring sum's error against the exact mean is ~2x one bf16 rounding (which the gradient
    carries anyway) -- RMS <= 0.6 % of the RMS gradient, cosine >= 0.9999 -- while fp32 buckets would double the 47 MB that leave per
    step.  bf16 buckets stay; what an Adam step sees (g / sqrt(v)) moves by the same fraction of a per-cent."""
    torch.manual_seed(0)
    n, m = 8, 1 << 18
    scale = torch.exp(torch.randn(m) * 2.0)                       # elements spanning orders of magnitude
    g = [(torch.randn(m) * scale).to(torch.bfloat16) for _ in range(n)]
    exact = sum(t.double() for t in g) / n
    acc = g[0].clone()
    for t in g[1:]:                                               # one hop of the ring: fp32 add, bf16 store
        acc = (acc.float() + t.float()).to(torch.bfloat16)
    ring = (acc.float() / n).to(torch.bfloat16).double()
    one = (exact.float()).to(torch.bfloat16).double()             # a single rounding of the exact mean: the floor of any bf16 result
    rel = lambda a: float(((a - exact) ** 2).mean().sqrt() / (exact ** 2).mean().sqrt())
    cos = float(ring @ exact / (ring.norm() * exact.norm()))
    assert rel(one) <= 2.5e-3 and rel(ring) <= 6e-3 and rel(ring) <= 3.0 * rel(one), (rel(one), rel(ring))
    assert cos >= 0.9999, cos


def _r50_grad_worker(rank, world, port, tmp):
    """One of EIGHT gloo ranks: real ResNet-50 gradients (the oracle network on this rank's own seeded batch, fp32 on the CPU), handed to
    BucketedGradSync in the dtypes of the product path (convolution weights bf16, everything else fp32), all-reduced through its flat buckets."""
    import numpy as np
    from epipolarpose_amd import distributed as epd
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from oracle import network as o_net
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    j, d, image, b = 4, 16, 64, 2
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, d, [image, image]
    torch.manual_seed(1)                                       # the same weights on every rank
    model = get_pose_net(cfg, True)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    for n in names:
        sd[n].requires_grad_(True)
    gen = torch.Generator().manual_seed(100 + rank)            # this rank's shard
    x = torch.randn((b, 3, image, image), generator=gen)
    gt = (torch.rand((b, 3 * j), generator=gen) - 0.5) * 0.4
    loss = o_net.joint_location_loss(o_net.forward(sd, x, 50, training=True, new_stats={}), gt, torch.ones(b, 3 * j), j, "smoothl1")
    loss.backward()
    params = [p for p in model.parameters() if p.requires_grad]
    index = {id(p): i for i, p in enumerate(params)}

    class _Copies:                                             # FusedAdam's bf16 training copies of the convolution weights
        def training_copies(self):
            out = {}
            for mod in model.modules():
                if (type(mod) is nn.Conv2d or getattr(mod, "supports_training_copy", False)) and mod.bias is None:
                    out[index[id(mod.weight)]] = mod.weight.detach().to(torch.bfloat16).requires_grad_(True)
            return out
    sync = epd.BucketedGradSync(model, optimizer=_Copies())
    sync.zero_grad()
    mine = []
    for n, p in zip(names, sync.params):                       # gradients arrive in the dtype of the parameter they belong to
        g = sd[n].grad.to(p.dtype)
        p.grad = g.clone()
        mine.append(g.double().reshape(-1))                    # what this rank contributes (already rounded to bf16 where the product rounds)
    sync.finish()
    got = torch.cat([p.grad.double().reshape(-1) for p in sync.params])
    exact = torch.cat(mine)
    dist.all_reduce(exact)                                     # float64 sum of the same contributions
    exact /= world
    if rank == 0:
        is_bf16 = torch.cat([torch.full((p.numel(),), p.dtype == torch.bfloat16) for p in sync.params])
        rep = {}
        for tag, m in (("bf16", is_bf16), ("fp32", ~is_bf16)):
            a, e = got[m], exact[m]
            rep[tag] = {"n": int(m.sum()), "rel_rms": float(((a - e) ** 2).mean().sqrt() / (e ** 2).mean().sqrt()),
                        "cos": float(a @ e / (a.norm() * e.norm())),
                        "rel_rms_single_rounding": float(((e.float().to(torch.bfloat16).double() - e) ** 2).mean().sqrt() / (e ** 2).mean().sqrt())}
        rep["buckets"] = len(sync.buckets)
        # per-tensor cosine: no gradient tensor of the network may be hurt, not only the concatenation
        worst, off = 1.0, 0
        for n, p in zip(names, sync.params):
            a, e = got[off:off + p.numel()], exact[off:off + p.numel()]
            off += p.numel()
            if float(e.norm()) > 0:
                worst = min(worst, float(a @ e / (a.norm() * e.norm())))
        rep["worst_tensor_cos"] = worst
        import json
        with open(os.path.join(tmp, "r50_bucket_sum.json"), "w") as f:
            json.dump(rep, f)
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_buckets_eight_gloo_ranks_real_resnet50_gradients(tmp_path):
    """the bf16 bucket sum over EIGHT ranks on the REAL ResNet-50 gradient set (161 tensors, 34 M elements, the
    magnitudes a backward pass produces -- not synthetic ones) through the real BucketedGradSync / gloo all-reduce: against the float64 mean
    of the same contributions the bf16 buckets stay within ~3 roundings of a bf16 value, the fp32 buckets at fp32 accuracy, and no single
    tensor's direction moves."""
    import json
    world = 8
    mp.spawn(_r50_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rep = json.load(open(os.path.join(str(tmp_path), "r50_bucket_sum.json")))
    assert rep["buckets"] <= 6 and rep["bf16"]["n"] > 30_000_000 and rep["fp32"]["n"] > 50_000
    assert rep["fp32"]["rel_rms"] <= 1e-6 and rep["fp32"]["cos"] >= 1 - 1e-9, rep
    assert rep["bf16"]["rel_rms"] <= 8e-3 and rep["bf16"]["rel_rms"] <= 3.5 * rep["bf16"]["rel_rms_single_rounding"], rep
    assert rep["bf16"]["cos"] >= 0.9999 and rep["worst_tensor_cos"] >= 0.999, rep


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment (how the round driver calls it) must start two ranks under
    torch.distributed.run by itself: both come up, join the process group (gloo here) and reach the device assert -- which is where a box
    without GPUs stops them.  (Reference semantics: N replicas from ONE command, scripts/train.py:93-94,143.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CUDA_VISIBLE_DEVICES"] = ""          # (a GPU box must stop at the same place as this container)
    env["HIP_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    err = r.stderr
    assert "torch.distributed.run" in err and "--nproc-per-node 2" in err, err[-2000:]
    assert "rank 0/2 up, backend gloo" in err and "rank 1/2 up, backend gloo" in err, err[-2000:]
    assert "bench.py needs MI355X GPUs" in err, err[-2000:]
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in err          # the round-3 usage error is gone


def _direct_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from epipolarpose_amd import distributed as epd
    epd.init_from_env(backend="gloo")
    torch.manual_seed(7)
    x, y = torch.randn(4 * world, 6), torch.randn(4 * world, 4)
    xs, ys = x[4 * rank:4 * rank + 4], y[4 * rank:4 * rank + 4]
    out = {}
    # every rank's LOCAL gradients (no exchange), gathered: the exact mean the collectives have to deliver
    local = _MixedNet()
    (((local(xs) - ys) ** 2).sum() / 4).backward()
    mine = torch.cat([p.grad.double().reshape(-1) for p in local.parameters()])
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    out["exact_mean"] = torch.stack(every).mean(0)
    for mode in ("allreduce", "direct"):
        model = _MixedNet()
        sync = epd.BucketedGradSync(model, bucket_bytes=1 << 9, collective=mode)     # small buckets: several collectives per dtype, ragged sizes
        assert len(sync.buckets) >= 3
        if mode == "direct":
            assert all(f.numel() % world == 0 for f, _, _ in sync.buckets)
        for _ in range(3):
            sync.zero_grad()
            (((model(xs) - ys) ** 2).sum() / 4).backward()
            sync.finish()
        out[mode] = [p.grad.clone() for p in model.parameters()]
    torch.save(out, os.path.join(tmp, "d%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_direct_collective_matches_allreduce(tmp_path, world):
    """the "direct" gradient collective (reduce-scatter + all-gather as grouped point-to-point transfers, one per peer = one per
    xGMI link) against the library all-reduce on the same gradients -- gloo, 2 and 8 ranks, fp32 and bf16 bucket chains with ragged bucket sizes: identical
    on every rank, equal to the all-reduce result to the rounding of the sum order (fp32 accumulation of the bf16 slices in the direct form)."""
    mp.spawn(_direct_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / ("d%d.pt" % r)) for r in range(world)]
    for r in range(1, world):
        for a, b in zip(res[0]["direct"], res[r]["direct"]):
            assert torch.equal(a, b)
    for a, b in zip(res[0]["direct"], res[0]["allreduce"]):
        tol = dict(rtol=2e-2, atol=2e-3) if a.dtype == torch.bfloat16 else dict(rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(a.float(), b.float(), **tol)
    # round 6: the direct form adds the N slices in fp32, scales by 1 / N and rounds ONCE -- a bf16 gradient is the exact mean of the (bf16) local gradients
    # rounded to bf16 (to one unit in the last place: the fp32 sum itself rounds), not the twice-rounded value of round 5; fp32 gradients to fp32 accuracy
    exact, off = res[0]["exact_mean"], 0
    for g in res[0]["direct"]:
        e = exact[off:off + g.numel()].reshape(g.shape)
        off += g.numel()
        if g.dtype == torch.bfloat16:
            half_ulp = e.abs() * 2.0 ** -8            # (round to nearest: at most 2^-8 relative -- reached just above a power of two; two roundings could double it)
            assert ((g.double() - e).abs() <= 1.001 * half_ulp + 1e-12).all(), float(((g.double() - e).abs() / (half_ulp + 1e-300)).max())
        else:
            torch.testing.assert_close(g.double(), e, rtol=1e-5, atol=1e-7)
