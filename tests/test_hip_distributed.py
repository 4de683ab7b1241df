"""GPU: the N > 1 training choreography on ONE device -- two ranks (gloo) sharing cuda:0 run the real bucketed gradient path on the
hand-written kernels: per-dtype flat buckets, learned bucket hooks that flush the deferred slab sums and the grouped weight gradients,
the second stream, asynchronous all-reduce handles, FusedAdam on the bucket views (reference: nn.DataParallel, scripts/train.py:93-94,
loss normalisation lib/core/integral_loss.py:29,45).  The synchronised gradients of the first step are compared with the mean over the two
shards of single-process gradients; the parameters must be bit-identical on both ranks after three steps."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(world, tmp_path, worker_args, timeout, extra_env=None):
    """One process per rank started directly (RANK / WORLD_SIZE / LOCAL_RANK in the environment), rendezvous through a FileStore in tmp_path: no master
    port that another socket could take between choosing and binding it, no launcher agent process.  -> (return codes, tail of every rank's output)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", WORLD_SIZE=str(world), **(extra_env or {}))
    store = str(tmp_path / "rendezvous")
    procs, logs = [], []
    for rank in range(world):
        log = open(str(tmp_path / ("rank%d.log" % rank)), "w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_same_device_worker.py"), "--init-file", store] + worker_args,
                                      env=dict(env, RANK=str(rank), LOCAL_RANK=str(rank)), cwd=ROOT, stdout=log, stderr=subprocess.STDOUT))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=timeout))
    finally:
        for p in procs:                         # (a rank that died leaves the others in a collective: end exactly the processes started here)
            if p.poll() is None:
                p.kill()
                p.wait()
    tails = []
    for rank, log in enumerate(logs):
        log.seek(0)
        tails.append("---- rank %d ----\n%s" % (rank, log.read()[-2500:]))
        log.close()
    return codes, "\n".join(tails)


FLOORS = {18: (0.999, 0.9999), 50: (0.999, 0.9999)}     # (min, median) cosine over all parameters; measured 0.999996 / 1.0 for both (call r04f)


@pytest.mark.parametrize("layers", [18, 50])
def test_two_ranks_on_one_device_bucketed_path(tmp_path, layers):
    out = str(tmp_path / "report.json")
    codes, tail = _run_ranks(2, tmp_path, ["--out", out, "--layers", str(layers), "--image", "64" if layers == 18 else "128"], timeout=600)
    assert codes == [0, 0], tail
    r0, r1 = (json.load(open(out + ".rank%d" % r)) for r in range(2))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_device_r%d.json" % layers), "w") as f:
        json.dump([r0, r1], f, indent=1)
    assert r0["params_identical_across_ranks"] and r1["params_identical_across_ranks"]
    assert r0["learning_done"] and r0["hooks_after"] == r0["buckets"] <= 6          # one learned hook per bucket after the first step
    assert set(r0["bucket_dtypes"]) == {"torch.float32", "torch.bfloat16"}
    # Both ranks and the single-process reference run in the library's deterministic mode (ordered BatchNorm sums, round 4): what is left between
    # the two is the fp32 summation order of the weight gradients (grouped launches vs one per layer) and the bf16 sum of the two ranks' bucket
    # contents.  Rounds 2-3 ran with atomics and had to accept the run-to-run noise of 50 bf16 layers at batch 8 (ResNet-50: min 0.77 / median
    # 0.92); a bucket that left before its gradient was final, a missing 1/N or a dropped shard is O(1).
    lo, med = FLOORS[layers]
    assert r0["min_cos"] >= lo and r0["median_cos"] >= med, (r0["median_cos"], r0["worst"])
    assert 0.8 <= r0["norm_ratio_range"][0] and r0["norm_ratio_range"][1] <= 1.25, r0["norm_ratio_range"]
    assert all(l == l and l < 10 for l in r0["losses"] + r1["losses"])


def test_eight_ranks_on_one_device_bucketed_path(tmp_path):
    """the BASELINE configs[3] choreography -- EIGHT ranks, one 4-view group (batch 4) per rank, ResNet-18 -- on one device
    through the real path (per-dtype buckets, learned hooks, second stream, deferred sums, asynchronous handles, FusedAdam on the bucket views; gloo
    carries the collectives): parameters bit-identical on all eight ranks after three steps, the synchronised gradients of step 1 = the mean over the eight
    shards of single-process gradients."""
    out = str(tmp_path / "report.json")
    codes, tail = _run_ranks(8, tmp_path, ["--out", out, "--layers", "18", "--image", "64", "--batch", "4"], timeout=900, extra_env={"OMP_NUM_THREADS": "1"})
    assert codes == [0] * 8, tail
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    reps = [json.load(open(out + ".rank%d" % r)) for r in range(8)]
    with open(os.path.join(ROOT, "gpurun_out", "eight_ranks_one_device_r18.json"), "w") as f:
        json.dump(reps, f, indent=1)
    assert all(r["params_identical_across_ranks"] for r in reps)
    r0 = reps[0]
    assert r0["learning_done"] and r0["hooks_after"] == r0["buckets"] <= 6
    # (batch 4 per rank: BatchNorm statistics over 4 images; the bf16 bucket sum of eight ranks adds ~3 bits of rounding to the two-rank case)
    assert r0["min_cos"] >= 0.995 and r0["median_cos"] >= 0.9995, (r0["min_cos"], r0["median_cos"], r0["worst"])
    assert 0.8 <= r0["norm_ratio_range"][0] and r0["norm_ratio_range"][1] <= 1.25, r0["norm_ratio_range"]
    assert all(l == l and l < 10 for r in reps for l in r["losses"])


def test_stock_ddp_wrapper_sees_finished_gradients(tmp_path):
    """torch's DistributedDataParallel registers its reducer as a post-hook of every AccumulateGrad node and copies the gradient
    into its buckets INSIDE the backward pass.  A weight gradient that is still in flight on the second stream (or unreduced in slabs) at that moment would
    be copied half-written.  The glue looks at acc->post_hooks() (gradient_consumed_after_backward) and keeps such gradients on the main stream with an
    immediate reduction: a DDP-wrapped network (world size 1, gloo, fp32 master weights used directly) must produce the gradients of the unwrapped one."""
    import copy
    import torch
    import torch.distributed as dist
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    dev = torch.device("cuda:0")
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.EXTRA.NUM_LAYERS = 18
    j = 4
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, 16, [128, 128]
    torch.manual_seed(11)
    base = get_pose_net(cfg, is_train=False).to(dev).train()
    x = torch.randn(16, 3, 128, 128, device=dev)
    gt = (torch.rand(16, 3 * j, device=dev) - 0.5) * 0.4
    vis = torch.ones(16, 3 * j, device=dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)
    # a well-conditioned state first (tests/test_hip_step_in_backward.py: at random initialisation two runs of the same code differ by tens of per cent)
    opt = torch.optim.Adam(base.parameters(), lr=1e-2)
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            crit(base(x), gt, vis).backward()
        opt.step()
    torch.cuda.synchronize()

    def grads(m):
        for p in m.parameters():
            p.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            crit(m(x), gt, vis).backward()
        torch.cuda.synchronize()
        return {k.replace("module.", ""): p.grad.detach().float().clone() for k, p in m.named_parameters()}

    from epipolarpose_amd import hip

    def cos(a, b):
        a, b = a.reshape(-1).double(), b.reshape(-1).double()
        return float(a @ b / (a.norm() * b.norm() + 1e-300))

    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", init_method="file://" + str(tmp_path / "rendezvous"), rank=0, world_size=1)
    # Judged in the library's deterministic mode only (ordered sums): plain runs repeat bit for bit there, so a DDP-wrapped run must repeat too -- a
    # gradient copied while its kernel or its slab sum was still running is garbage or half a sum and cannot (tools/debug_ddp_flake.py: 150 of 150 runs
    # identical).  (Rounds 2-5 also compared the atomics mode against a plain run of itself; that arm measured the run-to-run noise of the atomics, not the
    # wrapper, and is gone.)
    det_before = hip.set_deterministic(True)
    try:
        det_plain = grads(copy.deepcopy(base))
        assert all(torch.equal(det_plain[k], v) for k, v in grads(copy.deepcopy(base)).items())
        ddp_det = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(base), device_ids=[0], bucket_cap_mb=4)
        grads(ddp_det)                   # (the first pass builds the reducer's bucket order)
        det_first = grads(ddp_det)
        assert set(det_first) == set(det_plain)
        for _ in range(4):
            again = grads(ddp_det)
            differ = [k for k in det_first if not torch.equal(again[k], det_first[k])]
            assert not differ, differ[:5]
        # (the wrapped network reduces every weight gradient at once on the main stream, the plain one in grouped launches: another summation order)
        det_worst = min((cos(det_first[k], det_plain[k]), k) for k in det_plain)
        assert det_worst[0] >= 0.9999, det_worst
        for k in det_plain:
            ratio = float(det_first[k].norm() / (det_plain[k].norm() + 1e-30))
            assert 0.99 <= ratio <= 1.01, (k, ratio)
    finally:
        hip.set_deterministic(bool(det_before))
        if created:
            dist.destroy_process_group()
