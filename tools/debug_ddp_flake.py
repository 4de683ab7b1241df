"""tests/test_hip_distributed.py::test_stock_ddp_wrapper_sees_finished_gradients failed once in eight runs (worst cosine 0.954 on layer4.0.downsample.0.weight against
a two-run noise sample of 0.9974).  A copied-too-early gradient or the run-to-run spread of this small network?  (a) the spread of N plain runs against each other,
(b) the same comparison under epi_set_deterministic(1), where plain runs are bit-identical and a DDP-wrapped run must be too."""
import copy
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, "/root/repo")
from epipolarpose_amd import hip                                        # noqa: E402
from epipolarpose_amd.core.config import default_config                 # noqa: E402
from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss   # noqa: E402
from epipolarpose_amd.models.pose3d_resnet import get_pose_net          # noqa: E402

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


def cos(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29611", rank=0, world_size=1)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
# (a) the tail of the run-to-run spread, product path alone, non-deterministic mode
hip.set_deterministic(False)
ref = grads(copy.deepcopy(base))
tail = sorted(min((cos(g[k], ref[k]), k) for k in ref) for g in (grads(copy.deepcopy(base)) for _ in range(N)))
print("non-deterministic, %d plain runs against one plain run: min %.6f (%s), 2nd %.6f, median %.6f" % (N, tail[0][0], tail[0][1], tail[1][0], tail[N // 2][0]), flush=True)
ddp = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(base), device_ids=[0], bucket_cap_mb=4)
grads(ddp)
tail = sorted(min((cos(g[k], ref[k]), k) for k in ref) for g in (grads(ddp) for _ in range(N)))
print("non-deterministic, %d DDP-wrapped runs against the plain run: min %.6f (%s), 2nd %.6f, median %.6f" % (N, tail[0][0], tail[0][1], tail[1][0], tail[N // 2][0]), flush=True)
# (b) deterministic mode: every DDP-wrapped run must repeat the first one bit for bit -- a gradient copied before its kernel finished cannot
hip.set_deterministic(True)
ddp = torch.nn.parallel.DistributedDataParallel(copy.deepcopy(base), device_ids=[0], bucket_cap_mb=4)
grads(ddp)
first = grads(ddp)
bad = 0
for r in range(N):
    w = grads(ddp)
    diff = [k for k in first if not torch.equal(w[k], first[k])]
    if diff:
        bad += 1
        print("   deterministic DDP run %d differs from the first in %d parameters: %s" % (r, len(diff), [(k, round(cos(w[k], first[k]), 5)) for k in diff[:4]]), flush=True)
print("deterministic, %d DDP-wrapped runs against the first: %d differ" % (N, bad), flush=True)
hip.set_deterministic(False)
dist.destroy_process_group()
