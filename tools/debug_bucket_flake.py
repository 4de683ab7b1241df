"""Run-to-run spread of the four-step Adam trajectory of the small ResNet-18 used by tests/test_hip_step_in_backward.py, plain against plain
and plain against the bucket path, from a random and from a pre-trained state (the yardstick of the test was calibrated with this)."""
import copy, sys, torch
sys.path.insert(0, '/root/repo')
from epipolarpose_amd.core.config import default_config
from epipolarpose_amd.core.function import train_step
from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
from epipolarpose_amd.distributed import BucketedGradSync
from epipolarpose_amd.models.pose3d_resnet import get_pose_net
from epipolarpose_amd.optim import FusedAdam
dev = torch.device("cuda:0")
size, batch, pre, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = default_config(); cfg.MODEL.INIT_WEIGHTS = False; cfg.MODEL.EXTRA.NUM_LAYERS = 18
j = 4
cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, 16, [size, size]
torch.manual_seed(2)
base = get_pose_net(cfg, is_train=False).to(dev).train()
data = [torch.randn(batch, 3, size, size, device=dev) for _ in range(4)]
gt = (torch.rand(batch, 3 * j, device=dev) - 0.5) * 0.4
vis = torch.ones(batch, 3 * j, device=dev)
crit = SmoothL1JointLocationLoss(num_joints=j)
if pre:
    warm = copy.deepcopy(base)
    opt = FusedAdam(warm, lr=1e-2)
    for i in range(pre):
        train_step(warm, crit, opt, data[i % 4], gt, vis)
    torch.cuda.synchronize()
    base.load_state_dict(warm.state_dict())
def run(bucketed):
    m = copy.deepcopy(base)
    opt = FusedAdam(m, lr=1e-2)
    sync = BucketedGradSync(m, optimizer=opt, bucket_bytes=4 << 20) if bucketed else None
    for x in data:
        train_step(m, crit, opt, x, gt, vis, grad_sync=sync)
    torch.cuda.synchronize()
    return {k: v.detach().float().clone() for k, v in m.named_parameters()}
start = {k: v.detach().float() for k, v in base.named_parameters()}
plain = [run(False) for _ in range(iters)]
buck = [run(True) for _ in range(iters)]
step = sum(float((plain[0][k] - start[k]).abs().sum()) for k in start)
d = lambda u, v: sum(float((u[k] - v[k]).abs().sum()) for k in u) / step
pp = sorted(d(plain[i], plain[k]) for i in range(iters) for k in range(i))
pb = sorted(d(p, b) for p in plain for b in buck)
q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
print("size %d batch %d pre %d: plain-plain med %.3f p90 %.3f max %.3f | plain-bucket med %.3f p90 %.3f max %.3f" % (
    size, batch, pre, q(pp, .5), q(pp, .9), pp[-1], q(pb, .5), q(pb, .9), pb[-1]), flush=True)
