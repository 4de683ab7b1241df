"""Diagnostic (GPU): at the bench configuration (ResNet-50, 256x256, batch 32, D = 64) -- is it the bf16 product path or the fp32-grade mode
that leaves the other?  Third opinion: the oracle network on STOCK PyTorch-ROCm fp32 kernels (TF32 off).  Prints the fp32-grade loss trajectory
and, at steps 0 / 10 / 30, the three losses and gradient cosines of a few tensors."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from det_weights import seeded_array  # noqa: E402


def cosine(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float((a @ b) / (a.norm() * b.norm() + 1e-300))


def main():
    from epipolarpose_amd.core.config import default_config
    from epipolarpose_amd.core.integral_loss import SmoothL1JointLocationLoss
    from epipolarpose_amd.models import precise
    from epipolarpose_amd.models.pose3d_resnet import get_pose_net
    from oracle import network as o_net
    layers, image, j, d, b = 50, 256, 17, 64, int(os.environ.get("DIAG_BATCH", "32"))
    dev = torch.device("cuda:0")
    cfg = default_config()
    cfg.MODEL.INIT_WEIGHTS = False
    cfg.MODEL.NUM_JOINTS, cfg.MODEL.DEPTH_RES, cfg.MODEL.IMAGE_SIZE = j, d, [image, image]
    cfg.MODEL.EXTRA.NUM_LAYERS = layers
    torch.manual_seed(7)
    model = get_pose_net(cfg, is_train=True).to(dev)
    init = model.state_dict()
    for k, v in init.items():
        if v.dim() == 4 and (k.startswith("deconv_layers") or k.startswith("final_layer")):
            v.normal_(0, 0.001)
    sd = {k: v.detach().clone().float() if v.dtype.is_floating_point else v.detach().clone() for k, v in init.items()}
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    x = torch.from_numpy(seeded_array("img/trained", (b, 3, image, image))).to(dev)
    gt = torch.from_numpy(seeded_array("gt/trained", (b, 3 * j), scale=0.2)).to(dev)
    wt = torch.ones(b, 3 * j, device=dev)
    crit = SmoothL1JointLocationLoss(num_joints=j)
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-3)
    keys = ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv2.weight", "layer3.0.conv2.weight", "layer4.2.conv3.weight", "layer4.2.bn2.bias",
            "deconv_layers.0.weight", "deconv_layers.6.weight", "final_layer.weight", "final_layer.bias")

    def compare(tag):
        state = {k: v.detach().clone() for k, v in sd.items()}
        opt.zero_grad()
        lp = crit(precise.forward(sd, x, layers, training=True), gt, wt)
        lp.backward()
        gp = {k: sd[k].grad.detach().clone() for k in keys}
        # stock fp32
        s32 = {k: v.detach().clone() for k, v in state.items()}
        for k in s32:
            if s32[k].dtype.is_floating_point and "running" not in k:
                s32[k].requires_grad_(True)
        prev = (torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32)
        torch.backends.cudnn.allow_tf32 = torch.backends.cuda.matmul.allow_tf32 = False
        ls = o_net.joint_location_loss(o_net.forward(s32, x, layers, training=True, new_stats={}).float(), gt, wt, j, "smoothl1")
        ls.backward()
        torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = prev
        model.load_state_dict(state)
        model.train()
        model.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lb = crit(model(x), gt, wt)
        lb.backward()
        gb = dict(model.named_parameters())
        print("%s: loss precise %.6f  stock-fp32 %.6f  bf16 %.6f" % (tag, lp.item(), ls.item(), lb.item()), flush=True)
        for k in keys:
            print("    %-26s cos(precise, stock32) %.5f   cos(bf16, stock32) %.5f   cos(bf16, precise) %.5f   |g| %.3e" %
                  (k, cosine(gp[k], s32[k].grad), cosine(gb[k].grad.float(), s32[k].grad), cosine(gb[k].grad.float(), gp[k]), float(s32[k].grad.norm())), flush=True)
        opt.zero_grad()

    compare("step 0")
    for step in range(30):
        opt.zero_grad()
        loss = crit(precise.forward(sd, x, layers, training=True), gt, wt)
        loss.backward()
        opt.step()
        print("precise step %d loss %.6f" % (step, loss.item()), flush=True)
        if step + 1 in (10, 30):
            compare("after %d steps" % (step + 1))


if __name__ == "__main__":
    main()
