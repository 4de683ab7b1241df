#!/usr/bin/env python
"""Build container only (imports /root/reference): how well can ANY bf16 run reproduce the fp32 gradients of the reference network?
Trains the reference PoseResNet in fp32 on one fixed synthetic batch and, every few steps, compares the gradients of a bf16-autocast
backward pass with the fp32 ones (cosine per parameter).  Used to choose the state at which tests/golden/network_trained.npz is taken
(at random initialisation the early-layer cosines are 0.07 .. 0.1 for every bf16 implementation).
    python tools/probe_conditioning.py --layers 50 --image 128 --batch 8 --steps 40"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, ROOT)
import make_golden  # noqa: E402  (loads the reference through tests/golden/ref_shims.py)
from make_golden import ref_cfg  # noqa: E402
from epipolarpose_amd.synthetic import SyntheticScenes  # noqa: E402

REF = make_golden.REF
KEYS = ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.conv2.weight", "layer3.0.conv2.weight", "layer4.1.conv1.weight",
        "deconv_layers.0.weight", "deconv_layers.6.weight", "final_layer.weight", "bn1.weight", "layer3.1.bn2.bias")


def cos(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--image", type=int, default=128)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--joints", type=int, default=17)
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--every", type=int, default=5)
    ap.add_argument("--lr", type=float, default=1e-3)
    a = ap.parse_args()
    torch.set_num_threads(8)
    torch.manual_seed(0)
    model = REF.pose3d_resnet.get_pose_net(ref_cfg(a.layers, a.image, a.joints, a.depth), is_train=True)
    # the reference's own head initialisation (init_weights, pose3d_resnet.py:222-239) without the checkpoint it then insists on loading
    for name, m in list(model.deconv_layers.named_modules()) + [("final", model.final_layer)]:
        if isinstance(m, (torch.nn.ConvTranspose2d, torch.nn.Conv2d)):
            torch.nn.init.normal_(m.weight, std=0.001)
            if m.bias is not None:
                torch.nn.init.constant_(m.bias, 0)
    model.train()
    sc = SyntheticScenes(n_group=a.batch // 4, n_view=4, num_joints=a.joints, seed=11)
    x = torch.from_numpy(sc.images(size=a.image))
    gt, wt = torch.from_numpy(sc.label), torch.from_numpy(sc.weight)
    crit = REF.integral_loss.SmoothL1JointLocationLoss(num_joints=a.joints)
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    names = dict(model.named_parameters())

    def grads(autocast):
        model.zero_grad()
        saved = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            logits = model(x)
        loss = crit(logits.float(), gt, wt)
        loss.backward()
        model.load_state_dict(saved, strict=False)          # a probe must not advance the running statistics
        return float(loss), float(logits.float().abs().max()), {k: names[k].grad.clone() for k in KEYS}
    for step in range(a.steps + 1):
        if step % a.every == 0:
            t0 = time.time()
            l32, m32, g32 = grads(False)
            l16, m16, g16 = grads(True)
            print("step %3d  loss fp32 %.6f bf16 %.6f  max|logit| %.3g  cos: %s  (%.0f s)" % (
                step, l32, l16, m32, " ".join("%s=%.3f" % (k.replace(".weight", "").replace("layer", "l").replace("deconv_layers", "dc"), cos(g16[k], g32[k]))
                                              for k in KEYS), time.time() - t0), flush=True)
        if step == a.steps:
            break
        opt.zero_grad()
        loss = crit(model(x), gt, wt)
        loss.backward()
        opt.step()


if __name__ == "__main__":
    main()
