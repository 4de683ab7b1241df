"""Case tables shared by make_golden.py (generator) and the tests (consumers)."""
DLOGITS_STRIDE = 13     # big cases store every 13th gradient element ...
DLOGITS_STRIDE_HUGE = 509   # ... and the config-shape case (4.46 M logits) every 509th


def dlogits_stride(size):
    """Sub-sampling of a stored gradient by the size of the logits tensor (generator and tests share the rule)."""
    return DLOGITS_STRIDE if size <= 1000000 else DLOGITS_STRIDE_HUGE


INTEGRAL_CASES = [  # name, B, J, D, H, W, logit scale
    ("tiny", 2, 3, 8, 8, 8, 3.0),
    ("rect", 3, 4, 4, 8, 16, 2.0),
    ("mid", 2, 17, 16, 16, 16, 4.0),
    ("cube32", 1, 2, 32, 32, 32, 6.0),
    ("cfg", 1, 17, 64, 64, 64, 4.0),      # one image at the configuration's shape: J = 17 (train.yaml:28), D = 64 (config.py:48), heat-map 64 x 64
]
NETWORK_CASES = [  # name, layers, image, J, D, batch
    ("r18", 18, 64, 3, 8, 2),
    ("r50", 50, 128, 2, 16, 4),
]
# Full-size configurations (BASELINE.json configs 1, 2 and 5): logits are stored as every LOGIT_STRIDE-th element of the flattened
# NCHW tensor, weight gradients as every max(1, size // 50000)-th element; stored in network_big.npz.
LOGIT_STRIDE = 997
BIG_HEAD_STD = 0.001     # head weights of the full-size cases: the reference's own N(0, 0.001) initialisation (pose3d_resnet.py:222-239)
NETWORK_BIG_CASES = [  # name, layers, image, J, D, batch
    ("cfg1_r18_128", 18, 128, 17, 64, 2),     # configs[0]: ResNet-18, 2-view 128x128, batch 2 (heat-map 32^2, D = 64 != W)
    ("cfg2_r50_256", 50, 256, 17, 64, 4),     # configs[1] shape (the bench configuration) at batch 4
    ("cfg5_r152_384", 152, 384, 17, 64, 2),   # configs[4]: ResNet-152, 384x384 (heat-map 96^2, D = 64 != W)
]

# 20 optimisation steps of the live reference (model + SmoothL1 criterion + torch.optim.Adam, fp32, CPU) on one fixed batch: the loss
# trajectory and the end state, stored in trajectory.npz.  name, layers, image, J, D, batch, steps, lr
TRAJECTORY_CASES = [
    ("cfg1_r18_128", 18, 128, 17, 64, 2, 20, 1e-3),      # BASELINE.json configs[0]: ResNet-18, 2-view 128x128, batch 2
    ("r50_128_b8", 50, 128, 17, 32, 8, 20, 1e-3),        # the bench network at a quarter of the pixels, 2 groups x 4 views
]
TRAJECTORY_HEAD_STD = 0.001


# Gradient tensors stored per network golden (round 4: >= 20 per case -- one group per ResNet stage, the stem, EVERY head tensor).  The first
# GRAD_KEYS_50K entries keep the round-1..3 sub-sampling (every max(1, size // 50000)-th element); the rest store every
# max(1, size // GRAD_CAP)-th element.  grad_stride(name, size) is the rule generator and tests share.
GRAD_KEYS_50K = ("final_layer.weight", "final_layer.bias", "deconv_layers.6.weight", "deconv_layers.7.weight", "deconv_layers.0.weight",
                 "layer3.0.conv2.weight", "layer1.0.conv1.weight", "conv1.weight")
GRAD_CAP = 8192


def golden_grad_keys(names):
    """The parameters whose gradients a network golden stores, out of the model's parameter names (ResNet-18 .. 152)."""
    names = list(names)
    have = set(names)
    picks = [k for k in GRAD_KEYS_50K if k in have]
    extra = ["bn1.weight", "bn1.bias"]
    for s in (1, 2, 3, 4):
        last = max(int(n.split(".")[1]) for n in names if n.startswith("layer%d." % s))
        extra += ["layer%d.0.conv1.weight" % s, "layer%d.0.conv2.weight" % s, "layer%d.0.conv3.weight" % s, "layer%d.0.bn2.weight" % s,
                  "layer%d.0.downsample.0.weight" % s, "layer%d.%d.conv2.weight" % (s, last), "layer%d.%d.bn1.bias" % (s, last)]
    extra += [n for n in names if n.startswith(("deconv_layers.", "final_layer."))]
    for k in extra:
        if k in have and k not in picks:
            picks.append(k)
    return picks


def grad_stride(name, size):
    return max(1, size // (50000 if name in GRAD_KEYS_50K else GRAD_CAP))
