"""Case tables shared by make_golden.py (generator) and the tests (consumers)."""
DLOGITS_STRIDE = 13     # big cases store every 13th gradient element
INTEGRAL_CASES = [  # name, B, J, D, H, W, logit scale
    ("tiny", 2, 3, 8, 8, 8, 3.0),
    ("rect", 3, 4, 4, 8, 16, 2.0),
    ("mid", 2, 17, 16, 16, 16, 4.0),
    ("cube32", 1, 2, 32, 32, 32, 6.0),
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
