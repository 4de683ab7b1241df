#!/usr/bin/env python
"""Representative convolution GEMMs for an SQ-counter pass (rocprofv3 --pmc ...): a few launches each, nothing else on the device."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402

DEV = torch.device("cuda:0")
hip.load()
CASES = [(256, 256, 3, 1, 16), (128, 128, 3, 1, 32), (64, 64, 3, 1, 64), (512, 512, 3, 1, 8), (1024, 256, 1, 1, 16), (256, 64, 1, 1, 64)]
for cin, cout, k, s, h in CASES:
    pad = k // 2
    x = torch.randn(32, cin, h, h, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=DEV) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = hip.conv2d_fwd(x, w, s, pad)
    dy = torch.randn_like(y.float()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wb = hip.conv2d_pack_weight_bwd(w, s, pad)
    for _ in range(3):
        hip.conv2d_fwd(x, w, s, pad)
        hip.conv2d_bwd_data(dy, wb, tuple(x.shape), k, s, pad)
        hip.conv2d_bwd_weight(x, dy, k, s, pad, dtype=torch.bfloat16)
torch.cuda.synchronize()
print("pmc conv target done")
