#!/usr/bin/env python
"""GEMM-only target for SQ counter passes: a large square GEMM and the head shapes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip
DEV = torch.device("cuda:0")
hip.load()
for m, n, k in ((8192, 8192, 8192), (32768, 256, 1024), (131072, 1088, 256), (131072, 256, 1088)):
    a = torch.randn(m, k, device=DEV).to(torch.bfloat16)
    bt = torch.randn(n, k, device=DEV).to(torch.bfloat16)
    out = torch.empty(m, n, device=DEV, dtype=torch.bfloat16)
    for _ in range(3):
        hip.gemm_bf16(a, bt, out=out)
dl = torch.randn(131072, 1088, device=DEV).to(torch.bfloat16)
x = torch.randn(131072, 256, device=DEV).to(torch.bfloat16)
for _ in range(3):
    hip.gemm_tn_bf16(dl, x)
torch.cuda.synchronize()
