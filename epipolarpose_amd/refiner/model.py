"""Residual refinement MLP -- mirror of the reference's ``refiner/model.py:15-143`` (``LinearPG``, ``LinearModelPG``, ``get_model``,
``weight_init``): same constructor arguments, same parameter / buffer names (checkpoints interchange), same two-headed output.

Every Linear runs on the hand-written MFMA GEMM (``epi_gemm_bf16`` forward / backward-data, ``epi_gemm_tn_bf16`` weight gradient,
``epi_column_sums_bf16`` bias gradient; bf16 operands, fp32 accumulation), every BatchNorm1d + ReLU on the fused NHWC BatchNorm
kernels (a [B, C] matrix is an NHWC tensor with H = W = 1), dropout on ``epi_dropout_bf16`` (stateless mask, regenerated in the
backward pass).  Feature counts that are not multiples of 8 (the 45 / 48 joint coordinates) are zero-padded to the GEMM's
granularity.  GPU only.
"""
import itertools

import torch
import torch.nn as nn

from .. import hip
from ..models.fused import FusedBatchNormAct

_seed_counter = itertools.count(1)


def weight_init(m):
    """refiner/model.py:8-12."""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight)


def _pad8(n):
    return (n + 7) // 8 * 8


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b on the MFMA GEMM; x [B, K] bf16, W [N, K] fp32 master (converted per call: the refiner is ~20 MB of weights)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, k = weight.shape
        kp, np_ = _pad8(k), _pad8(n)
        w16 = weight.detach().to(torch.bfloat16)
        if kp != k or np_ != n:
            wp = torch.zeros((np_, kp), dtype=torch.bfloat16, device=weight.device)
            wp[:n, :k] = w16
            w16 = wp
        if x.shape[1] != kp:
            xp = torch.zeros((x.shape[0], kp), dtype=torch.bfloat16, device=x.device)
            xp[:, :k] = x
            x = xp
        b32 = None
        if bias is not None:
            b32 = torch.zeros(np_, dtype=torch.float32, device=weight.device)
            b32[:n] = bias.detach()
        y = hip.gemm_bf16(x, w16, bias=b32)
        ctx.save_for_backward(x, w16)
        ctx.dims = (n, k, bias is not None)
        return y[:, :n] if np_ != n else y

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        n, k, has_bias = ctx.dims
        np_, kp = w16.shape
        dy = dy.to(torch.bfloat16)
        if np_ != n:
            dp = torch.zeros((dy.shape[0], np_), dtype=torch.bfloat16, device=dy.device)
            dp[:, :n] = dy
            dy = dp
        else:
            dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.gemm_bf16(dy, w16.t().contiguous())[:, :k]              # dx = dy @ W
        if ctx.needs_input_grad[1]:
            dw = hip.gemm_tn_bf16(dy, x)[:n, :k]                             # dW = dy^T @ x
        if has_bias and ctx.needs_input_grad[2]:
            db = hip.column_sum_bf16(dy)[:n]
        return dx, dw, db


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        ctx.p, ctx.seed = p, seed
        return hip.dropout_bf16(x, p, seed)

    @staticmethod
    def backward(ctx, dy):
        return hip.dropout_bf16(dy.to(torch.bfloat16).contiguous(), ctx.p, ctx.seed), None, None


class _BatchNorm1dAct(FusedBatchNormAct):
    """nn.BatchNorm1d (+ ReLU / LeakyReLU is not fused: ReLU only) with ``nn.BatchNorm1d``'s state_dict names."""

    def forward(self, x):
        b, c = x.shape
        y = super().forward(x.reshape(b, c, 1, 1).contiguous(memory_format=torch.channels_last))
        return y.reshape(b, c)


class _Block(nn.Module):
    def _act(self, y, bn):
        if self.bn:
            y = bn(y)                                     # BatchNorm + ReLU fused (leaky: BatchNorm only, activation below)
        if self.leaky or not self.bn:
            y = torch.nn.functional.leaky_relu(y) if self.leaky else torch.relu(y)
        if self.training and self.p_dropout > 0:
            y = _DropoutFn.apply(y.to(torch.bfloat16), self.p_dropout, next(_seed_counter) * 0x9E3779B1 + self.seed_base)
        return y

    def _linear(self, lin, x):
        return _LinearFn.apply(x.to(torch.bfloat16), lin.weight, lin.bias)


class LinearPG(_Block):
    """refiner/model.py:15-71: two residual pairs of Linear -> BatchNorm -> ReLU -> Dropout."""

    def __init__(self, linear_size, p_dropout=0.5, bias=True, bn=True, leaky=False):
        super().__init__()
        self.l_size, self.bn, self.leaky, self.p_dropout, self.seed_base = linear_size, bn, leaky, p_dropout, 0
        self.relu = nn.LeakyReLU(inplace=True) if leaky else nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(p_dropout)
        for i in range(1, 5):
            setattr(self, "w%d" % i, nn.Linear(linear_size, linear_size, bias=bias))
        if bn:
            for i in range(1, 5):
                setattr(self, "batch_norm%d" % i, _BatchNorm1dAct(linear_size, relu=not leaky))

    def forward(self, x):
        bn = (lambda i: getattr(self, "batch_norm%d" % i)) if self.bn else (lambda i: None)
        y = self._act(self._linear(self.w1, x), bn(1))
        y = self._act(self._linear(self.w2, y), bn(2))
        out = x + y
        y = self._act(self._linear(self.w3, out), bn(3))
        y = self._act(self._linear(self.w4, y), bn(4))
        return out + y


class LinearModelPG(_Block):
    """refiner/model.py:74-143: pre-processing layer, stage 0, first head p1, re-injection of p1, stage 1, second head p2."""

    def __init__(self, linear_size=1024, num_stage=2, p_dropout=0.5, input_size=15 * 3, output_size=15 * 3, bias=True, bn=True, leaky=False):
        super().__init__()
        self.linear_size, self.bn, self.leaky, self.p_dropout, self.num_stage = linear_size, bn, leaky, p_dropout, num_stage
        self.input_size, self.output_size, self.seed_base = input_size, output_size, 0
        self.linear_stages = nn.ModuleList([LinearPG(linear_size, p_dropout, bias=bias, bn=bn, leaky=leaky) for _ in range(num_stage)])
        self.w1 = nn.Linear(input_size, linear_size, bias=bias)
        self.w2 = nn.Linear(linear_size, output_size, bias=bias)
        self.w3 = nn.Linear(output_size, linear_size, bias=bias)
        self.w4 = nn.Linear(linear_size, output_size, bias=bias)
        self.relu = nn.LeakyReLU(inplace=True) if leaky else nn.ReLU(inplace=True)
        if bn:
            self.batch_norm1 = _BatchNorm1dAct(linear_size, relu=not leaky)
            self.batch_norm3 = _BatchNorm1dAct(linear_size, relu=not leaky)
        self.dropout = nn.Dropout(p_dropout)

    def seed(self, seed):
        """Base seed of the dropout masks (the reference draws them from torch's global generator)."""
        self.seed_base = int(seed)
        for s in self.linear_stages:
            s.seed_base = int(seed)

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("refiner: input must live on the GPU (no CPU fallback in epipolarpose_amd)")
        inp = self._act(self._linear(self.w1, x), self.batch_norm1 if self.bn else None)
        s1 = self.linear_stages[0](inp)
        p1 = self._linear(self.w2, s1)
        y = self._act(self._linear(self.w3, p1), self.batch_norm3 if self.bn else None)
        y = s1 + y + inp
        y = self.linear_stages[1](y)
        y = inp + y
        p2 = self._linear(self.w4, y)
        return p1.float(), p2.float()


def get_model(weights, **kwargs):
    """refiner/model.py:146-150."""
    model = LinearModelPG(**kwargs)
    if weights:
        model.load_state_dict(torch.load(weights, map_location="cpu")['state_dict'])
    return model
