"""fp32-grade verification mode of the pose network -- on the SAME HIP kernels as the training path.

The training path (``models/pose3d_resnet.py``) multiplies bf16 by bf16 on the MFMA units and keeps bf16 activations in HBM, so
against the reference's fp32 network (``lib/models/pose3d_resnet.py:185-201``) it can only be held to a bf16 yardstick.  This module
runs the same network with fp32 activations through the same GEMM kernels: every GEMM operand is split into bf16 pieces
``x = hi + lo (+ lo2)`` (``epi_split_bf16``) that are laid out along the GEMM's reduction dimension -- channel blocks for the forward
and backward-data convolutions, batch blocks for the weight gradients -- so that one call of the unchanged bf16 kernel with fp32
accumulation and an fp32 result computes ``sum_{i+j<pieces} x_i * w_j``; BatchNorm (+residual +ReLU), max-pool and the bias sums run
on fp32 storage (the same kernel templates instantiated for ``float``), the criterion takes fp32 logits as it always could.
With two pieces the products carry ~2^-17 relative error, with three ~2^-24 (fp32 itself).

What it is for: ``tests/test_hip_precise.py`` holds logits, loss and every parameter gradient of this mode to an fp32-grade bar
against golden vectors of the live reference, and then uses it as the on-device fp32 yardstick for the bf16 training path.
It is NOT the product path (3-6x the GEMM work, Python autograd nodes) and nothing in the training loop imports it.

The 7x7 stem convolution (3 input channels) is left to the library in fp32, as the training path leaves it to the library in bf16.
"""
import ctypes

import torch
import torch.nn.functional as F

from .. import hip
from ..hip import EPI_F32, _check, _ptr, _stream, _workspace

STAGE_BLOCKS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottle", [3, 4, 6, 3]),
                101: ("bottle", [3, 4, 23, 3]), 152: ("bottle", [3, 8, 36, 3])}      # pose3d_resnet.py:288-292


def _patterns(pieces):
    """Piece indices of the left / right operand blocks: every product x_i * w_j with i + j < pieces."""
    pairs = [(i, j) for j in range(pieces) for i in range(pieces - j)]
    return [i for i, _ in pairs], [j for _, j in pairs]


def _split(x2d, pattern, row_concat):
    """x2d: fp32 [rows, C] contiguous view -> bf16 [rows, n*C] (channel blocks) or [n*rows, C] (row blocks)."""
    rows, c = x2d.shape
    n = len(pattern)
    out = torch.empty((n * rows, c) if row_concat else (rows, n * c), dtype=torch.bfloat16, device=x2d.device)
    arr = (ctypes.c_int * n)(*pattern)
    _check(hip.load().epi_split_bf16(_ptr(x2d), rows, c, arr, n, 1 if row_concat else 0, _ptr(out), _stream()), "epi_split_bf16")
    return out


def _cl(t):
    """fp32 channels_last tensor (the NHWC buffer the kernels address)."""
    t = t.float()
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def _rows(t):
    """[B, C, H, W] channels_last -> its [B*H*W, C] memory as a 2-D view."""
    b, c, h, w = t.shape
    return t.permute(0, 2, 3, 1).reshape(b * h * w, c)


class _Conv2d(torch.autograd.Function):
    """nn.Conv2d (no bias) of pose3d_resnet.py:21-88 on epi_conv2d_fwd_f32 / _bwd_data_f32 / epi_conv2d_bwd_weight with split operands."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, pieces):
        lib = hip.load()
        x, wc = _cl(x), _cl(w.detach())
        b, cin, h, wd = x.shape
        cout, _, k, _ = wc.shape
        pa, pb = _patterns(pieces)
        n = len(pa)
        xs = _split(_rows(x), pa, False)                                            # [B*H*W, n*Cin]
        ws = _split(wc.permute(0, 2, 3, 1).reshape(cout * k * k, cin), pb, False)   # [Cout*k*k, n*Cin] = [Cout][k][k][n*Cin]
        ho, wo = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
        y = torch.empty((b, cout, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        wsb = _workspace(lib.epi_conv2d_workspace_bytes(b, h, wd, n * cin, cout, k, k, stride, pad), x.device)
        _check(lib.epi_conv2d_fwd_f32(_ptr(xs), _ptr(ws), _ptr(y), b, h, wd, n * cin, cout, k, k, stride, pad, _ptr(wsb), wsb.numel(), _stream()),
               "epi_conv2d_fwd_f32")
        ctx.save_for_backward(x, wc)
        ctx.geo = (stride, pad, pieces)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        x, wc = ctx.saved_tensors
        stride, pad, pieces = ctx.geo
        dy = _cl(dy)
        b, cin, h, wd = x.shape
        cout, _, k, _ = wc.shape
        ho, wo = dy.shape[2], dy.shape[3]
        pa, pb = _patterns(pieces)
        n = len(pa)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dys = _split(_rows(dy), pa, False)                                                        # [.., n*Cout]
            wrow = _split(wc.permute(0, 2, 3, 1).reshape(cout * k * k, cin), pb, True)                # [n*Cout][k][k][Cin]
            wbwd = torch.empty(n * cout * k * k * cin, dtype=torch.bfloat16, device=x.device)
            _check(lib.epi_conv2d_pack_weight_bwd(_ptr(wrow), n * cout, cin, k, k, stride, pad, _ptr(wbwd), _stream()), "epi_conv2d_pack_weight_bwd")
            dx = torch.empty_like(x)
            wsb = _workspace(lib.epi_conv2d_workspace_bytes(b, h, wd, cin, n * cout, k, k, stride, pad), x.device)
            _check(lib.epi_conv2d_bwd_data_f32(_ptr(dys), _ptr(wbwd), _ptr(dx), b, h, wd, cin, n * cout, k, k, stride, pad, _ptr(wsb), wsb.numel(),
                                               _stream()), "epi_conv2d_bwd_data_f32")
        if ctx.needs_input_grad[1]:
            xr = _split(_rows(x), pa, True)                                                           # batch blocks: [n*B][H][W][Cin]
            dyr = _split(_rows(dy), pb, True)
            dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=x.device).contiguous(memory_format=torch.channels_last)
            wsb = _workspace(lib.epi_gemm_tn_workspace_bytes(n * b * ho * wo, cout, cin, k * k), x.device)
            _check(lib.epi_conv2d_bwd_weight(_ptr(xr), _ptr(dyr), _ptr(dw), EPI_F32, n * b, h, wd, cin, cout, k, k, stride, pad, _ptr(wsb),
                                             wsb.numel(), _stream()), "epi_conv2d_bwd_weight")
        return dx, dw, None, None, None


class _Deconv(torch.autograd.Function):
    """ConvTranspose2d(k4, s2, p1, no bias) of pose3d_resnet.py:158-183 on the epi_deconv4x4s2_* kernels with split operands."""

    @staticmethod
    def forward(ctx, x, w, pieces):
        lib = hip.load()
        x, wc = _cl(x), _cl(w.detach())                    # w [Cin, Cout, 4, 4]: channels_last memory [Cin][kh][kw][Cout]
        b, cin, h, wd = x.shape
        cout = wc.shape[1]
        pa, pb = _patterns(pieces)
        n = len(pa)
        xs = _split(_rows(x), pa, False)                                                       # [.., n*Cin]
        wrow = _split(wc.permute(0, 2, 3, 1).reshape(cin * 16, cout), pb, True)                # [n*Cin][16][Cout]
        wphase = torch.empty(4 * cout * 4 * n * cin, dtype=torch.bfloat16, device=x.device)
        _check(lib.epi_deconv4x4s2_pack_phase_cl(_ptr(wrow), n * cin, cout, _ptr(wphase), _stream()), "epi_deconv4x4s2_pack_phase_cl")
        y = torch.empty((b, cout, 2 * h, 2 * wd), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        wsb = _workspace(lib.epi_gemm_workspace_bytes(b * h * wd, cout, 4 * n * cin, 4), x.device)
        _check(lib.epi_deconv4x4s2_fwd_f32(_ptr(xs), _ptr(wphase), _ptr(y), b, h, wd, n * cin, cout, _ptr(wsb), wsb.numel(), _stream()),
               "epi_deconv4x4s2_fwd_f32")
        ctx.save_for_backward(x, wc)
        ctx.pieces = pieces
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        x, wc = ctx.saved_tensors
        dy = _cl(dy)
        b, cin, h, wd = x.shape
        cout = wc.shape[1]
        pa, pb = _patterns(ctx.pieces)
        n = len(pa)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dys = _split(_rows(dy), pa, False)                                                  # [.., n*Cout]
            wcol = _split(wc.permute(0, 2, 3, 1).reshape(cin * 16, cout), pb, False)            # [Cin][16][n*Cout]
            dx = torch.empty_like(x)
            wsb = _workspace(lib.epi_gemm_workspace_bytes(b * h * wd, cin, 16 * n * cout, 1), x.device)
            _check(lib.epi_deconv4x4s2_bwd_data_f32(_ptr(dys), _ptr(wcol), _ptr(dx), b, h, wd, cin, n * cout, _ptr(wsb), wsb.numel(), _stream()),
                   "epi_deconv4x4s2_bwd_data_f32")
        if ctx.needs_input_grad[1]:
            xr = _split(_rows(x), pa, True)
            dyr = _split(_rows(dy), pb, True)
            taps = torch.empty((cin, 16, cout), dtype=torch.float32, device=x.device)
            wsb = _workspace(lib.epi_gemm_tn_workspace_bytes(n * b * h * wd, cin, cout, 16), x.device)
            _check(lib.epi_deconv4x4s2_bwd_weight(_ptr(xr), _ptr(dyr), _ptr(taps), EPI_F32, n * b, h, wd, cin, cout, _ptr(wsb), wsb.numel(), _stream()),
                   "epi_deconv4x4s2_bwd_weight")
            dw = taps.view(cin, 4, 4, cout).permute(0, 3, 1, 2)
        return dx, dw, None


class _Conv1x1Bias(torch.autograd.Function):
    """The final 1x1 convolution with bias (pose3d_resnet.py:116-122,199) on epi_gemm_bf16 / epi_gemm_tn_bf16 (fp32 results)."""

    @staticmethod
    def forward(ctx, x, w, bias, pieces):
        lib = hip.load()
        x = _cl(x)
        b, cin, h, wd = x.shape
        cout = w.shape[0]
        w2 = w.detach().float().reshape(cout, cin).contiguous()
        pa, pb = _patterns(pieces)
        n = len(pa)
        m = b * h * wd
        xs, ws = _split(_rows(x), pa, False), _split(w2, pb, False)
        y = torch.empty((b, cout, h, wd), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        bz = bias.detach().float().contiguous() if bias is not None else None
        wsb = _workspace(lib.epi_gemm_workspace_bytes(m, cout, n * cin, 1), x.device)
        _check(lib.epi_gemm_bf16(_ptr(xs), n * cin, _ptr(ws), n * cin, _ptr(y), cout, EPI_F32, m, cout, n * cin, _ptr(bz), _ptr(wsb), wsb.numel(),
                                 _stream()), "epi_gemm_bf16")
        ctx.save_for_backward(x, w2)
        ctx.cfg = (pieces, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        x, w2 = ctx.saved_tensors
        pieces, has_bias = ctx.cfg
        dy = _cl(dy)
        b, cin, h, wd = x.shape
        cout = w2.shape[0]
        m = b * h * wd
        pa, pb = _patterns(pieces)
        n = len(pa)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dys, wt = _split(_rows(dy), pa, False), _split(w2.t().contiguous(), pb, False)        # [M, n*Cout], [Cin, n*Cout]
            dx = torch.empty_like(x)
            wsb = _workspace(lib.epi_gemm_workspace_bytes(m, cin, n * cout, 1), x.device)
            _check(lib.epi_gemm_bf16(_ptr(dys), n * cout, _ptr(wt), n * cout, _ptr(dx), cin, EPI_F32, m, cin, n * cout, None, _ptr(wsb), wsb.numel(),
                                     _stream()), "epi_gemm_bf16")
        if ctx.needs_input_grad[1]:
            dyr, xr = _split(_rows(dy), pa, True), _split(_rows(x), pb, True)                     # [n*M, Cout], [n*M, Cin]
            dw2 = torch.empty((cout, cin), dtype=torch.float32, device=x.device)
            wsb = _workspace(lib.epi_gemm_tn_workspace_bytes(n * m, cout, cin, 1), x.device)
            _check(lib.epi_gemm_tn_bf16(_ptr(dyr), cout, _ptr(xr), cin, _ptr(dw2), n * m, cout, cin, _ptr(wsb), wsb.numel(), _stream()), "epi_gemm_tn_bf16")
            dw = dw2.view(cout, cin, 1, 1)
        if has_bias and ctx.needs_input_grad[2]:
            sums = torch.zeros(2 * cout, dtype=torch.float32, device=x.device)
            _check(lib.epi_column_sums_f32(_ptr(_rows(dy)), m, cout, _ptr(sums), _stream()), "epi_column_sums_f32")
            db = sums[:cout]
        return dx, dw, db, None


class _BnAct(torch.autograd.Function):
    """BatchNorm2d (+ residual) (+ ReLU) on fp32 storage: epi_bn_act_fwd_f32 / _bwd_f32 (the training path's kernels for ``float``)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, num_batches, training, momentum, eps, relu):
        lib = hip.load()
        x = _cl(x)
        res = _cl(residual) if residual is not None else None
        b, c, h, w = x.shape
        r = b * h * w
        y = torch.empty_like(x)
        stats = torch.empty(4 * c, dtype=torch.float32, device=x.device)            # mean | rstd | scale | shift
        sums = torch.zeros(hip.bn_sum_copies(c) * 2 * c, dtype=torch.float32, device=x.device) if training else None
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _check(lib.epi_bn_act_fwd_f32(_ptr(x), _ptr(res), r, c, _ptr(g32), _ptr(b32), float(eps), float(momentum), 1 if training else 0,
                                      1 if relu else 0, _ptr(running_mean), _ptr(running_var), _ptr(num_batches) if training else None,
                                      _ptr(stats), _ptr(stats[c:]), _ptr(stats[2 * c:]), _ptr(sums), None, _ptr(y), _stream()), "epi_bn_act_fwd_f32")
        ctx.save_for_backward(x, y if (relu and res is not None) else None, stats, g32)
        ctx.cfg = (relu, res is not None, training)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        x, y, stats, g32 = ctx.saved_tensors
        relu, has_res, training = ctx.cfg
        if not training:
            raise RuntimeError("precise BatchNorm: backward through inference-mode statistics is not supported")
        dy = _cl(dy)
        b, c, h, w = x.shape
        r = b * h * w
        sums = torch.zeros(2 * c, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        _check(lib.epi_bn_act_bwd_f32(_ptr(dy), _ptr(x), _ptr(y), r, c, _ptr(g32), _ptr(stats), _ptr(stats[c:]), _ptr(stats[2 * c:]), 1 if relu else 0,
                                      _ptr(sums), _ptr(dx), _ptr(dres), None, None, _stream()), "epi_bn_act_bwd_f32")
        return dx, sums[c:], sums[:c], dres, None, None, None, None, None, None, None


class _MaxPool(torch.autograd.Function):
    """MaxPool2d(3, 2, 1) of the stem on fp32 storage (epi_maxpool3x3s2_*_f32)."""

    @staticmethod
    def forward(ctx, x):
        lib = hip.load()
        x = _cl(x)
        b, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = torch.empty((b, c, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
        pos = torch.empty((b, ho, wo, c), dtype=torch.uint8, device=x.device)
        _check(lib.epi_maxpool3x3s2_fwd_f32(_ptr(x), _ptr(y), _ptr(pos), b, h, w, c, _stream()), "epi_maxpool3x3s2_fwd_f32")
        ctx.save_for_backward(pos)
        ctx.shape = (b, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        pos, = ctx.saved_tensors
        b, c, h, w = ctx.shape
        dy = _cl(dy)
        dx = torch.empty((b, c, h, w), dtype=torch.float32, device=dy.device, memory_format=torch.channels_last)
        _check(lib.epi_maxpool3x3s2_bwd_f32(_ptr(dy), _ptr(pos), _ptr(dx), b, h, w, c, _stream()), "epi_maxpool3x3s2_bwd_f32")
        return dx


def conv2d(x, w, stride=1, pad=0, pieces=2):
    return _Conv2d.apply(x, w, stride, pad, pieces)


def deconv4x4s2(x, w, pieces=2):
    return _Deconv.apply(x, w, pieces)


def conv1x1_bias(x, w, bias, pieces=2):
    return _Conv1x1Bias.apply(x, w, bias, pieces)


def bn_act(x, sd, prefix, residual=None, relu=True, training=True, momentum=0.1, eps=1e-5):
    return _BnAct.apply(x, sd[prefix + ".weight"], sd[prefix + ".bias"], residual, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".num_batches_tracked"], training, momentum, eps, relu)


def maxpool3x3s2(x):
    return _MaxPool.apply(x)


def forward(sd, x, num_layers, num_deconv=3, training=True, pieces=2):
    """The reference network (pose3d_resnet.py:185-201, volume branch) on a reference-format state dict ``sd`` of fp32 GPU tensors
    (parameters may require grad; the BatchNorm buffers are updated in place in training mode, as nn.BatchNorm2d does).
    x: [B, 3, H, W] fp32.  Returns fp32 logits [B, J*D, H/4, W/4] (channels_last memory)."""
    if not x.is_cuda:
        raise RuntimeError("precise.forward: tensors must live on the GPU (no CPU fallback in epipolarpose_amd)")
    kind, blocks = STAGE_BLOCKS[num_layers]
    x = F.conv2d(x.float(), sd["conv1.weight"].float(), None, stride=2, padding=3)           # the library, fp32 (see the module docstring)
    x = bn_act(x, sd, "bn1", training=training)
    x = maxpool3x3s2(x)
    for li, nblk in enumerate(blocks, start=1):
        for bi in range(nblk):
            p = "layer%d.%d" % (li, bi)
            stride = 2 if (li > 1 and bi == 0) else 1
            res = x
            if (p + ".downsample.0.weight") in sd:
                res = conv2d(x, sd[p + ".downsample.0.weight"], stride, 0, pieces)
                res = bn_act(res, sd, p + ".downsample.1", relu=False, training=training)
            if kind == "basic":
                o = conv2d(x, sd[p + ".conv1.weight"], stride, 1, pieces)
                o = bn_act(o, sd, p + ".bn1", training=training)
                o = conv2d(o, sd[p + ".conv2.weight"], 1, 1, pieces)
                x = bn_act(o, sd, p + ".bn2", residual=res, training=training)
            else:
                o = conv2d(x, sd[p + ".conv1.weight"], 1, 0, pieces)
                o = bn_act(o, sd, p + ".bn1", training=training)
                o = conv2d(o, sd[p + ".conv2.weight"], stride, 1, pieces)
                o = bn_act(o, sd, p + ".bn2", training=training)
                o = conv2d(o, sd[p + ".conv3.weight"], 1, 0, pieces)
                x = bn_act(o, sd, p + ".bn3", residual=res, training=training)
    for di in range(num_deconv):
        if ("deconv_layers.%d.bias" % (3 * di)) in sd:
            raise NotImplementedError("precise.forward: DECONV_WITH_BIAS")
        x = deconv4x4s2(x, sd["deconv_layers.%d.weight" % (3 * di)], pieces)
        x = bn_act(x, sd, "deconv_layers.%d" % (3 * di + 1), training=training)
    w = sd["final_layer.weight"]
    if w.shape[-1] != 1:
        raise NotImplementedError("precise.forward: FINAL_CONV_KERNEL 3")
    return conv1x1_bias(x, w, sd.get("final_layer.bias"), pieces)
