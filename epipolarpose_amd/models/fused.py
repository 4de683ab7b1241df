"""Network building blocks backed by the HIP kernels of ``libepipolar_hip.so``.

* ``FusedBatchNormAct``  -- BatchNorm2d (+ residual add) (+ ReLU) in one apply pass (``epi_bn_act_fwd/bwd``); same
  parameters / buffers (and therefore ``state_dict`` keys) as ``nn.BatchNorm2d``.
* ``Deconv4x4s2``        -- ``nn.ConvTranspose2d(k=4, s=2, p=1)`` as MFMA implicit GEMMs (``epi_deconv4x4s2_*``).
* ``Conv1x1``            -- the final 1x1 convolution as an MFMA GEMM (``epi_gemm_bf16``).
All activations are NHWC (``channels_last``) bf16; parameters stay fp32 (master weights).  No CPU path.
"""
import torch
import torch.nn as nn

from .. import hip


def _nhwc_bf16(x):
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


class _BNActFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, module, relu):
        training = module.training
        if training:
            # accumulator hand-over between the two directions (no memsets in the steady fwd -> bwd -> fwd -> ... pattern):
            # the forward apply clears bwd_sums, the backward apply clears sums_ws.  A second training-mode forward
            # before the backward finds sums_ws still holding the previous call's sums and clears it here.
            if module._fwd_dirty:
                module.sums_ws.zero_()
            module._fwd_dirty, module._bwd_dirty = True, False
        y, stats = hip.bn_act_fwd(x, residual, module.pointers(), training, module.momentum, module.eps, relu)
        ctx.relu, ctx.has_res, ctx.training, ctx.module = relu, residual is not None, training, module
        if training:
            ctx.save_for_backward(x, y if (relu and residual is not None) else None, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("FusedBatchNormAct: backward through inference-mode statistics is not supported")
        x, y, stats = ctx.saved_tensors
        module = ctx.module
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        c = x.shape[1]
        # bwd_sums was cleared by this layer's forward pass; a second backward without a forward in between (two calls of
        # the layer inside one autograd graph) must not touch it again -- the first one's gradients may alias it
        sums = torch.zeros(2 * c, dtype=torch.float32, device=x.device) if module._bwd_dirty else module.bwd_sums
        ptrs = module.pointers()
        dx, dres = hip.bn_act_bwd(dy, x, y, ptrs[0], stats, ctx.relu, ctx.has_res, sums, ptrs[5])
        module._fwd_dirty, module._bwd_dirty = False, True
        return dx, sums[c:], sums[:c], dres, None, None


class FusedBatchNormAct(nn.Module):
    """y = act(BN(x) [+ residual]);  training-mode semantics of ``nn.BatchNorm2d(momentum)``: biased batch variance
    for normalisation, unbiased for the running estimate (pose3d_resnet.py:24,57,61,65,133,174 use momentum 0.1)."""

    def __init__(self, num_features, momentum=0.1, eps=1e-5, relu=True):
        super().__init__()
        self.num_features, self.momentum, self.eps, self.relu = num_features, momentum, eps, relu
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.register_buffer("sums_ws", torch.zeros(2 * num_features), persistent=False)   # kernel accumulator, kept zero
        self.register_buffer("bwd_sums", torch.zeros(2 * num_features), persistent=False)  # (dbeta | dgamma) accumulator
        self._ptrs = None
        self._fwd_dirty = False       # sums_ws holds a forward's sums that no backward has cleared yet
        self._bwd_dirty = False       # bwd_sums holds a backward's sums that no forward has cleared yet

    def pointers(self):
        """Device addresses of the persistent tensors, cached (hundreds of BatchNorm calls per step: the host-side cost
        of re-deriving them is measurable); re-derived whenever the weight tensor has moved (``.to()``, reload)."""
        p = self._ptrs
        if p is None or p[0] != self.weight.data_ptr() or p[2] != self.running_mean.data_ptr():
            p = self._ptrs = hip.bn_module_pointers(self.weight, self.bias, self.running_mean, self.running_var,
                                                    self.num_batches_tracked, self.sums_ws, self.bwd_sums)
        return p

    def forward(self, x, residual=None):
        if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            x = _nhwc_bf16(x)
        if residual is not None and (residual.dtype != torch.bfloat16 or not residual.is_contiguous(memory_format=torch.channels_last)):
            residual = _nhwc_bf16(residual)
        if not x.is_cuda:
            raise RuntimeError("FusedBatchNormAct: input must live on the GPU (no CPU fallback in epipolarpose_amd)")
        return _BNActFunction.apply(x, self.weight, self.bias, residual, self, self.relu)

    def extra_repr(self):
        return "{num_features}, eps={eps}, momentum={momentum}, relu={relu}".format(**self.__dict__)


class PointwiseConv(nn.Module):
    """Bias-free 1x1 stride-1 convolution of an NHWC tensor as a plain library GEMM (hipBLASLt through ``F.linear`` on the
    zero-copy [B*H*W, Cin] view); weight [Cout, Cin, 1, 1] keeps the ``nn.Conv2d`` name and shape (bottleneck conv1 / conv3,
    pose3d_resnet.py:56,61).  At batch 32 hipBLASLt runs these shapes at 0.7-1.0 PFLOP/s where MIOpen's implicit-GEMM
    convolution kernels reach ~0.3; forward and both backward GEMMs are stock autograd (no custom Function, no host cost)."""

    supports_training_copy = True

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1, 1))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)          # nn.Conv2d default initialisation
        self.register_parameter("bias", None)

    def forward(self, x):
        w = getattr(self, "weight_lp", None)
        if w is None:
            w = self.weight
        if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            x = _nhwc_bf16(x)
        if w.dtype != torch.bfloat16:
            w = w.to(torch.bfloat16)
        y = torch.nn.functional.linear(x.permute(0, 2, 3, 1), w.reshape(self.out_channels, self.in_channels))
        return y.permute(0, 3, 1, 2)                                # logical NCHW, NHWC memory


class _DeconvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        w_phase, w_bwd = hip.deconv_pack_weight(weight, want_phase=True, want_bwd=True)
        ctx.save_for_backward(x, weight, w_bwd)
        return hip.deconv4x4s2_fwd(x, w_phase)

    @staticmethod
    def backward(ctx, dy):
        x, weight, w_bwd = ctx.saved_tensors
        dy = _nhwc_bf16(dy)
        dx = hip.deconv4x4s2_bwd_data(dy, w_bwd) if ctx.needs_input_grad[0] else None
        dw = hip.deconv4x4s2_bwd_weight(x, dy).to(weight.dtype) if ctx.needs_input_grad[1] else None
        return dx, dw


class Deconv4x4s2(nn.Module):
    """ConvTranspose2d(kernel 4, stride 2, padding 1, no bias); weight [Cin, Cout, 4, 4] as in the reference."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 4, 4))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)          # nn.ConvTranspose2d default initialisation

    def forward(self, x):
        return _DeconvFunction.apply(_nhwc_bf16(x), self.weight)


class _Conv1x1Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        b, cin, h, w = x.shape
        cout = weight.shape[0]
        w2 = weight.detach().reshape(cout, cin)
        if w2.dtype != torch.bfloat16:
            w2 = w2.to(torch.bfloat16)
        x2 = x.permute(0, 2, 3, 1).reshape(b * h * w, cin)          # NHWC view, no copy
        out = hip.gemm_bf16(x2, w2, bias=None if bias is None else bias.detach().float())
        ctx.save_for_backward(x, w2)
        ctx.has_bias = bias is not None
        ctx.lp = weight.dtype == torch.bfloat16
        return out.reshape(b, h, w, cout).permute(0, 3, 1, 2)       # logical NCHW, NHWC memory

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        b, cin, h, w = x.shape
        cout = w2.shape[0]
        dy2 = _nhwc_bf16(dy).permute(0, 2, 3, 1).reshape(b * h * w, cout)
        x2 = x.permute(0, 2, 3, 1).reshape(b * h * w, cin)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = hip.gemm_bf16(dy2, w2.t().contiguous()).reshape(b, h, w, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = hip.gemm_tn_bf16(dy2, x2).reshape(cout, cin, 1, 1)
            if ctx.lp:
                dw = dw.to(torch.bfloat16)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = hip.column_sum_bf16(dy2)
        return dx, dw, db


class Conv1x1(nn.Module):
    """1x1 stride-1 convolution as an MFMA GEMM; weight [Cout, Cin, 1, 1] (+ bias [Cout]) named as in ``nn.Conv2d``
    (final layer: pose3d_resnet.py:116-122; bottleneck conv1/conv3: :56,61).  When an optimizer installs a bf16 training
    copy (``weight_lp``, see optim.FusedAdam) the forward reads it instead of casting the fp32 master every step."""

    supports_training_copy = True

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 1, 1))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)          # nn.Conv2d default initialisation
        if bias:
            self.bias = nn.Parameter(torch.empty(out_channels))
            bound = 1.0 / in_channels ** 0.5
            nn.init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter("bias", None)

    def forward(self, x):
        w = getattr(self, "weight_lp", None)
        return _Conv1x1Function.apply(_nhwc_bf16(x), self.weight if w is None else w, self.bias)
