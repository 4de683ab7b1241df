"""Network building blocks backed by the HIP kernels of ``libepipolar_hip.so``.

* ``FusedBatchNormAct``  -- BatchNorm2d (+ residual add) (+ ReLU) in one apply pass (``epi_bn_act_fwd/bwd``, reached through the
  C++ autograd glue ``csrc/torch_glue.cpp``); same parameters / buffers (and therefore ``state_dict`` keys) as ``nn.BatchNorm2d``.
* ``Deconv4x4s2``        -- ``nn.ConvTranspose2d(k=4, s=2, p=1)`` as MFMA implicit GEMMs (``epi_deconv4x4s2_*``).
* ``Conv1x1``            -- the final 1x1 convolution as an MFMA GEMM (``epi_gemm_bf16``).
All activations are NHWC (``channels_last``) bf16; parameters stay fp32 (master weights).  No CPU path.
"""
import torch
import torch.nn as nn

from .. import hip
from ..optim import sync_training_copy


def _nhwc_bf16(x):
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


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
        # kernel accumulator [copies][2C], kept zero
        self.register_buffer("sums_ws", torch.zeros(hip.bn_sum_copies(num_features) * 2 * num_features), persistent=False)
        self.register_buffer("bwd_sums", torch.zeros(2 * num_features), persistent=False)  # (dbeta | dgamma) accumulator
        # accumulator hand-over state shared with the C++ autograd glue (csrc/torch_glue.cpp): [0] sums_ws holds a forward's
        # sums that no backward has cleared yet, [1] bwd_sums holds a backward's sums that no forward has cleared yet
        self._flags = torch.zeros(2, dtype=torch.int32)
        self._args = None

    def _tensors(self):
        """(weight, bias, running_mean, running_var, num_batches_tracked, sums_ws, bwd_sums), cached: seven nn.Module
        attribute look-ups per call are measurable at 53 layers x 2 directions per step; refreshed when ``.to()`` /
        ``load_state_dict`` replaced a buffer or parameter object."""
        a, bufs, prm = self._args, self._buffers, self._parameters
        if a is None or a[0] is not prm["weight"] or a[2] is not bufs["running_mean"] or a[5] is not bufs["sums_ws"]:
            a = self._args = (prm["weight"], prm["bias"], bufs["running_mean"], bufs["running_var"], bufs["num_batches_tracked"],
                              bufs["sums_ws"], bufs["bwd_sums"])
        return a

    def forward(self, x, residual=None):
        if x.dtype != torch.bfloat16 or not x.is_contiguous(memory_format=torch.channels_last):
            x = _nhwc_bf16(x)
        if residual is not None and (residual.dtype != torch.bfloat16 or not residual.is_contiguous(memory_format=torch.channels_last)):
            residual = _nhwc_bf16(residual)
        w, b, rm, rv, nbt, sums_ws, bwd_sums = self._tensors()
        return hip.glue().bn_act(x, w, b, residual, rm, rv, nbt, sums_ws, bwd_sums, self._flags, self.training, self.momentum,
                                 self.eps, self.relu)

    def extra_repr(self):
        return "{num_features}, eps={eps}, momentum={momentum}, relu={relu}".format(**self.__dict__)


def conv_is_fusable(conv):
    """Can this convolution run on the hand-written implicit-GEMM kernels (epi_conv2d_*)?  Bias-free nn.Conv2d, groups 1,
    dilation 1, square kernel 1 or 3 with the matching padding, stride 1 or 2, channel counts the kernels tile (multiples of 64;
    see include/epipolar_hip.h).  The 7x7 stem on 3 input channels is not: it stays with the library."""
    if type(conv) is not nn.Conv2d or conv.bias is not None or conv.groups != 1 or conv.dilation != (1, 1):
        return False
    k, s, p = conv.kernel_size, conv.stride, conv.padding
    if k[0] != k[1] or k[0] not in (1, 3) or s[0] != s[1] or s[0] not in (1, 2) or p != (k[0] // 2, k[0] // 2):
        return False
    return conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0


class FusedConvBn:
    """conv -> BatchNorm (+ residual) (+ ReLU) of one residual-unit stage as ONE autograd node (``conv_bn_act`` of the C++
    glue): forward = implicit-GEMM convolution + BatchNorm statistics + apply, backward = BatchNorm reduce + apply +
    convolution backward-data + backward-weight, all enqueued from C++.  Holds no parameters: it reads them from the
    ``nn.Conv2d`` / ``FusedBatchNormAct`` modules it was built from (which keep the reference's ``state_dict`` names).

    Weight operands: the forward reads the bf16 training copy an optimizer installed (``conv.weight_lp``, optim.FusedAdam) or
    converts the fp32 master per call; the backward-data operand (transposed / phase-ordered weight) is ``conv.weight_bwd``
    when the optimizer maintains it (one multi-layer pack launch per step), else it is packed per call inside the node."""

    def __init__(self, conv, bn):
        self.conv, self.bn = conv, bn
        self.stride, self.pad = conv.stride[0], conv.padding[0]
        conv.epi_geometry = (conv.kernel_size[0], self.stride, self.pad)      # optimizers look for this to maintain weight_bwd

    def __call__(self, x, residual=None):
        conv, bn = self.conv, self.bn
        if getattr(conv, "weight_lp", None) is not None:
            w = sync_training_copy(conv)
            wb = getattr(conv, "weight_bwd", None)
        else:
            w, wb = conv.weight, None
        g, b, rm, rv, nbt, sums_ws, bwd_sums = bn._tensors()
        return hip.glue().conv_bn_act(x, w, wb, self.stride, self.pad, g, b, residual, rm, rv, nbt, sums_ws, bwd_sums, bn._flags,
                                      bn.training, bn.momentum, bn.eps, bn.relu)


_EMPTY = {}


def _empty_bf16(device):
    t = _EMPTY.get(device)
    if t is None:
        t = _EMPTY[device] = torch.empty(0, dtype=torch.bfloat16, device=device)
    return t


class FusedResidualUnit:
    """A whole BasicBlock / Bottleneck as ONE autograd node (``residual_unit`` of the C++ glue): every conv -> BatchNorm -> ReLU
    stage, the shortcut (identity or 1x1 projection + BatchNorm) and the final add + ReLU forward; in the backward pass the
    gradient arriving through the shortcut is added in the epilogue of the first stage's backward-data GEMM, so autograd's
    separate accumulation launch per unit disappears.  ``stages``: FusedConvBn objects, main path first, the projection last."""

    def __init__(self, stages, has_downsample):
        self.stages, self.has_downsample = tuple(stages), has_downsample
        self.geometry = [v for st in self.stages for v in (st.stride, st.pad)]

    def __call__(self, x):
        tensors = []
        for st in self.stages:
            conv, bn = st.conv, st.bn
            if getattr(conv, "weight_lp", None) is not None:
                w = sync_training_copy(conv)
                wb = getattr(conv, "weight_bwd", None)
            else:
                w, wb = conv.weight, None
            tensors.append(w)
            tensors.append(wb if wb is not None else _empty_bf16(w.device))
            tensors.extend(bn._tensors())
            tensors.append(bn._flags)
        bn0 = self.stages[0].bn
        return hip.glue().residual_unit(x, tensors, self.geometry, self.has_downsample, bn0.training, bn0.momentum, bn0.eps)


class _DeconvFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, w_phase):
        # weight: the bf16 training copy (optimizer-maintained w_phase) or the fp32 master (converted and packed here)
        w_phase, w16 = hip.deconv_weight_forms(weight, w_phase)
        ctx.save_for_backward(x, w16)
        ctx.grad_dtype = weight.dtype
        return hip.deconv4x4s2_fwd(x, w_phase)

    @staticmethod
    def backward(ctx, dy):
        x, w16 = ctx.saved_tensors
        dy = _nhwc_bf16(dy)
        cin, cout = w16.shape[0], w16.shape[1]
        # channels_last memory of the [Cin, Cout, 4, 4] weight = [Cin][16*Cout] = the backward-data operand as it stands
        dx = hip.deconv4x4s2_bwd_data(dy, torch.as_strided(w16, (cin, 16 * cout), (16 * cout, 1), w16.storage_offset())) if ctx.needs_input_grad[0] else None
        dw = hip.deconv4x4s2_bwd_weight(x, dy, dtype=ctx.grad_dtype) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class Deconv4x4s2(nn.Module):
    """ConvTranspose2d(kernel 4, stride 2, padding 1, no bias); weight [Cin, Cout, 4, 4] as in the reference.  In channels_last
    memory (what ``PoseResNet.to(memory_format=torch.channels_last)`` gives every 4-D parameter) the weight is [Cin][kh][kw][Cout]:
    the backward-data operand as it stands and the order the weight-gradient kernel writes; an optimizer that installs a bf16
    training copy (optim.FusedAdam) also maintains the packed forward operand ``weight_phase``."""

    supports_training_copy = True
    epi_deconv = True

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, 4, 4))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)          # nn.ConvTranspose2d default initialisation
        self.register_parameter("bias", None)

    def operands(self):
        """(weight the autograd node differentiates, packed forward operand or None)"""
        if getattr(self, "weight_lp", None) is not None:
            return sync_training_copy(self), getattr(self, "weight_phase", None)
        return self.weight, None

    def forward(self, x):
        w, wp = self.operands()
        return _DeconvFunction.apply(_nhwc_bf16(x), w, wp)


def deconv_bn_act(deconv, bn, x):
    """``Deconv4x4s2`` -> ``FusedBatchNormAct`` as ONE C++ autograd node (``deconv_bn_act`` of the glue): transposed convolution +
    BatchNorm statistics + apply forward; BatchNorm backward + backward-data + backward-weight (on the second stream) backward."""
    w, wp = deconv.operands()
    g, b, rm, rv, nbt, sums_ws, bwd_sums = bn._tensors()
    return hip.glue().deconv_bn_act(x, w, wp, g, b, rm, rv, nbt, sums_ws, bwd_sums, bn._flags, bn.training, bn.momentum, bn.eps, bn.relu)


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
        w = self.weight if getattr(self, "weight_lp", None) is None else sync_training_copy(self)
        return hip.glue().conv1x1_bias(x, w, self.bias)
