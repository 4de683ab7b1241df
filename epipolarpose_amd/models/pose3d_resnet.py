"""Volumetric-heat-map pose network for MI355X -- mirror of the reference's ``lib/models/pose3d_resnet.py``.

API surface kept: ``get_pose_net(cfg, is_train)``, ``PoseResNet.forward(x[B,3,H,W]) -> [B, J*D, H/4, W/4]``
(``(heatmap, depth_fc)`` when ``MODEL.VOLUME`` is false), ``init_weights`` / ``load_pretrained_pose_model`` and -- the
compatibility surface proper -- the ``state_dict()`` key names and shapes of the reference (pose3d_resnet.py:93-126):
``conv1 bn1 layer{1..4}.{i}.{conv,bn}{1..3} layer*.0.downsample.{0,1} deconv_layers.{0,1,3,4,6,7} final_layer``.

MI355X-first choices: the network is built channels-last (NHWC) bf16 end to end and runs on hand-written HIP kernels only
(csrc/): every convolution -- the 7x7 stem through its space-to-depth form, the residual units' 1x1 / 3x3, the deconvolution head,
the final 1x1 -- is an MFMA implicit GEMM (``epi_conv2d_*``, ``epi_stem7x7s2_*``, ``epi_deconv4x4s2_*``, ``epi_gemm_bf16``); EVERY
BatchNorm (+ReLU, + residual add) is one fused apply pass; a whole residual unit, a deconvolution stage, the stem (+ its max-pool) and
the final layer are ONE C++ autograd node each (csrc/torch_glue.cpp); residual units are generated from a per-depth plan table instead
of two hand-written block classes.  The ``nn.Conv2d`` / ``FusedBatchNormAct`` modules only hold the parameters under the reference's
``state_dict`` names.  GPU only: there is no CPU path and no library (MIOpen / hipBLASLt) back-end; layer shapes the kernels do not
tile (channel counts that are not multiples of 64, deconvolution kernels other than 4, a final 3x3) fall to the stock ``nn`` modules,
which no shipped experiment uses.
"""
import logging
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import hip
from ..optim import sync_training_copy
from .fused import Conv1x1, Deconv4x4s2, deconv_bn_act, FusedBatchNormAct, FusedConvBn, FusedResidualUnit, conv_is_fusable

BN_MOMENTUM = 0.1
logger = logging.getLogger(__name__)

# depth -> (unit plan, units per stage).  A plan lists (kernel, width multiplier, carries the stride) per conv.
_BASIC = ((3, 1, True), (3, 1, False))                    # reference BasicBlock, pose3d_resnet.py:18-47
_BOTTLENECK = ((1, 1, False), (3, 1, True), (1, 4, False))  # reference Bottleneck (stride on the 3x3), :50-88
resnet_spec = {18: (_BASIC, [2, 2, 2, 2]),
               34: (_BASIC, [3, 4, 6, 3]),
               50: (_BOTTLENECK, [3, 4, 6, 3]),
               101: (_BOTTLENECK, [3, 4, 23, 3]),
               152: (_BOTTLENECK, [3, 8, 36, 3])}


class ResidualUnit(nn.Module):
    """conv-bn(-relu) chain + identity/projection shortcut; parameters are named conv{i}/bn{i}/downsample.{0,1}."""

    def __init__(self, inplanes, planes, plan, stride=1, unit_node=True):
        """unit_node: the whole unit as ONE autograd node (the training path); False = one node per conv/bn stage with autograd's own accumulation at
        the junction -- the same kernels, kept as the cross-check of tests/test_hip_conv.py."""
        super().__init__()
        self.n_conv = len(plan)
        self._unit_node = unit_node
        cin = inplanes
        for i, (k, mult, strided) in enumerate(plan, start=1):
            cout = planes * mult
            s_i = stride if strided else 1
            conv = nn.Conv2d(cin, cout, kernel_size=k, stride=s_i, padding=k // 2, bias=False)       # (parameter holder: FusedConvBn runs it)
            setattr(self, "conv%d" % i, conv)
            setattr(self, "bn%d" % i, FusedBatchNormAct(cout, momentum=BN_MOMENTUM, relu=True))
            cin = cout
        self.out_planes = cin
        self.downsample = None
        if stride != 1 or inplanes != cin:                 # pose3d_resnet.py:130-136
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, cin, kernel_size=1, stride=stride, bias=False),
                                            FusedBatchNormAct(cin, momentum=BN_MOMENTUM, relu=False))

        self._unit = None
        self._fused = self._build_fused()

    def _build_fused(self):
        """One FusedConvBn per conv/bn pair when every convolution of the unit qualifies (else the module-by-module path)."""
        pairs = [(getattr(self, "conv%d" % i), getattr(self, "bn%d" % i)) for i in range(1, self.n_conv + 1)]
        if self.downsample is not None:
            pairs.append((self.downsample[0], self.downsample[1]))
        if not all(conv_is_fusable(c) for c, _ in pairs):
            return ()
        fused = tuple(FusedConvBn(c, b) for c, b in pairs)
        self._unit = FusedResidualUnit(fused, self.downsample is not None) if self._unit_node else None
        return fused

    def forward(self, x):
        fused = self._fused
        if fused and x.is_cuda:
            if self._unit is not None:
                return self._unit(x)
            out = x
            for i in range(self.n_conv - 1):
                out = fused[i](out)                                                      # conv -> BN -> ReLU: one node
            shortcut = x if self.downsample is None else fused[self.n_conv](x)
            return fused[self.n_conv - 1](out, shortcut)                                 # conv -> BN -> (+ shortcut) -> ReLU
        out = x
        for i in range(1, self.n_conv):
            out = getattr(self, "bn%d" % i)(getattr(self, "conv%d" % i)(out))          # conv -> BN -> ReLU, fused
        shortcut = x if self.downsample is None else self.downsample(x)
        last = self.n_conv
        # conv -> BN -> (+ shortcut) -> ReLU in one apply pass (pose3d_resnet.py:44-45,85-86)
        return getattr(self, "bn%d" % last)(getattr(self, "conv%d" % last)(out), residual=shortcut)


def _deconv_geometry(kernel):
    """kernel -> (padding, output_padding) so that the output is exactly 2x the input (pose3d_resnet.py:145-156)."""
    table = {4: (1, 0), 3: (1, 1), 2: (0, 0)}
    if kernel not in table:
        raise ValueError("unsupported deconv kernel %r" % (kernel,))
    return table[kernel]


class PoseResNet(nn.Module):

    def __init__(self, plan, units, cfg, **kwargs):
        super().__init__()
        extra = cfg.MODEL.EXTRA
        self.deconv_with_bias = extra.DECONV_WITH_BIAS
        self.volume = cfg.MODEL.VOLUME
        self.num_joints = cfg.MODEL.NUM_JOINTS
        self.depth_res = cfg.MODEL.DEPTH_RES

        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = FusedBatchNormAct(64, momentum=BN_MOMENTUM, relu=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        width = 64
        for stage, (planes, n_unit) in enumerate(zip((64, 128, 256, 512), units), start=1):
            blocks = []
            for u in range(n_unit):
                blk = ResidualUnit(width, planes, plan, stride=2 if (u == 0 and stage > 1) else 1)
                width = blk.out_planes
                blocks.append(blk)
            setattr(self, "layer%d" % stage, nn.Sequential(*blocks))
        self.backbone_planes = width

        n_dec, filters, kernels = extra.NUM_DECONV_LAYERS, list(extra.NUM_DECONV_FILTERS), list(extra.NUM_DECONV_KERNELS)
        assert n_dec == len(filters), 'ERROR: num_deconv_layers is different len(num_deconv_filters)'
        assert n_dec == len(kernels), 'ERROR: num_deconv_layers is different len(num_deconv_filters)'
        head = []
        for planes, k in zip(filters, kernels):
            pad, out_pad = _deconv_geometry(k)
            if k == 4 and not self.deconv_with_bias and width % 64 == 0 and planes % 64 == 0:
                deconv = Deconv4x4s2(width, planes)                           # MFMA implicit GEMM
            else:   # configurations no shipped experiment uses (kernel 2/3, bias): MIOpen through torch
                deconv = nn.ConvTranspose2d(width, planes, kernel_size=k, stride=2, padding=pad, output_padding=out_pad,
                                            bias=self.deconv_with_bias)
            # third slot keeps the reference's Sequential indices (ReLU is fused into the BatchNorm pass)
            head += [deconv, FusedBatchNormAct(planes, momentum=BN_MOMENTUM, relu=True), nn.Identity()]
            width = planes
        self.deconv_layers = nn.Sequential(*head)
        # (deconvolution, BatchNorm) pairs that run as one C++ autograd node each; empty: module by module
        pairs = [(head[i], head[i + 1]) for i in range(0, len(head), 3)]
        self._head_pairs = tuple(pairs) if all(isinstance(d, Deconv4x4s2) for d, _ in pairs) else ()

        fk = extra.FINAL_CONV_KERNEL
        out_ch = self.num_joints * self.depth_res if self.volume else self.num_joints
        if fk == 1 and width % 64 == 0 and out_ch % 8 == 0:
            self.final_layer = Conv1x1(width, out_ch)                         # MFMA GEMM
        else:
            self.final_layer = nn.Conv2d(width, out_ch, kernel_size=fk, stride=1, padding=1 if fk == 3 else 0)

        if not self.volume:                                # legacy 2-D heat-map + depth FC branch, :124-126
            self.avgpool = nn.AvgPool2d(kernel_size=int(cfg.MODEL.IMAGE_SIZE[0] / 2 ** 5), stride=1)
            self.depth_fc = nn.Linear(self.backbone_planes, self.num_joints * self.depth_res)
        self.to(memory_format=torch.channels_last)

    def step_in_backward_split(self):
        """(boundary, late modules) for ``optim.FusedAdam.enable_step_in_backward``: when the backward pass crosses the output of
        ``layer1`` every gradient behind it (layers 2-4, the head: 99 % of the parameters) is final."""
        return self.layer1, [self.conv1, self.bn1, self.layer1]

    def stem(self, x, pool=False):
        """conv1 -> bn1 -> relu (pose3d_resnet.py:186-188); ``pool``: also the max-pool, inside the same node (the normalised tensor is then never
        written: csrc/torch_glue.cpp StemConvBnAct).  Returns (tensor, pooled)."""
        conv, bn = self.conv1, self.bn1
        if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0
                and conv.out_channels % 8 == 0):
            w = sync_training_copy(conv) if getattr(conv, "weight_lp", None) is not None else conv.weight
            g, b, rm, rv, nbt, sums_ws, bwd_sums = bn._tensors()
            pool = bool(pool and bn.relu)
            return hip.glue().stem_conv_bn_act(x, w, g, b, rm, rv, nbt, sums_ws, bwd_sums, bn._flags, bn.training, bn.momentum, bn.eps, bn.relu, pool), pool
        return bn(conv(x)), False

    def features(self, x):
        own_pool = x.is_cuda and self.conv1.out_channels % 8 == 0      # MaxPool2d(3, 2, 1) on epi_maxpool3x3s2_* (csrc/pool.hip)
        x, pooled = self.stem(x, pool=own_pool)
        if not pooled:
            x = hip.glue().maxpool3x3s2(x) if own_pool else self.maxpool(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    def forward(self, x):
        if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        feat = self.features(x)
        if self._head_pairs and feat.is_cuda:
            up = feat
            for deconv, bn in self._head_pairs:
                up = deconv_bn_act(deconv, bn, up)
        else:
            up = self.deconv_layers(feat)
        heat = self.final_layer(up)
        if self.volume:
            return heat
        depth = self.depth_fc(self.avgpool(feat).flatten(1))
        return heat, depth

    def init_weights(self, pretrained=''):
        """pose3d_resnet.py:214-255: N(0, 0.001) head, BN (1, 0), then load by file-name substring; a missing file
        is an error exactly as in the reference."""
        if not os.path.isfile(pretrained):
            logger.error('=> imagenet pretrained model dose not exist')
            logger.error('=> please download it first')
            raise ValueError('imagenet pretrained model does not exist')
        for m in self.deconv_layers.modules():
            if isinstance(m, (nn.ConvTranspose2d, Deconv4x4s2)):
                nn.init.normal_(m.weight, std=0.001)
                if self.deconv_with_bias:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, FusedBatchNormAct)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.final_layer.weight, std=0.001)
        nn.init.constant_(self.final_layer.bias, 0)
        if 'mpii' in pretrained or 'coco' in pretrained:
            logger.info('=> loading pretrained pose model {}'.format(pretrained))
            self.load_pretrained_pose_model(pretrained)
        elif 'imagenet' in pretrained:
            logger.info('=> loading pretrained imagenet model {}'.format(pretrained))
            self.load_state_dict(torch.load(pretrained, map_location="cpu"), strict=False)

    def load_pretrained_pose_model(self, pretrained):
        """pose3d_resnet.py:257-286: strip a DataParallel ``module.`` prefix, drop shape-mismatched keys."""
        loaded = torch.load(pretrained, map_location="cpu")
        if len(loaded) and all('module' in k for k in loaded):
            loaded = OrderedDict((k[7:], v) for k, v in loaded.items())
        own = self.state_dict()
        usable = OrderedDict()
        for k, v in loaded.items():
            if k not in own:
                logger.info('%s not in model_dict' % k)
                usable[k] = v
            elif own[k].shape != v.shape:
                logger.info('WARNING! There is a mismatch in => %s (%s, %s)' % (k, own[k].size(), v.size()))
            else:
                usable[k] = v
        self.load_state_dict(usable, strict=False)


def get_pose_net(cfg, is_train, **kwargs):
    """pose3d_resnet.py:295-305."""
    plan, units = resnet_spec[cfg.MODEL.EXTRA.NUM_LAYERS]
    model = PoseResNet(plan, units, cfg, **kwargs)
    if is_train and cfg.MODEL.INIT_WEIGHTS:
        model.init_weights(cfg.MODEL.PRETRAINED)
    return model
