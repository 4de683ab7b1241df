"""Optimizer for the MI355X training path: one fused Adam launch + bf16 training copies of the convolution weights.

``FusedAdam`` is a ``torch.optim.Optimizer`` with ``torch.optim.Adam``'s arithmetic (no weight decay, no amsgrad:
what the reference builds in ``lib/utils/utils.py:55-59``) executed by ``epi_adam_step`` in a single kernel launch over
all parameters.  With ``low_precision_convs=True`` (default) every ``nn.Conv2d`` / deconvolution / 1x1 module of the model is switched to
a bf16 *training copy* of its weight: the fp32 ``weight`` Parameter stays the master (and the ``state_dict`` entry), the
forward uses a bf16 leaf tensor whose gradient the weight-gradient kernels write directly in bf16, and the Adam kernel reads that bf16
gradient and writes master + copy.  This removes autocast's per-step fp32->bf16 weight casts and bf16->fp32 gradient
casts (~110 launches, ~0.5 ms per step at batch 32) on top of Adam's own ~25 launches.
"""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

_ROW = 7        # int64 slots per table row: p, g, m, v, shadow, n, flags (low 32 bits: gradient is bf16)


def sync_training_copy(module):
    """Bring a module's bf16 training copy back in line with its fp32 master when the master was written by anyone but the
    Adam kernel (``load_state_dict``, a broadcast, an initialiser: all bump ``weight._version``; the kernel writes through raw
    pointers and refreshes the copy itself).  One integer compare per forward."""
    w = module.weight
    if w._version != module._lp_version:
        with torch.no_grad():
            module.weight_lp.copy_(w)
            wb = getattr(module, "weight_bwd", None)
            if wb is not None:                 # the packed backward-data operand follows the copy
                _, stride, pad = module.epi_geometry
                hip.conv2d_pack_weight_bwd(module.weight_lp, stride, pad, out=wb)
            wp = getattr(module, "weight_phase", None)
            if wp is not None:                 # deconvolution: the packed forward operand follows the copy
                hip._check(hip.load().epi_deconv4x4s2_pack_phase_cl(module.weight_lp.data_ptr(), w.shape[0], w.shape[1], wp.data_ptr(),
                                                                    hip._stream()), "epi_deconv4x4s2_pack_phase_cl")
        module._lp_version = w._version
    return module.weight_lp


def _lp_conv_forward(self, x):
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return F.conv2d(x, sync_training_copy(self), None, self.stride, self.padding, self.dilation, self.groups)


def _same_layout(a, b):
    """Same memory order: strides agree on every dimension of extent > 1 (a 1x1 convolution weight is the same memory in its
    contiguous and its channels_last form)."""
    return all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n > 1)


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model_or_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, low_precision_convs=True, max_grad_norm=None):
        """``max_grad_norm``: fold ``torch.nn.utils.clip_grad_norm_(params, max_grad_norm)`` into the step (reference refiner/main.py:53);
        ``last_grad_norm_sq`` then holds the squared total norm of the step (device scalar)."""
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm_sq = None
        lp_modules = []
        if isinstance(model_or_params, nn.Module):
            params = [p for p in model_or_params.parameters() if p.requires_grad]
            if low_precision_convs:
                lp_modules = [m for m in model_or_params.modules()
                              if (type(m) is nn.Conv2d or getattr(m, "supports_training_copy", False)) and m.bias is None
                              and m.weight.is_cuda and m.weight.requires_grad]
        else:
            params = [p for p in model_or_params if p.requires_grad]
        if not params or not all(p.is_cuda and p.dtype == torch.float32 and _dense(p) for p in params):
            raise ValueError("FusedAdam needs dense float32 CUDA parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        assert hip.load().epi_adam_tensor_bytes() == 8 * _ROW
        self._step = 0
        self._params = params
        dev = params[0].device
        offs, total = [], 0
        for p in params:                                   # 16-byte aligned slices of the flat state buffers
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        self._exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._shadow = torch.zeros(total, dtype=torch.bfloat16, device=dev) if lp_modules else None
        self._lp_of = {}
        self._parts, self._early_done, self._early_handle = None, False, None
        self._clear_list = None
        self._lp_modules = lp_modules
        self._offs = offs
        for m in lp_modules:
            w = m.weight
            idx = next(i for i, p in enumerate(params) if p is w)
            # the training copy has the master's memory layout (dense, possibly channels_last strides)
            lp = torch.as_strided(self._shadow, w.shape, w.stride(), storage_offset=offs[idx])
            lp.copy_(w.detach())
            lp.requires_grad_(True)
            object.__setattr__(m, "weight_lp", lp)
            object.__setattr__(m, "_lp_version", w._version)
            if type(m) is nn.Conv2d:
                m.forward = types.MethodType(_lp_conv_forward, m)
            self._lp_of[idx] = lp
        self._init_packed_weights(lp_modules, dev)
        # pinned staging copy of the device table (uploaded asynchronously whenever an address changed)
        self._table_host = torch.zeros(len(params) * _ROW, dtype=torch.int64).pin_memory()
        self._table_event = None
        self._table_dev = torch.zeros(len(params) * _ROW, dtype=torch.int64, device=dev)
        chunk = hip.load().epi_adam_chunk_elems()
        chunks = []
        t = self._table_host.numpy()
        for i, p in enumerate(params):
            n = p.numel()
            t[_ROW * i + 0] = p.data_ptr()
            t[_ROW * i + 2] = self._exp_avg.data_ptr() + 4 * offs[i]
            t[_ROW * i + 3] = self._exp_avg_sq.data_ptr() + 4 * offs[i]
            t[_ROW * i + 4] = (self._shadow.data_ptr() + 2 * offs[i]) if i in self._lp_of else 0
            t[_ROW * i + 5] = n
            chunks += [(i, c) for c in range((n + chunk - 1) // chunk)]
        self._chunks_all = chunks
        self._chunks_dev = torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous()
        self._copies = [self._lp_of.get(i) for i in range(len(params))]

    def _init_packed_weights(self, lp_modules, dev):
        """Backward-data operands (``weight_bwd``: transposed / parity-phase-ordered bf16 weights, include/epipolar_hip.h) of every
        convolution that runs on the implicit-GEMM kernels (modules carrying ``epi_geometry``, models/fused.py:FusedConvBn): one flat
        buffer, refreshed by ONE multi-layer pack launch after each Adam step."""
        convs = [m for m in lp_modules if getattr(m, "epi_geometry", None) is not None]
        deconvs = [m for m in lp_modules if getattr(m, "epi_deconv", False) and m.weight.is_contiguous(memory_format=torch.channels_last)]
        self._pack_rows, self._pack_tiles, self._pack_table = 0, 0, None
        self._pack_convs, self._pack_deconvs = convs, deconvs
        if not convs and not deconvs:
            return
        total = sum(m.weight.numel() for m in convs) + sum(m.weight.numel() for m in deconvs)
        self._packed = torch.empty(total, dtype=torch.bfloat16, device=dev)
        off = 0
        for m in convs:
            n = m.weight.numel()
            object.__setattr__(m, "weight_bwd", self._packed[off:off + n])
            off += n
        for m in deconvs:       # ConvTranspose2d(k4 s2 p1): the channels_last copy IS the backward-data operand; pack the forward one
            cin, cout = m.weight.shape[0], m.weight.shape[1]
            n = m.weight.numel()
            object.__setattr__(m, "weight_phase", self._packed[off:off + n].view(4, cout, 4 * cin))
            off += n
        self._pack_table, self._pack_rows, self._pack_tiles = self._build_pack_table(convs, deconvs, dev)
        self._pack_weights()

    def _build_pack_table(self, convs, deconvs, dev):
        """Device table of the multi-layer pack launch for the given modules: (table, rows, tiles)."""
        import ctypes
        lib = hip.load()
        if not convs and not deconvs:
            return None, 0, 0
        row_bytes = lib.epi_conv2d_pack_row_bytes()
        host = torch.zeros((len(convs) + len(deconvs)) * row_bytes, dtype=torch.uint8)
        tiles, r = 0, 0
        nt = ctypes.c_longlong(0)
        for m in convs:
            k, stride, pad = m.epi_geometry
            cout, cin = m.weight.shape[0], m.weight.shape[1]
            hip._check(lib.epi_conv2d_pack_fill_row(host.data_ptr() + r * row_bytes, m.weight_lp.data_ptr(), m.weight_bwd.data_ptr(), cout, cin, k, k,
                                                    stride, pad, tiles, ctypes.byref(nt)), "epi_conv2d_pack_fill_row")
            tiles += nt.value
            r += 1
        for m in deconvs:
            cin, cout = m.weight.shape[0], m.weight.shape[1]
            hip._check(lib.epi_deconv4x4s2_pack_fill_row(host.data_ptr() + r * row_bytes, m.weight_lp.data_ptr(), m.weight_phase.data_ptr(), cin, cout,
                                                         tiles, ctypes.byref(nt)), "epi_deconv4x4s2_pack_fill_row")
            tiles += nt.value
            r += 1
        return host.to(dev), r, tiles

    def _pack_weights(self, part=None):
        table, rows, tiles = (self._pack_table, self._pack_rows, self._pack_tiles) if part is None else part["pack"]
        if rows:
            hip._check(hip.load().epi_conv2d_pack_weight_bwd_multi(table.data_ptr(), rows, tiles, hip._stream()), "epi_conv2d_pack_weight_bwd_multi")

    # ---- the update of the deep part of the network INSIDE the backward pass ---------------------------------------------------------
    def enable_step_in_backward(self, boundary, late_modules):
        """Run the update of every parameter whose gradient is final when the backward pass crosses ``boundary``'s output (everything
        NOT in ``late_modules``: for PoseResNet ``boundary = model.layer1``, ``late_modules = [conv1, bn1, layer1]`` leaves 99 % of the
        parameters) from a tensor hook at that point, on the second HIP stream of the C++ glue: the slab sums of their weight gradients,
        their Adam update and the re-packing of their backward operands then run BESIDE the rest of the backward pass instead of after
        it (0.4 ms of HBM-bound work at the end of every ResNet-50 step).  ``step()`` finishes the late parameters.
        Measured gain on MI355X: below 1 % (see ``enable_step_in_backward`` at the end of this file), so nothing enables it by default.

        Opt-in, because it changes WHEN parameters change: only for loops that run exactly one backward pass per ``step()``, always
        call ``step()`` after it, use no gradient clipping (``max_grad_norm``) and no gradient all-reduce between backward and step
        (one process per GPU with ``BucketedGradSync``: leave it off).  The arithmetic of the update is unchanged."""
        if self.max_grad_norm is not None:
            raise ValueError("enable_step_in_backward: not with gradient clipping (the norm needs every gradient)")
        late_ids = {id(p) for m in late_modules for p in m.parameters()}
        late_mods = {id(m) for lm in late_modules for m in lm.modules()}
        early_idx = [i for i, p in enumerate(self._params) if id(p) not in late_ids]
        late_idx = [i for i, p in enumerate(self._params) if id(p) in late_ids]
        if not early_idx or not late_idx:
            raise ValueError("enable_step_in_backward: the split leaves one side empty")
        dev = self._table_dev.device
        self._parts = {}
        for name, idx in (("early", early_idx), ("late", late_idx)):
            keep = set(idx)
            chunks = [c for c in self._chunks_all if c[0] in keep]
            in_part = (lambda m: id(m) not in late_mods) if name == "early" else (lambda m: id(m) in late_mods)
            self._parts[name] = {"idx": idx, "nchunks": len(chunks), "chunks": torch.tensor(chunks, dtype=torch.int32, device=dev).contiguous(),
                                 "pack": self._build_pack_table([m for m in self._pack_convs if in_part(m)],
                                                                [m for m in self._pack_deconvs if in_part(m)], dev)}
        self._early_done = False
        if self._early_handle is not None:
            self._early_handle.remove()
        self._early_handle = boundary.register_forward_hook(self._boundary_forward_hook)

    def disable_step_in_backward(self):
        if self._early_handle is not None:
            self._early_handle.remove()
        self._early_handle, self._parts, self._early_done = None, None, False

    def _boundary_forward_hook(self, module, inputs, output):
        if self._parts is not None and torch.is_grad_enabled() and isinstance(output, torch.Tensor) and output.requires_grad:
            output.register_hook(self._early_step_hook)

    def _early_step_hook(self, grad):
        if self._parts is None:
            return None
        if self._early_done:
            raise RuntimeError("FusedAdam.enable_step_in_backward: a second backward pass before optimizer.step()")
        handle = hip.glue().begin_early_step(grad.device.index)       # sums the pending weight-gradient slabs on the second stream
        if handle:
            with torch.cuda.stream(torch.cuda.ExternalStream(handle, device=grad.device)), torch.no_grad():
                self._update(self._parts["early"], self._step + 1)
            self._early_done = True
        return None

    def _update(self, part, step_no):
        """Adam + backward-operand packing of one part of the parameters (None: all) on the current stream."""
        if self._table_event is not None:
            self._table_event.synchronize()              # the previous upload (a whole step ago) must have left the staging buffer
            self._table_event = None
        changed, keep = hip.glue().adam_prepare(self._params, self._copies, self._table_host, _ROW, [] if part is None else part["idx"])
        if changed:
            self._table_dev.copy_(self._table_host, non_blocking=True)
            self._table_event = torch.cuda.Event()
            self._table_event.record()
        group = self.param_groups[0]
        chunks, nchunks = (self._chunks_dev, len(self._chunks_all)) if part is None else (part["chunks"], part["nchunks"])
        if self.max_grad_norm is not None:
            self.last_grad_norm_sq = torch.zeros((), dtype=torch.float32, device=self._table_dev.device)
            hip.adam_step_clipped(self._table_dev, chunks, nchunks, group["lr"], group["betas"][0], group["betas"][1],
                                  group["eps"], step_no, float(self.max_grad_norm), self.last_grad_norm_sq)
        else:
            hip.adam_step(self._table_dev, chunks, nchunks, group["lr"], group["betas"][0], group["betas"][1], group["eps"], step_no)
        del keep
        self._pack_weights(part)                     # backward-data operands of the implicit-GEMM convolutions follow the update

    def refresh_training_copies(self):
        """Re-copy every fp32 master into its bf16 training copy (after the masters were loaded / broadcast)."""
        with torch.no_grad():
            for m in self._lp_modules:
                m.weight_lp.copy_(m.weight)
                m._lp_version = m.weight._version
        self._pack_weights()

    def _state_view(self, flat, i):
        """Slice of a flat moment buffer with parameter i's shape and strides."""
        p = self._params[i]
        return torch.as_strided(flat, p.shape, p.stride(), storage_offset=self._offs[i])

    def training_copies(self):
        """{parameter index: bf16 leaf tensor} of the convolution weights trained through a bf16 copy."""
        return dict(self._lp_of)

    def zero_grad(self, set_to_none=True):
        self._early_done = False             # (a pass whose step() never came -- an exception in the loop -- must not block the next one)
        if set_to_none:                      # one C++ call for the ~170 parameters and their bf16 training copies
            if self._clear_list is None:
                self._clear_list = list(self._params) + list(self._lp_of.values())
            hip.glue().clear_grads(self._clear_list)
            return
        super().zero_grad(set_to_none=False)
        for lp in self._lp_of.values():
            if lp.grad is not None:
                lp.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # gradient / parameter / shadow addresses -> the pinned host table, in C++ (no Python loop over ~170 parameters); the
        # addresses are stable once the allocator has warmed up, so the table upload is rare
        self._step += 1
        if self._parts is not None and self._early_done:      # the deep part was updated inside the backward pass
            self._early_done = False
            self._update(self._parts["late"], self._step)
        else:
            self._update(None, self._step)
        return loss

    def state_dict(self):
        """``torch.optim.Adam``'s format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so checkpoints move freely
        between this optimizer and the reference's ``torch.optim.Adam`` (scripts/train.py:118,177)."""
        sd = super().state_dict()
        if self._step > 0:
            step = torch.tensor(float(self._step))
            sd["state"] = {i: {"step": step.clone(), "exp_avg": self._state_view(self._exp_avg, i).clone(),
                               "exp_avg_sq": self._state_view(self._exp_avg_sq, i).clone()} for i in range(len(self._params))}
        return sd

    def load_state_dict(self, state_dict):
        state = state_dict.get("state", {})
        fused = state_dict.get("fused")                      # round-1 checkpoints: flat buffers
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        if fused is not None:
            self._step = int(fused["step"])
            self._exp_avg.copy_(fused["exp_avg"])
            self._exp_avg_sq.copy_(fused["exp_avg_sq"])
        elif state:
            if set(int(k) for k in state) != set(range(len(self._params))):
                raise ValueError("FusedAdam.load_state_dict: optimizer state covers %d of %d parameters" % (len(state), len(self._params)))
            steps = set()
            for k, st in state.items():
                i = int(k)
                if "max_exp_avg_sq" in st:
                    raise ValueError("FusedAdam.load_state_dict: amsgrad state is not supported")
                self._state_view(self._exp_avg, i).copy_(st["exp_avg"])
                self._state_view(self._exp_avg_sq, i).copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
            if len(steps) != 1:
                raise ValueError("FusedAdam.load_state_dict: per-parameter step counts differ: %s" % sorted(steps))
            self._step = steps.pop()
        else:
            self._step = 0
            self._exp_avg.zero_()
            self._exp_avg_sq.zero_()
        self.refresh_training_copies()               # masters may have been reloaded too


def enable_step_in_backward(optimizer, model, grad_sync=None):
    """Switch on ``FusedAdam.enable_step_in_backward`` when asked for (EPI_STEP_IN_BACKWARD=1) and where it applies: a FusedAdam
    without clipping, a model that names its split (``PoseResNet.step_in_backward_split``), no gradient all-reduce between backward
    and step.  Returns whether it is on.  OFF by default -- measured on MI355X (ResNet-50, batch 32, four boxes): the 0.4 ms of slab
    sums + Adam + weight packing leave the end of the step, but they are HBM-bound work beside an HBM-bound part of the backward pass
    (BatchNorm of the wide early layers), and the step gains 0.3 .. 0.9 % (7.24 / 7.27 vs 7.27 / 7.29 ms; boundary after layer 2, 3
    or 4 instead of 1: the same)."""
    import os
    if os.environ.get("EPI_STEP_IN_BACKWARD", "0") != "1" or grad_sync is not None:
        return False
    if not isinstance(optimizer, FusedAdam) or optimizer.max_grad_norm is not None or not hasattr(model, "step_in_backward_split"):
        return False
    boundary, late = model.step_in_backward_split()
    optimizer.enable_step_in_backward(boundary, late)
    return True
