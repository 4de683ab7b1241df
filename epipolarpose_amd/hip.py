"""ctypes binding of libepipolar_hip.so (the C ABI declared in include/epipolar_hip.h).

This is the only way the package reaches the device kernels.  There is NO CPU fallback: if the shared
library is missing or a tensor is not on the GPU, the call raises.  PyTorch is used as plumbing only
(device memory, the current HIP stream).
"""
import ctypes
import os

import torch

# EPI_LIB_DIR: load libepipolar_hip.so (and the glue extension beside it) from another directory -- same-box A/B runs of two builds
# (tools/ab_bench.sh); unset: the in-tree build.
_LIB_DIR = os.environ.get("EPI_LIB_DIR") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib")
_LIB_PATH = os.path.join(_LIB_DIR, "libepipolar_hip.so")
_lib = None

EPI_F32, EPI_BF16, EPI_F64 = 0, 1, 2
EPI_NCHW, EPI_NHWC = 0, 1
LOSS_KINDS = {"l1": 0, "l2": 1, "smoothl1": 2}
TRI_METHODS = {"iterative": 0, "ls": 1, "dlt": 2, "poly": 3}

_vp, _i, _d, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t


class EpiViewMeta(ctypes.Structure):
    _fields_ = [(k, _vp) for k in ("center_x", "center_y", "width", "height", "scale", "rot", "R", "T", "f", "c", "P")]


_SIGNATURES = {
    "epi_version": (ctypes.c_char_p, []),
    "epi_status_string": (ctypes.c_char_p, [_i]),
    "epi_softargmax3d_workspace_bytes": (_sz, [_i] * 5),
    "epi_softargmax3d_fwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "epi_softargmax3d_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epi_softargmax3d_bwd_colsums": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int), _vp]),
    "epi_joint_loss": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    "epi_argmax_workspace_bytes": (_sz, [_i, _i]),
    "epi_argmax_rows": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_decode_to_image": (_i, [_vp, _i, _i, ctypes.POINTER(EpiViewMeta), _d, _d, _d, _vp, _vp]),
    "epi_triangulate_iterls": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _d, _i, _vp, _vp, _vp]),
    "epi_triangulate_ls": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "epi_triangulate_dlt": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "epi_triangulate_poly": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "epi_triangulate_staged": (_i, [_i]),
    "epi_fundamental_8point": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "epi_fundamental_lmeds_medians": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp]),
    "epi_fundamental_errors": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "epi_correct_matches": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "epi_reproject_labels": (_i, [_vp, _i, _i, _i, ctypes.POINTER(EpiViewMeta), _d, _d, _d, _i, _vp, _vp, _vp]),
    "epi_self_supervision": (_i, [_vp, _i, _i, _i, ctypes.POINTER(EpiViewMeta), _d, _d, _d, _i, _i, _d, _i, _vp, _vp, _vp, _vp]),
    "epi_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "epi_gemm_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "epi_deconv4x4s2_pack_weight": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "epi_deconv4x4s2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_deconv4x4s2_fwd_stats": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_deconv4x4s2_bwd_data": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_conv2d_workspace_bytes": (_sz, [_i] * 9),
    "epi_conv2d_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_conv1x1_fwd_bn_in": (_i, [_vp, _vp, _i, ctypes.c_float, ctypes.c_float, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_conv2d_pack_weight_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "epi_conv2d_pack_row_bytes": (_sz, []),
    "epi_conv2d_pack_fill_row": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]),
    "epi_conv2d_pack_weight_bwd_multi": (_i, [_vp, _i, ctypes.c_longlong, _vp]),
    "epi_conv2d_bwd_data": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "epi_conv2d_bwd_data_bnred": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_conv2d_bwd_data_half_addend_ok": (_i, [_i, _i, _i, _i, _i]),
    "epi_deconv4x4s2_bwd_data_bnred": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_gemm_bf16_bnred": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_bn_finalize": (_i, [_vp, ctypes.c_longlong, _i, ctypes.c_float, ctypes.c_float, _i, _vp]),
    "epi_maxpool3x3s2_bn_relu_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "epi_bn_act_fwd_dual": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _vp, _vp]),
    "epi_bn_act_bwd_reduced": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epi_conv2d_bwd_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "epi_conv2d_bwd_weight_deferred": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp, _vp]),
    "epi_slab_reduce_chunks": (ctypes.c_longlong, [ctypes.c_longlong]),
    "epi_slab_reduce_multi": (_i, [_vp, _i, ctypes.c_longlong, _vp]),
    "epi_wgrad_group_max": (_i, []),
    "epi_wgrad_group_plan": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_int)]),
    "epi_wgrad_group": (_i, [_vp, _i, _vp, _sz, _vp, _vp]),
    "epi_wgrad_item": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "epi_bn_sum_copies": (_i, [_i]),
    "epi_conv3x3_patch_mode": (_i, [_i]),
    "epi_gemm_tune": (_i, [_i, _i]),
    "epi_gemm_store_policy": (_i, [_i]),
    "epi_set_deterministic": (_i, [_i]),
    "epi_bn_act_fwd": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _i, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp, _vp, _vp, _vp]),
    "epi_bn_act_bwd": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epi_gemm_tn_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "epi_gemm_tn_plan": (_i, [_i, _i, _i, _i, ctypes.POINTER(ctypes.c_longlong)]),
    "epi_gemm_tn_bf16": (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "epi_deconv4x4s2_bwd_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_deconv4x4s2_pack_phase_cl": (_i, [_vp, _i, _i, _vp, _vp]),
    "epi_deconv4x4s2_pack_fill_row": (_i, [_vp, _vp, _vp, _i, _i, ctypes.c_longlong, ctypes.POINTER(ctypes.c_longlong)]),
    "epi_column_sums_bf16": (_i, [_vp, ctypes.c_longlong, _i, _vp, _vp]),
    "epi_stem7x7s2_s2d_bytes": (_sz, [_i, _i, _i]),
    "epi_stem7x7s2_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "epi_stem7x7s2_s2d": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "epi_stem7x7s2_pack_weight": (_i, [_vp, _i, _i, _vp, _vp]),
    "epi_stem7x7s2_unpack_weight_grad": (_i, [_vp, _i, _i, _vp, _i, _vp]),
    "epi_stem7x7s2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "epi_stem7x7s2_bwd_weight": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_split_bf16": (_i, [_vp, ctypes.c_longlong, _i, _vp, _i, _i, _vp, _vp]),
    "epi_conv2d_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_conv2d_bwd_data_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_deconv4x4s2_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_deconv4x4s2_bwd_data_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "epi_bn_act_fwd_f32": (_i, [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, ctypes.c_float, ctypes.c_float, _i, _i, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _vp, _vp, _vp]),
    "epi_bn_act_bwd_f32": (_i, [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "epi_column_sums_f32": (_i, [_vp, ctypes.c_longlong, _i, _vp, _vp]),
    "epi_maxpool3x3s2_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "epi_maxpool3x3s2_bwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "epi_adam_tensor_bytes": (_sz, []),
    "epi_adam_chunk_elems": (_i, []),
    "epi_adam_step": (_i, [_vp, _vp, _i, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_longlong, _vp]),
    "epi_adam_step_clipped": (_i, [_vp, _vp, _i, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_longlong,
                                   ctypes.c_float, _vp, _vp]),
    "epi_dropout_bf16": (_i, [_vp, _vp, ctypes.c_longlong, ctypes.c_float, ctypes.c_ulonglong, _vp]),
    "epi_maxpool3x3s2_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "epi_maxpool3x3s2_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "epi_crop_patches": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _i, _i, _i, _vp,
                              _i, _i, _vp]),
    "epi_crop_patches_occluded": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), _i, _i, _i, _vp,
                                       _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "epi_evaluate_poses": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path():
    return _LIB_PATH


class _Library:
    """The loaded library plus the names an older A/B build (EPI_LIB_DIR) does not export: using one of those says so by name."""

    def __init__(self, cdll, path, missing):
        self.__dict__.update(_cdll=cdll, _path=path, _missing=frozenset(missing))

    def __getattr__(self, name):
        if name in self._missing:
            raise RuntimeError("%s (loaded through EPI_LIB_DIR) does not export %s: that build predates the entry point -- "
                               "this code path cannot be part of an A/B run against it" % (self._path, name))
        return getattr(self._cdll, name)


def load():
    """Load the shared library (once).  Raises if it has not been built (python -m epipolarpose_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError("libepipolar_hip.so not found at %s -- build it with `python -m epipolarpose_amd.build` "
                               "(there is no CPU fallback)" % _LIB_PATH)
        lib = ctypes.CDLL(_LIB_PATH)
        missing = []
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                if os.environ.get("EPI_LIB_DIR"):      # an older build loaded for an A/B run: entry points added since are absent, and named on use
                    missing.append(name)
                    continue
                raise RuntimeError("%s does not export %s: stale build -- run `python -m epipolarpose_amd.build`" % (_LIB_PATH, name))
            fn.restype = res
            fn.argtypes = args
        _lib = _Library(lib, _LIB_PATH, missing)
    return _lib


class KernelTimer:
    """Optional per-entry-point timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  Used by bench.py for the live ``roofline`` figure; off by default."""

    def __init__(self):
        self.enabled = False
        self.records = {}

    def reset(self):
        self.records = {}

    def start(self, name):
        if not self.enabled:
            return None
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        self.records.setdefault(name, []).append(ev)
        return ev

    @staticmethod
    def stop(ev):
        if ev is not None:
            ev[1].record()

    def summary(self):
        """{name: (launches, mean milliseconds)} -- call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.records.items()}


timer = KernelTimer()


def _check(status, what):
    if status != 0:
        raise RuntimeError("%s failed: %s" % (what, load().epi_status_string(status).decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device_index=None):
    """Raw hipStream_t of torch's current stream.  The fast private accessor keeps the per-call Python overhead low
    (hundreds of launches per step); it follows stream switches (side streams, graph capture) like the public API."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream().cuda_stream


class _NoCtx:
    """Kernels run on the CURRENT device; callers keep tensors on it (one process per GPU).  `with _on(t.device)` used
    to switch devices per call, which cost ~15 us of host time per launch."""

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOCTX = _NoCtx()


def _on(device):
    return _NOCTX


def _dev(t, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (no CPU fallback in epipolarpose_amd)" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _ptr(t):
    """Device address as a plain int (ctypes converts it for c_void_p parameters; None -> NULL)."""
    return t.data_ptr() if t is not None else None


def _logits_format(logits):
    """(dtype enum, layout enum, dense tensor) for a [B, C, H, W] logits tensor in either memory format."""
    if logits.dim() != 4:
        raise ValueError("logits must be [B, J*D, H, W]")
    if logits.dtype == torch.float32:
        dt = EPI_F32
    elif logits.dtype == torch.bfloat16:
        dt = EPI_BF16
    else:
        raise TypeError("logits must be float32 or bfloat16, got %s" % logits.dtype)
    if logits.is_contiguous():
        return dt, EPI_NCHW, logits
    if logits.is_contiguous(memory_format=torch.channels_last):
        return dt, EPI_NHWC, logits
    return dt, EPI_NCHW, logits.contiguous()


_workspaces = {}


def _workspace(nbytes, device):
    """Per-(device, stream) scratch buffer, grown on demand (allocation is torch's caching allocator)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 16), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def softargmax3d_fwd(logits, num_joints):
    """-> (xyz [B,3J] f32, row_max [B*J] f32, row_sum [B*J] f32).  Reference: integral_loss.py:71-86."""
    lib = load()
    _dev(logits, name="logits")
    dt, layout, logits = _logits_format(logits)
    b, c, h, w = logits.shape
    if c % num_joints:
        raise ValueError("channels %d not divisible by num_joints %d" % (c, num_joints))
    d = c // num_joints
    xyz = torch.empty((b, 3 * num_joints), dtype=torch.float32, device=logits.device)
    rmax = torch.empty((b * num_joints,), dtype=torch.float32, device=logits.device)
    rsum = torch.empty_like(rmax)
    nbytes = lib.epi_softargmax3d_workspace_bytes(b, num_joints, d, h, w)
    ws = _workspace(nbytes, logits.device)
    with _on(logits.device):
        ev = timer.start("epi_softargmax3d_fwd")
        _check(lib.epi_softargmax3d_fwd(_ptr(logits), dt, layout, b, num_joints, d, h, w, _ptr(xyz), _ptr(rmax), _ptr(rsum),
                                        _ptr(ws), ws.numel(), _stream()), "epi_softargmax3d_fwd")
        timer.stop(ev)
    return xyz, rmax, rsum


def softargmax3d_bwd(logits, num_joints, row_max, row_sum, xyz, grad_xyz, grad_scale=None, col_sums=None):
    """-> dlogits (same dtype / memory format as logits).  col_sums: a ZEROED float32 [C] tensor that receives the per-channel sums of dlogits
    in the same pass (the bias gradient of the convolution that produced the logits); then -> (dlogits, delivered: bool)."""
    lib = load()
    _dev(logits, name="logits")
    dt, layout, logits = _logits_format(logits)
    b, c, h, w = logits.shape
    d = c // num_joints
    grad_xyz = _dev(grad_xyz, torch.float32, "grad_xyz").contiguous()
    dlogits = torch.empty_like(logits)      # preserves the memory format
    with _on(logits.device):
        ev = timer.start("epi_softargmax3d_bwd")
        if col_sums is not None:
            if col_sums.dtype != torch.float32 or col_sums.numel() != c or not col_sums.is_contiguous() or col_sums.device != logits.device:
                raise ValueError("col_sums must be a contiguous float32 [C] tensor on the logits' device")
            done = ctypes.c_int(0)
            _check(lib.epi_softargmax3d_bwd_colsums(_ptr(logits), dt, layout, b, num_joints, d, h, w, _ptr(row_max), _ptr(row_sum), _ptr(xyz),
                                                    _ptr(grad_xyz), _ptr(grad_scale), _ptr(dlogits), _ptr(col_sums), ctypes.byref(done), _stream()),
                   "epi_softargmax3d_bwd_colsums")
            timer.stop(ev)
            return dlogits, bool(done.value)
        _check(lib.epi_softargmax3d_bwd(_ptr(logits), dt, layout, b, num_joints, d, h, w, _ptr(row_max), _ptr(row_sum),
                                        _ptr(xyz), _ptr(grad_xyz), _ptr(grad_scale), _ptr(dlogits), _stream()),
               "epi_softargmax3d_bwd")
        timer.stop(ev)
    return dlogits


def joint_loss(pred, target, weight, kind, norm=False, size_average=True, need_grad=True):
    """-> (loss 0-dim f32, grad_pred [B,n] f32 or None).  Reference: integral_loss.py:7-47."""
    lib = load()
    pred = _dev(pred, torch.float32, "pred").contiguous()
    target = _dev(target, name="target").to(torch.float32).contiguous()
    weight = _dev(weight, name="weight").to(torch.float32).contiguous()
    if pred.shape != target.shape or pred.shape != weight.shape or pred.dim() != 2:
        raise ValueError("pred/target/weight must share a [B, n] shape")
    b, n = pred.shape
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if need_grad else None
    with _on(pred.device):
        _check(lib.epi_joint_loss(_ptr(pred), _ptr(target), _ptr(weight), b, n, LOSS_KINDS[kind], int(bool(norm)),
                                  int(bool(size_average)), _ptr(loss), _ptr(grad), _stream()), "epi_joint_loss")
    return loss, grad


def argmax_rows(x):
    """x [rows, n] f32/bf16 -> (idx int64 [rows], val f32 [rows]); first maximum (inference.py:25-26)."""
    lib = load()
    _dev(x, name="x")
    x = x.contiguous()
    dt = EPI_F32 if x.dtype == torch.float32 else EPI_BF16 if x.dtype == torch.bfloat16 else None
    if dt is None:
        raise TypeError("argmax_rows: float32 or bfloat16 only")
    rows, n = x.shape
    idx = torch.empty((rows,), dtype=torch.int64, device=x.device)
    val = torch.empty((rows,), dtype=torch.float32, device=x.device)
    nbytes = lib.epi_argmax_workspace_bytes(rows, n)
    ws = _workspace(nbytes, x.device)
    with _on(x.device):
        _check(lib.epi_argmax_rows(_ptr(x), dt, rows, n, _ptr(idx), _ptr(val), _ptr(ws), ws.numel(), _stream()),
               "epi_argmax_rows")
    return idx, val


class DeviceMeta:
    """float64 device copies of the per-sample camera / crop arrays + the C struct that points at them.

    Host inputs (what ``default_collate`` makes of the reference's ``meta`` dict, h36m.py:73-86: float64 tensors, or plain arrays)
    are packed into ONE staging buffer and cross PCIe in ONE copy -- the training loop builds a ``DeviceMeta`` per step, and eleven
    separate small H2D copies cost more host time than the whole self-supervision kernel."""

    KEYS = ("center_x", "center_y", "width", "height", "scale", "rot", "R", "T", "f", "c", "projection_matrix")
    FIELDS = (("center_x", "center_x"), ("center_y", "center_y"), ("width", "width"), ("height", "height"), ("scale", "scale"),
              ("rot", "rot"), ("R", "R"), ("T", "T"), ("f", "f"), ("c", "c"), ("P", "projection_matrix"))

    def __init__(self, meta, device):
        self.tensors = {}
        host = []
        for k in self.KEYS:
            if k not in meta:
                continue
            v = meta[k]
            t = v if isinstance(v, torch.Tensor) else torch.as_tensor(v)
            if t.is_cuda:
                self.tensors[k] = t.to(device=device, dtype=torch.float64).contiguous()
            else:
                host.append((k, t))
        if host:
            total = sum(t.numel() for _, t in host)
            stage = torch.empty(total, dtype=torch.float64)
            if torch.cuda.is_available():
                stage = stage.pin_memory()
            off = 0
            for _, t in host:
                stage[off:off + t.numel()].copy_(t.reshape(-1))          # converts dtype on the way
                off += t.numel()
            dev = stage.to(device=device, non_blocking=True)
            self._stage = stage                                           # keep the pinned buffer alive until the copy ran
            off = 0
            for k, t in host:
                self.tensors[k] = dev[off:off + t.numel()].view(t.shape)
                off += t.numel()
        self.batch = int(self.tensors["center_x"].shape[0])
        s = EpiViewMeta()
        for field, key in self.FIELDS:
            setattr(s, field, self.tensors[key].data_ptr() if key in self.tensors else None)
        self.struct = s


def decode_to_image(xyz, meta, patch_w=256.0, patch_h=256.0, rect3d=2000.0):
    """xyz [B,3J] f32 -> kps_img [B,J,3] f64 (u, v, z_mm).  Reference: img_utils.py:141-155,171-185."""
    lib = load()
    xyz = _dev(xyz, torch.float32, "xyz").contiguous()
    b, j = xyz.shape[0], xyz.shape[1] // 3
    out = torch.empty((b, j, 3), dtype=torch.float64, device=xyz.device)
    with _on(xyz.device):
        _check(lib.epi_decode_to_image(_ptr(xyz), b, j, ctypes.byref(meta.struct), patch_w, patch_h, rect3d, _ptr(out),
                                       _stream()), "epi_decode_to_image")
    return out


def triangulate(kps, proj, n_view, method="iterative", tolerance=3.0e-5, max_iter=10):
    """kps [B,J,>=2], proj [B,3,4] (f64 or f32, view-major batch) -> (X [G,J,3], status int32 [G,J])."""
    lib = load()
    _dev(kps, name="kps")
    if kps.dtype not in (torch.float64, torch.float32) or proj.dtype != kps.dtype:
        raise TypeError("kps and proj must both be float64 or both float32")
    kps, proj = kps.contiguous(), _dev(proj, name="proj").contiguous()
    b, j, stride = kps.shape
    if b % n_view:
        raise ValueError("batch %d is not a multiple of n_view %d" % (b, n_view))
    g = b // n_view
    dt = EPI_F64 if kps.dtype == torch.float64 else EPI_F32
    x = torch.empty((g, j, 3), dtype=kps.dtype, device=kps.device)
    status = torch.empty((g, j), dtype=torch.int32, device=kps.device) if method != "ls" else None
    with _on(kps.device):
        if method == "iterative":
            st = lib.epi_triangulate_iterls(_ptr(kps), stride, _ptr(proj), dt, g, n_view, j, tolerance, max_iter, _ptr(x),
                                            _ptr(status), _stream())
        elif method == "ls":
            # the linear solve has no failure mode: its status is all ones (triangulation.py:97 returns np.ones(len(u1), dtype=bool)) -- a broadcast constant
            # instead of 68 bytes per group written by the kernel beside the 940 it has to move (the C entry point still fills a status array it is given)
            status = torch.ones((), dtype=torch.int32, device=kps.device).expand(g, j)
            st = lib.epi_triangulate_ls(_ptr(kps), stride, _ptr(proj), dt, g, n_view, j, _ptr(x), None, _stream())
        elif method == "dlt":
            st = lib.epi_triangulate_dlt(_ptr(kps), stride, _ptr(proj), dt, g, n_view, j, _ptr(x), _ptr(status), _stream())
        elif method == "poly":
            st = lib.epi_triangulate_poly(_ptr(kps), stride, _ptr(proj), dt, g, n_view, j, _ptr(x), _ptr(status), _stream())
        else:
            raise ValueError(method)
    _check(st, "epi_triangulate_" + method)
    return x, status


def fundamental_8point(u1, u2):
    """cv2.findFundamentalMat(..., FM_8POINT), batched.  u1, u2 [G, J, 2] float64 -> (F [G, 3, 3], status int32 [G])."""
    lib = load()
    u1, u2 = _dev(u1, torch.float64, "u1").contiguous(), _dev(u2, torch.float64, "u2").contiguous()
    g, j, _ = u1.shape
    if u2.shape != u1.shape or u1.shape[2] != 2:
        raise ValueError("u1, u2 must both be [G, J, 2]")
    f = torch.empty((g, 3, 3), dtype=torch.float64, device=u1.device)
    status = torch.empty((g,), dtype=torch.int32, device=u1.device)
    with _on(u1.device):
        _check(lib.epi_fundamental_8point(_ptr(u1), _ptr(u2), g, j, _ptr(f), _ptr(status), _stream()), "epi_fundamental_8point")
    return f, status


def fundamental_lmeds_medians(f, u1, u2):
    """Median symmetric epipolar error of every candidate: f [H, 3, 3] float64, u1 / u2 [N, 2] float64 -> medians [H] float64."""
    lib = load()
    f = _dev(f, torch.float64, "f").contiguous()
    u1, u2 = _dev(u1, torch.float64, "u1").contiguous(), _dev(u2, torch.float64, "u2").contiguous()
    if f.dim() != 3 or f.shape[1:] != (3, 3) or u1.dim() != 2 or u1.shape[1] != 2 or u2.shape != u1.shape:
        raise ValueError("f must be [H, 3, 3] and u1, u2 [N, 2]")
    med = torch.empty((f.shape[0],), dtype=torch.float64, device=f.device)
    with _on(f.device):
        _check(lib.epi_fundamental_lmeds_medians(_ptr(f), f.shape[0], _ptr(u1), _ptr(u2), u1.shape[0], _ptr(med), _stream()),
               "epi_fundamental_lmeds_medians")
    return med


def fundamental_errors(f, u1, u2):
    """float32 symmetric epipolar errors [N] of ONE matrix f [3, 3] float64 over the pairs u1 / u2 [N, 2] float64."""
    lib = load()
    f = _dev(f, torch.float64, "f").contiguous()
    u1, u2 = _dev(u1, torch.float64, "u1").contiguous(), _dev(u2, torch.float64, "u2").contiguous()
    if f.shape != (3, 3) or u1.dim() != 2 or u1.shape[1] != 2 or u2.shape != u1.shape:
        raise ValueError("f must be [3, 3] and u1, u2 [N, 2]")
    err = torch.empty((u1.shape[0],), dtype=torch.float32, device=f.device)
    with _on(f.device):
        _check(lib.epi_fundamental_errors(_ptr(f), _ptr(u1), _ptr(u2), u1.shape[0], _ptr(err), _stream()), "epi_fundamental_errors")
    return err


def correct_matches(f, u1, u2):
    """cv2.correctMatches, batched.  f [G,3,3], u1/u2 [G,J,2] float64 -> (u1', u2') [G,J,2]."""
    lib = load()
    f = _dev(f, torch.float64, "f").contiguous()
    u1, u2 = _dev(u1, torch.float64, "u1").contiguous(), _dev(u2, torch.float64, "u2").contiguous()
    g, j, _ = u1.shape
    if f.shape != (g, 3, 3) or u2.shape != u1.shape or u1.shape[2] != 2:
        raise ValueError("f must be [G,3,3] and u1, u2 [G,J,2]")
    o1, o2 = torch.empty_like(u1), torch.empty_like(u2)
    with _on(u1.device):
        _check(lib.epi_correct_matches(_ptr(f), _ptr(u1), _ptr(u2), g, j, _ptr(o1), _ptr(o2), _stream()), "epi_correct_matches")
    return o1, o2


def reproject_labels(x_world, meta, n_view, patch_w=256.0, patch_h=256.0, rect3d=2000.0, root_joint=0):
    """X [G,J,3] f64 -> (label [B,3J] f32, weight [B,3J] f32).  Reference: img_utils.py:212-243."""
    lib = load()
    x_world = _dev(x_world, torch.float64, "x_world").contiguous()
    g, j, _ = x_world.shape
    b = g * n_view
    label = torch.empty((b, 3 * j), dtype=torch.float32, device=x_world.device)
    weight = torch.empty_like(label)
    with _on(x_world.device):
        _check(lib.epi_reproject_labels(_ptr(x_world), g, n_view, j, ctypes.byref(meta.struct), patch_w, patch_h, rect3d,
                                        root_joint, _ptr(label), _ptr(weight), _stream()), "epi_reproject_labels")
    return label, weight


def self_supervision(xyz, meta, n_view, method="iterative", patch_w=256.0, patch_h=256.0, rect3d=2000.0, root_joint=0,
                     tolerance=3.0e-5, max_iter=10, want_world=False):
    """Fused decode -> triangulate -> re-project.  -> (label, weight[, X]).  Reference: img_utils.py:166-190."""
    lib = load()
    xyz = _dev(xyz, torch.float32, "xyz").contiguous()
    b, j = xyz.shape[0], xyz.shape[1] // 3
    if b % n_view:
        raise ValueError("batch %d is not a multiple of n_view %d" % (b, n_view))
    g = b // n_view
    label = torch.empty((b, 3 * j), dtype=torch.float32, device=xyz.device)
    weight = torch.empty_like(label)
    xw = torch.empty((g, j, 3), dtype=torch.float64, device=xyz.device) if want_world else None
    with _on(xyz.device):
        ev = timer.start("epi_self_supervision")
        _check(lib.epi_self_supervision(_ptr(xyz), g, n_view, j, ctypes.byref(meta.struct), patch_w, patch_h, rect3d,
                                        root_joint, TRI_METHODS[method], tolerance, max_iter, _ptr(label), _ptr(weight),
                                        _ptr(xw), _stream()), "epi_self_supervision")
        timer.stop(ev)
    return (label, weight, xw) if want_world else (label, weight)


# ---------------------------------------------------------------------------------------------------------------
# Deconvolution head (MFMA implicit GEMMs).  Activations: NHWC bf16, passed as torch tensors of logical shape
# [B, C, H, W] in channels_last memory format (so .data_ptr() is the NHWC buffer).
# ---------------------------------------------------------------------------------------------------------------
def _nhwc_bf16(t, name):
    _dev(t, torch.bfloat16, name)
    if t.dim() != 4:
        raise ValueError("%s must be [B, C, H, W]" % name)
    return t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous(memory_format=torch.channels_last)


def gemm_bf16(a, bt, bias=None, out_dtype=torch.bfloat16, out=None):
    """C[M,N] = A[M,K] @ Bt[N,K]^T (+ bias).  a, bt: 2-D bf16 CUDA tensors with unit inner stride."""
    lib = load()
    _dev(a, torch.bfloat16, "a")
    _dev(bt, torch.bfloat16, "bt")
    if a.stride(1) != 1:
        a = a.contiguous()
    if bt.stride(1) != 1:
        bt = bt.contiguous()
    m, k = a.shape
    n = bt.shape[0]
    if bt.shape[1] != k:
        raise ValueError("inner dimensions differ")
    if out is None:
        out = torch.empty((m, n), dtype=out_dtype, device=a.device)
    cdt = EPI_BF16 if out.dtype == torch.bfloat16 else EPI_F32
    if bias is not None:
        bias = _dev(bias, torch.float32, "bias").contiguous()
    ws = _workspace(lib.epi_gemm_workspace_bytes(m, n, k, 1), a.device)
    with _on(a.device):
        ev = timer.start("epi_gemm_bf16")
        _check(lib.epi_gemm_bf16(_ptr(a), a.stride(0), _ptr(bt), bt.stride(0), _ptr(out), out.stride(0), cdt, m, n, k,
                                 _ptr(bias), _ptr(ws), ws.numel(), _stream()), "epi_gemm_bf16")
        timer.stop(ev)
    return out


def deconv_pack_weight(weight, want_phase=True, want_bwd=True):
    """weight [Cin, Cout, 4, 4] (any float dtype) -> (w_phase [4, Cout, 4*Cin] bf16, w_bwd [Cin, 16*Cout] bf16)."""
    lib = load()
    _dev(weight, name="weight")
    cin, cout, kh, kw = weight.shape
    if (kh, kw) != (4, 4):
        raise ValueError("deconv head kernels cover kernel 4 / stride 2 / padding 1 only")
    w = weight.detach().to(torch.bfloat16).contiguous()
    wp = torch.empty((4, cout, 4 * cin), dtype=torch.bfloat16, device=w.device) if want_phase else None
    wb = torch.empty((cin, 16 * cout), dtype=torch.bfloat16, device=w.device) if want_bwd else None
    with _on(w.device):
        _check(lib.epi_deconv4x4s2_pack_weight(_ptr(w), cin, cout, _ptr(wp), _ptr(wb), _stream()), "epi_deconv4x4s2_pack_weight")
    return wp, wb


def deconv_weight_forms(weight, w_phase=None):
    """weight [Cin, Cout, 4, 4] -> (w_phase [4, Cout, 4*Cin] bf16, w_bwd [Cin, 16*Cout] bf16) WITHOUT re-laying the weight out: in
    channels_last memory ([Cin][kh][kw][Cout], what ``model.to(memory_format=torch.channels_last)`` gives it) the bf16 weight is the
    backward-data operand as it stands; only the forward operand is packed (skipped when the optimizer maintains ``w_phase``)."""
    lib = load()
    _dev(weight, name="weight")
    cin, cout, kh, kw = weight.shape
    if (kh, kw) != (4, 4):
        raise ValueError("deconv head kernels cover kernel 4 / stride 2 / padding 1 only")
    w = weight.detach()
    if w.dtype != torch.bfloat16:
        w = w.to(torch.bfloat16)
    if not w.is_contiguous(memory_format=torch.channels_last):
        w = w.contiguous(memory_format=torch.channels_last)
    if w_phase is None:
        w_phase = torch.empty((4, cout, 4 * cin), dtype=torch.bfloat16, device=w.device)
        _check(lib.epi_deconv4x4s2_pack_phase_cl(_ptr(w), cin, cout, _ptr(w_phase), _stream()), "epi_deconv4x4s2_pack_phase_cl")
    return w_phase, w            # w: logical [Cin, Cout, 4, 4], memory [Cin][16*Cout]


def deconv4x4s2_fwd(x, w_phase, bn_sums=None):
    """x [B, Cin, H, W] channels_last bf16 -> y [B, Cout, 2H, 2W] channels_last bf16 (raw ConvTranspose2d output).  ``bn_sums`` (zeroed f32
    [bn_sum_copies(Cout) * 2*Cout]): asks for the per-channel (sum, sum of squares) of the result from the GEMM epilogue; returns (y, done) then."""
    lib = load()
    x = _nhwc_bf16(x, "x")
    b, cin, h, w = x.shape
    cout = w_phase.shape[1]
    y = torch.empty((b, cout, 2 * h, 2 * w), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ws = _workspace(lib.epi_gemm_workspace_bytes(b * h * w, cout, 4 * cin, 4), x.device)
    with _on(x.device):
        ev = timer.start("epi_deconv4x4s2_fwd")
        done = ctypes.c_int(0)
        _check(lib.epi_deconv4x4s2_fwd_stats(_ptr(x), _ptr(w_phase), _ptr(y), b, h, w, cin, cout, _ptr(bn_sums),
                                             ctypes.addressof(done) if bn_sums is not None else None, _ptr(ws), ws.numel(), _stream()),
               "epi_deconv4x4s2_fwd")
        timer.stop(ev)
    return y if bn_sums is None else (y, bool(done.value))


def deconv4x4s2_bwd_data(dy, w_bwd):
    """dy [B, Cout, 2H, 2W] channels_last bf16 -> dx [B, Cin, H, W] channels_last bf16."""
    lib = load()
    dy = _nhwc_bf16(dy, "dy")
    b, cout, h2, w2 = dy.shape
    cin = w_bwd.shape[0]
    dx = torch.empty((b, cin, h2 // 2, w2 // 2), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    ws = _workspace(lib.epi_gemm_workspace_bytes(b * (h2 // 2) * (w2 // 2), cin, 16 * cout, 1), dy.device)
    with _on(dy.device):
        ev = timer.start("epi_deconv4x4s2_bwd_data")
        _check(lib.epi_deconv4x4s2_bwd_data(_ptr(dy), _ptr(w_bwd), _ptr(dx), b, h2 // 2, w2 // 2, cin, cout, _ptr(ws), ws.numel(),
                                            _stream()), "epi_deconv4x4s2_bwd_data")
        timer.stop(ev)
    return dx


_glue = None


def glue():
    """The C++ autograd glue over the C ABI (csrc/torch_glue.cpp -> _lib/epi_torch_glue*.so), loaded once; built by
    ``epipolarpose_amd.build.build_glue()`` (``__graft_entry__.build()``).  Fails loudly when missing."""
    global _glue
    if _glue is None:
        import importlib.util
        from . import build as _build
        load()
        path = os.path.join(_LIB_DIR, os.path.basename(_build.glue_path()))
        if not os.path.exists(path):
            raise RuntimeError("epipolarpose_amd: %s is missing -- run `python -m epipolarpose_amd.build` (hipcc + g++, gfx950); "
                               "there is no Python/CPU fallback for the fused BatchNorm" % path)
        spec = importlib.util.spec_from_file_location(_build.GLUE_NAME, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        ver = load().epi_version().decode()
        if mod.abi_version() != ver:
            raise RuntimeError("epipolarpose_amd: glue / library version mismatch (%s vs %s): rebuild" % (mod.abi_version(), ver))
        _glue = mod
    return _glue


def gemm_tn_plan(r, i, j, ntap=1):
    """The plan of a weight-gradient launch (host only): dict(cfg, tiles, nsplit, rows_per_split, slab_bytes)."""
    plan = (ctypes.c_longlong * 4)()
    _check(load().epi_gemm_tn_plan(r, i, j, ntap, plan), "epi_gemm_tn_plan")
    return {"cfg": int(plan[0]), "tiles": int(plan[1]), "nsplit": int(plan[2]), "rows_per_split": int(plan[3]),
            "slab_bytes": int(plan[2]) * i * j * ntap * 4 if plan[2] > 1 else 0}


def gemm_tn_bf16(a, b):
    """C[I, J] (f32) = a[R, I]^T @ b[R, J]; a, b bf16 with unit inner stride (weight gradient of the 1x1 conv)."""
    lib = load()
    _dev(a, torch.bfloat16, "a")
    _dev(b, torch.bfloat16, "b")
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    r, i = a.shape
    j = b.shape[1]
    out = torch.empty((i, j), dtype=torch.float32, device=a.device)
    ws = _workspace(lib.epi_gemm_tn_workspace_bytes(r, i, j, 1), a.device)
    with _on(a.device):
        ev = timer.start("epi_gemm_tn_bf16")
        _check(lib.epi_gemm_tn_bf16(_ptr(a), a.stride(0), _ptr(b), b.stride(0), _ptr(out), r, i, j, _ptr(ws), ws.numel(), _stream()),
               "epi_gemm_tn_bf16")
        timer.stop(ev)
    return out


def deconv4x4s2_bwd_weight(x, dy, dtype=torch.float32):
    """x [B, Cin, H, W], dy [B, Cout, 2H, 2W] (channels_last bf16) -> dW [Cin, Cout, 4, 4] (f32 or bf16) in channels_last memory
    ([Cin][kh][kw][Cout]: the kernel's own output order, no permuting copy)."""
    lib = load()
    x, dy = _nhwc_bf16(x, "x"), _nhwc_bf16(dy, "dy")
    b, cin, h, w = x.shape
    cout = dy.shape[1]
    taps = torch.empty((cin, 16, cout), dtype=dtype, device=x.device)
    ws = _workspace(lib.epi_gemm_tn_workspace_bytes(b * h * w, cin, cout, 16), x.device)
    with _on(x.device):
        ev = timer.start("epi_deconv4x4s2_bwd_weight")
        _check(lib.epi_deconv4x4s2_bwd_weight(_ptr(x), _ptr(dy), _ptr(taps), EPI_BF16 if dtype == torch.bfloat16 else EPI_F32, b, h, w, cin, cout,
                                              _ptr(ws), ws.numel(), _stream()), "epi_deconv4x4s2_bwd_weight")
        timer.stop(ev)
    return taps.view(cin, 4, 4, cout).permute(0, 3, 1, 2)


def conv2d_bwd_weight(x, dy, kernel, stride=1, padding=0, dtype=torch.float32):
    """Weight gradient of a Conv2d (groups 1): x [B, Cin, H, W], dy [B, Cout, Ho, Wo] (channels_last bf16) -> dW [Cout, Cin, KH, KW]
    (channels_last memory), f32 or bf16."""
    lib = load()
    x, dy = _nhwc_bf16(x, "x"), _nhwc_bf16(dy, "dy")
    b, cin, h, w = x.shape
    cout = dy.shape[1]
    kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
    ho, wo = (h + 2 * padding - kh) // stride + 1, (w + 2 * padding - kw) // stride + 1
    if tuple(dy.shape) != (b, cout, ho, wo):
        raise ValueError("dy shape %s does not match the convolution geometry %s" % (tuple(dy.shape), (b, cout, ho, wo)))
    dw = torch.empty((cout, cin, kh, kw), dtype=dtype, device=x.device).contiguous(memory_format=torch.channels_last)
    ws = _workspace(lib.epi_gemm_tn_workspace_bytes(b * ho * wo, cout, cin, kh * kw), x.device)
    with _on(x.device):
        _check(lib.epi_conv2d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), EPI_BF16 if dtype == torch.bfloat16 else EPI_F32, b, h, w, cin, cout,
                                         kh, kw, stride, padding, _ptr(ws), ws.numel(), _stream()), "epi_conv2d_bwd_weight")
    return dw


class EpiWgradItem(ctypes.Structure):
    _fields_ = ([("x", _vp), ("dy", _vp), ("dw", _vp)] + [(k, _i) for k in ("dw_dtype", "kind", "B", "H", "W", "Cin", "Cout", "KH", "KW", "stride", "pad")]
                + [("x_scale_shift", _vp)])        # (round 4: the input's pending BatchNorm, epipolar_hip.h; NULL = x is the activation itself)


class EpiBnReduce(ctypes.Structure):
    """include/epipolar_hip.h EpiBnReduce: the BatchNorm-backward reduction fused into a backward-data launch."""
    _fields_ = [("z", _vp), ("y", _vp), ("bn", _vp), ("sums", _vp), ("relu", _i)]


class EpiSlabReduce(ctypes.Structure):
    _fields_ = [("slabs", _vp), ("out", _vp), ("n", ctypes.c_longlong), ("chunk_begin", ctypes.c_longlong), ("nsplit", _i), ("out_bf16", _i)]


def wgrad_group_max():
    return int(load().epi_wgrad_group_max())


def wgrad_group_plan(shapes):
    """Host only: the reduction slices and slab bytes of a grouped weight-gradient launch.  shapes: (kind, B, H, W, Cin, Cout, k, stride, pad)
    per item, kind 'conv' | 'deconv'.  Returns (slab_bytes, [nsplit per item])."""
    n = len(shapes)
    items = (EpiWgradItem * n)()
    for it, (kind, b, h, w, cin, cout, k, stride, pad) in zip(items, shapes):
        it.dw_dtype, it.kind = EPI_BF16, 1 if kind == "deconv" else 0
        it.B, it.H, it.W, it.Cin, it.Cout, it.KH, it.KW, it.stride, it.pad = b, h, w, cin, cout, k, k, stride, pad
    slab = ctypes.c_size_t(0)
    ns = (ctypes.c_int * n)()
    _check(load().epi_wgrad_group_plan(items, n, ctypes.byref(slab), ns), "epi_wgrad_group_plan")
    return int(slab.value), list(ns)


def wgrad_group(jobs, dtype=torch.float32):
    """MANY weight gradients in one grouped launch (epi_wgrad_group) + one slab sum (epi_slab_reduce_multi).
    jobs: (kind, x, dy, kernel, stride, padding) per item -- 'conv': x [B, Cin, H, W], dy [B, Cout, Ho, Wo] -> dW [Cout, Cin, k, k];
    'deconv': x [B, Cin, H, W], dy [B, Cout, 2H, 2W] -> dW [Cin, Cout, 4, 4]; channels_last bf16 operands, channels_last results."""
    lib = load()
    n = len(jobs)
    items = (EpiWgradItem * n)()
    outs, keep = [], []
    for it, (kind, x, dy, k, stride, pad) in zip(items, jobs):
        x, dy = _nhwc_bf16(x, "x"), _nhwc_bf16(dy, "dy")
        keep += [x, dy]
        b, cin, h, w = x.shape
        cout = dy.shape[1]
        if kind == "deconv":
            dw = torch.empty((cin, 16, cout), dtype=dtype, device=x.device)
            outs.append(dw.view(cin, 4, 4, cout).permute(0, 3, 1, 2))
            it.kind = 1
        else:
            dw = torch.empty((cout, cin, k, k), dtype=dtype, device=x.device).contiguous(memory_format=torch.channels_last)
            outs.append(dw)
            it.kind = 0
        keep.append(dw)
        it.x, it.dy, it.dw = _ptr(x), _ptr(dy), _ptr(dw)
        it.dw_dtype = EPI_BF16 if dtype == torch.bfloat16 else EPI_F32
        it.B, it.H, it.W, it.Cin, it.Cout, it.KH, it.KW, it.stride, it.pad = b, h, w, cin, cout, k, k, stride, pad
    dev = keep[0].device
    slab = ctypes.c_size_t(0)
    _check(lib.epi_wgrad_group_plan(items, n, ctypes.byref(slab), None), "epi_wgrad_group_plan")
    ws = torch.empty(max(int(slab.value), 256), dtype=torch.uint8, device=dev)
    pend = (EpiSlabReduce * n)()
    with _on(dev):
        _check(lib.epi_wgrad_group(items, n, _ptr(ws), ws.numel(), pend, _stream()), "epi_wgrad_group")
        rows = [p for p in pend if p.nsplit > 0]
        if rows:
            table = (EpiSlabReduce * len(rows))()
            chunks = 0
            for dst, src in zip(table, rows):
                ctypes.memmove(ctypes.byref(dst), ctypes.byref(src), ctypes.sizeof(EpiSlabReduce))
                dst.chunk_begin = chunks
                chunks += int(lib.epi_slab_reduce_chunks(src.n))
            raw = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(dev)
            _check(lib.epi_slab_reduce_multi(_ptr(raw), len(rows), chunks, _stream()), "epi_slab_reduce_multi")
            keep.append(raw)
        torch.cuda.current_stream(dev).synchronize()
    return outs


def stem_conv_fwd(x, weight, bn_sums=None):
    """The 7x7 / stride-2 / pad-3 stem convolution (pose3d_resnet.py:99): x [B, 3, H, W] f32 or bf16 (NCHW or channels_last), weight
    [Cout, 3, 7, 7] -> (y [B, Cout, H/2, W/2] channels_last bf16, s2d) on the implicit-GEMM kernel through the space-to-depth image
    (``epi_stem7x7s2_*``); ``s2d`` is what the weight gradient reads."""
    lib = load()
    _dev(x, name="x")
    b, _, h, w = x.shape
    cout = weight.shape[0]
    nhwc = x.is_contiguous(memory_format=torch.channels_last)
    if not nhwc:
        x = x.contiguous()
    w16 = weight.detach().to(torch.bfloat16)
    w_cl = w16.is_contiguous(memory_format=torch.channels_last) and not w16.is_contiguous()
    if not w_cl:
        w16 = w16.contiguous()
    s2d = torch.empty(lib.epi_stem7x7s2_s2d_bytes(b, h, w) // 2, dtype=torch.bfloat16, device=x.device)
    wp = torch.empty(cout * 256, dtype=torch.bfloat16, device=x.device)
    y = torch.empty((b, cout, h // 2, w // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ws = _workspace(lib.epi_stem7x7s2_workspace_bytes(b, h, w, cout), x.device)
    done = ctypes.c_int(0)
    _check(lib.epi_stem7x7s2_s2d(_ptr(x), EPI_F32 if x.dtype == torch.float32 else EPI_BF16, EPI_NHWC if nhwc else EPI_NCHW, b, h, w, _ptr(s2d), _stream()),
           "epi_stem7x7s2_s2d")
    _check(lib.epi_stem7x7s2_pack_weight(_ptr(w16), 1 if w_cl else 0, cout, _ptr(wp), _stream()), "epi_stem7x7s2_pack_weight")
    _check(lib.epi_stem7x7s2_fwd(_ptr(s2d), _ptr(wp), _ptr(y), b, h, w, cout, _ptr(bn_sums), ctypes.addressof(done) if bn_sums is not None else None,
                                 _ptr(ws), ws.numel(), _stream()), "epi_stem7x7s2_fwd")
    return y, s2d


def stem_conv_bwd_weight(s2d, dy, image_hw, dtype=torch.float32):
    """Weight gradient of the stem convolution from the space-to-depth image: dy [B, Cout, H/2, W/2] channels_last bf16 -> dW [Cout, 3, 7, 7]."""
    lib = load()
    dy = _nhwc_bf16(dy, "dy")
    b, cout = dy.shape[0], dy.shape[1]
    h, w = image_hw
    dwp = torch.empty(cout * 256, dtype=torch.float32, device=dy.device)
    dw = torch.empty((cout, 3, 7, 7), dtype=dtype, device=dy.device)
    ws = _workspace(lib.epi_stem7x7s2_workspace_bytes(b, h, w, cout), dy.device)
    _check(lib.epi_stem7x7s2_bwd_weight(_ptr(s2d), _ptr(dy), _ptr(dwp), b, h, w, cout, _ptr(ws), ws.numel(), _stream()), "epi_stem7x7s2_bwd_weight")
    _check(lib.epi_stem7x7s2_unpack_weight_grad(_ptr(dwp), cout, 0, _ptr(dw), EPI_BF16 if dtype == torch.bfloat16 else EPI_F32, _stream()),
           "epi_stem7x7s2_unpack_weight_grad")
    return dw


def _cl_weight_bf16(weight):
    """[Cout, Cin, KH, KW] weight -> bf16 tensor whose MEMORY is [Cout][KH][KW][Cin] (channels_last)."""
    _dev(weight, name="weight")
    w = weight.detach()
    if w.dtype != torch.bfloat16:
        w = w.to(torch.bfloat16)
    return w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)


def conv3x3_patch_mode(mode=-1):
    """Kernel choice for 3x3 / stride-1 convolutions (0 generic gather kernel, 1 patch kernel where it needs no channel split, 2 patch
    kernel always -- the default); sets ``mode`` when 0 .. 2 and returns the previous mode (epipolar_hip.h)."""
    return int(load().epi_conv3x3_patch_mode(int(mode)))


def set_deterministic(on=True):
    """The reference's ``CUDNN.DETERMINISTIC`` (lib/core/config.py:21; scripts/train.py:80 hands it to ``torch.backends.cudnn.deterministic``):
    every cross-workgroup sum of the library in a fixed order -- BatchNorm statistics and the BatchNorm-backward sums by their own passes with
    ordered partial sums instead of fp32 atomics from the GEMM epilogues (``epi_set_deterministic``, epipolar_hip.h).  The weight gradients stay
    on their second stream (their values do not depend on timing).  Two runs of the same step are then bit-identical
    (tests/test_hip_deterministic.py).  Returns the previous setting."""
    before = int(load().epi_set_deterministic(1 if on else 0))
    if before < 0:
        raise RuntimeError("epi_set_deterministic: could not allocate the partial-sum scratch")
    return bool(before)


def is_deterministic():
    return bool(load().epi_set_deterministic(-1))


def bn_sum_copies(channels):
    """Accumulator copies of a BatchNorm layer's forward statistics buffer (``sums_ws`` is [copies][2C] f32, epipolar_hip.h)."""
    return 4 if channels <= 512 else 1


def conv2d_fwd(x, weight, stride=1, padding=0, bn_sums=None):
    """x [B, Cin, H, W] channels_last bf16, weight [Cout, Cin, KH, KW] (channels_last memory) -> y [B, Cout, Ho, Wo]
    channels_last bf16.  Reference: the nn.Conv2d calls of pose3d_resnet.py:21-88.  ``bn_sums`` (zeroed f32 [bn_sum_copies(Cout) * 2*Cout]): asks for
    the per-channel (sum, sum of squares) of the result from the GEMM epilogue; returns (y, done) then."""
    lib = load()
    x = _nhwc_bf16(x, "x")
    w = _cl_weight_bf16(weight)
    b, cin, h, wd = x.shape
    cout, _, kh, kw = w.shape
    ho, wo = (h + 2 * padding - kh) // stride + 1, (wd + 2 * padding - kw) // stride + 1
    y = torch.empty((b, cout, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ws = _workspace(lib.epi_conv2d_workspace_bytes(b, h, wd, cin, cout, kh, kw, stride, padding), x.device)
    ev = timer.start("epi_conv2d_fwd")
    done = ctypes.c_int(0)
    _check(lib.epi_conv2d_fwd(_ptr(x), _ptr(w), _ptr(y), b, h, wd, cin, cout, kh, kw, stride, padding, _ptr(bn_sums),
                              ctypes.addressof(done) if bn_sums is not None else None, _ptr(ws), ws.numel(), _stream()), "epi_conv2d_fwd")
    timer.stop(ev)
    return y if bn_sums is None else (y, bool(done.value))


def conv2d_pack_weight_bwd(weight, stride=1, padding=0, out=None):
    """weight [Cout, Cin, KH, KW] -> the backward-data operand (flat bf16, Cin*KH*KW*Cout elements; see epipolar_hip.h)."""
    lib = load()
    w = _cl_weight_bf16(weight)
    cout, cin, kh, kw = w.shape
    if out is None:
        out = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
    _check(lib.epi_conv2d_pack_weight_bwd(_ptr(w), cout, cin, kh, kw, stride, padding, _ptr(out), _stream()), "epi_conv2d_pack_weight_bwd")
    return out


def conv2d_bwd_data_bnred(dy, w_bwd, in_shape, kernel, stride, padding, z, bn, relu=True, y=None, addend=None, addend_step=1):
    """``conv2d_bwd_data`` with the BatchNorm-backward reduction of the layer that produced this convolution's input in the epilogue
    (epi_conv2d_bwd_data_bnred).  z: that layer's raw output (shape ``in_shape``, channels_last bf16); bn: f32 [4, C] = mean | rstd |
    scale | shift; y: its saved output when the mask comes from there; addend_step 2: ``addend`` is the half-resolution tensor of the even pixels.  Returns (dx or dz, sums [2, C] f32, fused: bool) -- fused False:
    dx is the plain gradient and sums is zero."""
    lib = load()
    dy = _nhwc_bf16(dy, "dy")
    z = _nhwc_bf16(z, "z")
    b, cin, h, wd = in_shape
    cout = dy.shape[1]
    kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
    if tuple(z.shape) != (b, cin, h, wd) or tuple(bn.shape) != (4, cin) or bn.dtype != torch.float32 or not bn.is_contiguous():
        raise ValueError("z must have the shape of dx and bn must be f32 [4, Cin]")
    dx = torch.empty((b, cin, h, wd), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    sums = torch.zeros(2, cin, dtype=torch.float32, device=dy.device)
    ws = _workspace(lib.epi_conv2d_workspace_bytes(b, h, wd, cin, cout, kh, kw, stride, padding), dy.device)
    if addend is not None:
        addend = _nhwc_bf16(addend, "addend")
    if y is not None:
        y = _nhwc_bf16(y, "y")
    red = EpiBnReduce(_ptr(z), _ptr(y), _ptr(bn), _ptr(sums), 1 if relu else 0)
    done = ctypes.c_int(0)
    with _on(dy.device):
        _check(lib.epi_conv2d_bwd_data_bnred(_ptr(dy), _ptr(w_bwd), _ptr(dx), b, h, wd, cin, cout, kh, kw, stride, padding, _ptr(addend), addend_step,
                                             ctypes.byref(red), ctypes.byref(done), _ptr(ws), ws.numel(), _stream()), "epi_conv2d_bwd_data_bnred")
    return dx, sums, bool(done.value)


def conv2d_bwd_data(dy, w_bwd, in_shape, kernel, stride=1, padding=0, addend=None):
    """dy [B, Cout, Ho, Wo] channels_last bf16, w_bwd from ``conv2d_pack_weight_bwd`` -> dx of shape ``in_shape`` = (B, Cin, H, W);
    ``addend`` (same shape as dx, channels_last bf16) is added in the epilogue (the other branch's gradient at a residual junction)."""
    lib = load()
    dy = _nhwc_bf16(dy, "dy")
    b, cin, h, wd = in_shape
    cout = dy.shape[1]
    kh, kw = (kernel, kernel) if isinstance(kernel, int) else kernel
    dx = torch.empty((b, cin, h, wd), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    ws = _workspace(lib.epi_conv2d_workspace_bytes(b, h, wd, cin, cout, kh, kw, stride, padding), dy.device)
    ev = timer.start("epi_conv2d_bwd_data")
    if addend is not None:
        addend = _nhwc_bf16(addend, "addend")
        if tuple(addend.shape) != tuple(dx.shape):
            raise ValueError("addend must have the shape of dx")
    _check(lib.epi_conv2d_bwd_data(_ptr(dy), _ptr(w_bwd), _ptr(dx), b, h, wd, cin, cout, kh, kw, stride, padding, _ptr(addend), _ptr(ws),
                                   ws.numel(), _stream()), "epi_conv2d_bwd_data")
    timer.stop(ev)
    return dx


def column_sum_bf16(x):
    """x [R, C] bf16 -> per-column sums [C] f32."""
    lib = load()
    _dev(x, torch.bfloat16, "x")
    x = x if x.is_contiguous() else x.contiguous()
    r, c = x.shape
    sums = torch.zeros(2 * c, dtype=torch.float32, device=x.device)
    with _on(x.device):
        _check(lib.epi_column_sums_bf16(_ptr(x), r, c, _ptr(sums), _stream()), "epi_column_sums_bf16")
    return sums[:c]


def adam_step(table_dev, chunks_dev, nchunks, lr, beta1, beta2, eps, step):
    """One fused Adam update over every tensor described by the device-resident table (see optim.FusedAdam)."""
    lib = load()
    ev = timer.start("epi_adam_step")
    _check(lib.epi_adam_step(_ptr(table_dev), _ptr(chunks_dev), nchunks, lr, beta1, beta2, eps, step, _stream()), "epi_adam_step")
    timer.stop(ev)


def adam_step_clipped(table_dev, chunks_dev, nchunks, lr, beta1, beta2, eps, step, max_norm, norm_sq):
    """Fused Adam with clip_grad_norm_(max_norm) folded in; ``norm_sq`` (zeroed f32 device scalar) receives the squared total norm."""
    lib = load()
    _check(lib.epi_adam_step_clipped(_ptr(table_dev), _ptr(chunks_dev), nchunks, lr, beta1, beta2, eps, step, max_norm, _ptr(norm_sq),
                                     _stream()), "epi_adam_step_clipped")


def dropout_bf16(x, p, seed, out=None):
    """Inverted dropout of a bf16 tensor with a stateless (seed, index) mask; the same call on the gradient is the backward."""
    lib = load()
    _dev(x, torch.bfloat16, "x")
    x = x.contiguous()
    out = torch.empty_like(x) if out is None else out
    _check(lib.epi_dropout_bf16(_ptr(x), _ptr(out), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()), "epi_dropout_bf16")
    return out


def maxpool3x3s2_fwd(x):
    """nn.MaxPool2d(3, 2, 1) of a channels_last bf16 tensor [B, C, H, W] -> (y [B, C, Ho, Wo] channels_last bf16, pos uint8 [B, Ho, Wo, C]:
    the window position every output element came from -- what ``maxpool3x3s2_bwd`` needs).  Reference: pose3d_resnet.py:104,186."""
    lib = load()
    x = _nhwc_bf16(x, "x")
    b, c, h, w = x.shape
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = torch.empty((b, c, ho, wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    pos = torch.empty((b, ho, wo, c), dtype=torch.uint8, device=x.device)
    with _on(x.device):
        _check(lib.epi_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(pos), b, h, w, c, _stream()), "epi_maxpool3x3s2_fwd")
    return y, pos


def maxpool3x3s2_bwd(dy, pos, in_hw):
    """dy [B, C, Ho, Wo] channels_last bf16, pos from the forward pass -> dx [B, C, H, W] channels_last bf16."""
    lib = load()
    dy = _nhwc_bf16(dy, "dy")
    _dev(pos, torch.uint8, "pos")
    b, c = dy.shape[0], dy.shape[1]
    h, w = in_hw
    dx = torch.empty((b, c, h, w), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    with _on(dy.device):
        _check(lib.epi_maxpool3x3s2_bwd(_ptr(dy), _ptr(pos), _ptr(dx), b, h, w, c, _stream()), "epi_maxpool3x3s2_bwd")
    return dx


def evaluate_poses(pred_img, gt_img, pelvis_z, fl, c_p, root, j14):
    """pred_img / gt_img [N, J, 3] f64 CUDA (image coordinates + root-relative depth) -> (metrics [N, 9] f64, per_joint [N, J] f64).
    Reference: lib/dataset/h36m.py:168-378."""
    lib = load()
    for t in (pred_img, gt_img, pelvis_z, fl, c_p):
        _dev(t, torch.float64)
    pred_img, gt_img = pred_img.contiguous(), gt_img.contiguous()
    n, j, _ = pred_img.shape
    j14_t = torch.tensor(list(j14), dtype=torch.int32, device=pred_img.device)
    metrics = torch.empty((n, 9), dtype=torch.float64, device=pred_img.device)
    per_joint = torch.empty((n, j), dtype=torch.float64, device=pred_img.device)
    _check(lib.epi_evaluate_poses(_ptr(pred_img), _ptr(gt_img), _ptr(pelvis_z.contiguous()), _ptr(fl.contiguous()), _ptr(c_p.contiguous()),
                                  n, j, root, _ptr(j14_t), len(j14), _ptr(metrics), _ptr(per_joint), _stream()), "epi_evaluate_poses")
    return metrics, per_joint


def crop_patches(frames, frame_offset, frame_hw, trans, patch_h, patch_w, do_flip=None, color_scale=None, mean=None, std=None,
                 dtype=torch.float32, channels_last=False, occluders=None, placements=None):
    """Batched crop + warp + normalise (``epi_crop_patches``).  frames: uint8 CUDA tensor holding the BGR frames back to back;
    frame_offset int64 [B]; frame_hw int32 [B, 2]; trans float64 [B, 2, 3] (forward affine frame -> patch).  -> [B, 3, ph, pw]
    (logical NCHW; channels_last memory when asked), RGB order.  Reference: img_utils.py:114-127,265-279.
    ``occluders`` = (bank uint8 CUDA bytes of the RGBA occluder images, offsets int64 [N], hw int32 [N, 2]) and ``placements`` int32
    [B, max_occ, 5] (index | -1, pasted w, h, x0, y0) add the synthetic-occlusion stage (augmentation.py:61-114, ``epi_crop_patches_occluded``)."""
    lib = load()
    _dev(frames, torch.uint8, "frames")
    frame_offset = _dev(frame_offset, torch.int64, "frame_offset").contiguous()
    frame_hw = _dev(frame_hw, torch.int32, "frame_hw").contiguous()
    trans = _dev(trans, torch.float64, "trans").contiguous()
    b = trans.shape[0]
    if do_flip is not None:
        do_flip = _dev(do_flip, name="do_flip").to(torch.int32).contiguous()
    if color_scale is not None:
        color_scale = _dev(color_scale, name="color_scale").to(torch.float32).contiguous()
    out = torch.empty((b, 3, patch_h, patch_w), dtype=dtype, device=frames.device,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    m3 = (ctypes.c_float * 3)(*[float(v) for v in mean]) if mean is not None else None
    s3 = (ctypes.c_float * 3)(*[float(v) for v in std]) if std is not None else None
    if occluders is not None and placements is not None:
        bank, occ_off, occ_hw = occluders
        _dev(bank, torch.uint8, "occluder bank")
        occ_off = _dev(occ_off, torch.int64, "occluder offsets").contiguous()
        occ_hw = _dev(occ_hw, torch.int32, "occluder sizes").contiguous()
        placements = _dev(placements, torch.int32, "placements").contiguous()
        if placements.dim() != 3 or placements.shape[0] != b or placements.shape[2] != 5:
            raise ValueError("placements must be int32 [B, max_occ, 5]")
        _check(lib.epi_crop_patches_occluded(_ptr(frames), _ptr(frame_offset), _ptr(frame_hw), _ptr(trans), _ptr(do_flip), _ptr(color_scale), m3, s3,
                                             b, patch_h, patch_w, _ptr(bank), _ptr(occ_off), _ptr(occ_hw), _ptr(placements), placements.shape[1],
                                             _ptr(out), EPI_BF16 if dtype == torch.bfloat16 else EPI_F32,
                                             EPI_NHWC if channels_last else EPI_NCHW, _stream()), "epi_crop_patches_occluded")
        return out
    _check(lib.epi_crop_patches(_ptr(frames), _ptr(frame_offset), _ptr(frame_hw), _ptr(trans), _ptr(do_flip), _ptr(color_scale), m3, s3, b,
                                patch_h, patch_w, _ptr(out), EPI_BF16 if dtype == torch.bfloat16 else EPI_F32,
                                EPI_NHWC if channels_last else EPI_NCHW, _stream()), "epi_crop_patches")
    return out
