"""Hard arg-max decoding -- mirror of the reference's ``lib/core/inference.py:12-40`` (``get_max_preds``).

The row arg-max runs on the GPU (``epi_argmax_rows``: first maximum, NumPy tie rule); only the [B,J] indices and
maxima come back to the host, where the (x, y) split and the ``max > 0`` mask are applied as in the reference.
"""
import numpy as np
import torch

from .. import hip


def get_max_preds_device(batch_heatmaps):
    """[B,J,H,W] CUDA tensor -> (preds f32 [B,J,2], maxvals f32 [B,J,1], idx int64 [B,J]) on the device."""
    assert batch_heatmaps.dim() == 4, 'batch_images should be 4-ndim'
    b, j, h, w = batch_heatmaps.shape
    hm = batch_heatmaps if batch_heatmaps.dtype in (torch.float32, torch.bfloat16) else batch_heatmaps.float()
    idx, val = hip.argmax_rows(hm.reshape(b * j, h * w))
    idx = idx.reshape(b, j)
    val = val.reshape(b, j, 1)
    preds = torch.stack((idx % w, torch.div(idx, w, rounding_mode="floor")), dim=2).to(torch.float32)
    preds = preds * (val > 0.0).to(torch.float32)            # inference.py:35-38
    return preds, val, idx


def get_max_preds(batch_heatmaps):
    """Reference signature: numpy in ([B,J,H,W]) -> (preds f32 [B,J,2], maxvals [B,J,1]); computed on cuda:current."""
    assert isinstance(batch_heatmaps, np.ndarray), 'batch_heatmaps should be numpy.ndarray'
    assert batch_heatmaps.ndim == 4, 'batch_images should be 4-ndim'
    t = torch.from_numpy(np.ascontiguousarray(batch_heatmaps, dtype=np.float32)).cuda()
    preds, val, _ = get_max_preds_device(t)
    return preds.cpu().numpy(), val.cpu().numpy().astype(batch_heatmaps.dtype)
