"""Oracle (test infrastructure): hard arg-max decoding of heat-maps.

Restates ``/root/reference/lib/core/inference.py:12-40`` (``get_max_preds``) in NumPy.
Not imported by the product.
"""
import numpy as np


def argmax_rows(rows):
    """First-maximum flat index and max value per row (np.argmax tie rule, inference.py:25-26)."""
    rows = np.asarray(rows)
    flat = rows.reshape(rows.shape[0], rows.shape[1], -1)
    return np.argmax(flat, axis=2), np.amax(flat, axis=2)


def get_max_preds(batch_heatmaps):
    """inference.py:12-40: [B,J,H,W] -> (preds f32 [B,J,2] = (idx % W, floor(idx / W)), maxvals [B,J,1]).

    Coordinates are zeroed where the maximum is <= 0 (:35-38).
    """
    hm = np.asarray(batch_heatmaps)
    assert hm.ndim == 4
    width = hm.shape[3]
    idx, maxvals = argmax_rows(hm)
    preds = np.stack([idx % width, idx // width], axis=2).astype(np.float32)
    mask = (maxvals > 0.0).astype(np.float32)[:, :, None]
    return preds * mask, maxvals[:, :, None]
