"""Batch layout for the self-supervised (``DATASET.TRI``) mode.

In TRI mode the reference dataset returns ``{'cam_1': bundle, 'cam_2': bundle}`` per item (h36m.py:32-47: a random camera and
one of its neighbours), and ``triangulate`` pairs sample ``i`` with sample ``i + B/2`` (img_utils.py:194-199): the batch must be
VIEW-MAJOR -- every cam_1 sample first, then every cam_2 sample in the same order.  A stock ``DataLoader`` collates the item
dicts into ``{'cam_1': [img, label, weight, meta], 'cam_2': [...]}``; ``tri_batch_to_view_major`` turns that (or a list of items,
as a ``collate_fn``) into the ``(data, label, weight, meta)`` tuple ``train_integral`` consumes.
"""
import torch
from torch.utils.data._utils.collate import default_collate

VIEW_KEYS = ("cam_1", "cam_2")


def _cat_meta(metas):
    out = {}
    for k in metas[0]:
        vals = [m[k] for m in metas]
        if isinstance(vals[0], torch.Tensor):
            out[k] = torch.cat(vals, dim=0)
        else:                                   # lists of strings (image paths) and the like
            out[k] = [x for v in vals for x in v]
    return out


def tri_batch_to_view_major(collated):
    """{'cam_1': (img, label, weight, meta), 'cam_2': (...)} (each already batched) -> (img, label, weight, meta) with the
    cam_1 half first.  Works for any number of views named cam_1 .. cam_V."""
    keys = sorted((k for k in collated if k.startswith("cam_")), key=lambda k: int(k[4:]))
    bundles = [collated[k] for k in keys]
    data = torch.cat([b[0] for b in bundles], dim=0)
    label = torch.cat([b[1] for b in bundles], dim=0)
    weight = torch.cat([b[2] for b in bundles], dim=0)
    return data, label, weight, _cat_meta([b[3] for b in bundles])


def view_major_collate(items):
    """``collate_fn`` for a DataLoader over a TRI dataset: a list of per-item dicts -> the view-major batch tuple.  Plain
    (non-TRI) items fall through to ``default_collate``."""
    if isinstance(items[0], dict) and all(k.startswith("cam_") for k in items[0]):
        return tri_batch_to_view_major(default_collate(items))
    return default_collate(items)
