"""Datasets -- ``lib.dataset`` of the reference exposes ``h36m`` / ``mpii_integral`` (lib/dataset/__init__.py:11-12).

``h36m(cfg, root, image_set, is_train)`` reads the reference's on-disk format (``<root>/annot/<image_set>.pkl`` with pickled ``Camera`` objects + JPEG
frames: ``dataset/h36m.py``) where that file exists; there is no H36M / MPII data on the build and GPU boxes, so otherwise -- and for ``mpii_integral`` --
the name resolves to the synthetic stand-in that honours the reference's constructor signature, item contract, ``db`` record format and ``evaluate``."""
from .collate import tri_batch_to_view_major, view_major_collate  # noqa: F401
from .h36m import H36M_Integral, H36MFrames, h36m  # noqa: F401
from .synthetic import SyntheticH36M  # noqa: F401
from .synthetic import SyntheticH36M as mpii_integral  # noqa: F401
