"""Datasets -- ``lib.dataset`` of the reference exposes ``h36m`` / ``mpii_integral`` (lib/dataset/__init__.py:11-12); there is
no H36M / MPII data on the build and GPU boxes, so both names resolve to the synthetic stand-in that honours the reference's
constructor signature, item contract, ``db`` record format and ``evaluate``."""
from .collate import tri_batch_to_view_major, view_major_collate  # noqa: F401
from .synthetic import SyntheticH36M  # noqa: F401
from .synthetic import SyntheticH36M as h36m  # noqa: F401
from .synthetic import SyntheticH36M as mpii_integral  # noqa: F401
