"""epipolarpose_amd -- MI355X-native implementation of the EpipolarPose training hot path.

Sub-packages mirror the reference's ``lib`` package for the modules on the hot path:
``core.integral_loss``, ``core.function``, ``core.config``, ``core.inference``, ``models.pose3d_resnet``,
``utils.img_utils``, ``utils.triangulation``, ``utils.utils``.  All device arithmetic goes through the C ABI of
``libepipolar_hip.so`` (``hip.py``).
"""
import importlib
import sys

__version__ = "0.1.0"

_LIB_ALIASES = {
    "lib": "epipolarpose_amd",
    "lib.core": "epipolarpose_amd.core",
    "lib.core.integral_loss": "epipolarpose_amd.core.integral_loss",
    "lib.core.function": "epipolarpose_amd.core.function",
    "lib.core.config": "epipolarpose_amd.core.config",
    "lib.core.inference": "epipolarpose_amd.core.inference",
    "lib.models": "epipolarpose_amd.models",
    "lib.models.pose3d_resnet": "epipolarpose_amd.models.pose3d_resnet",
    "lib.utils": "epipolarpose_amd.utils",
    "lib.utils.img_utils": "epipolarpose_amd.utils.img_utils",
    "lib.utils.triangulation": "epipolarpose_amd.utils.triangulation",
    "lib.utils.utils": "epipolarpose_amd.utils.utils",
    "lib.utils.prep_h36m": "epipolarpose_amd.utils.prep_h36m",
    "lib.utils.cameras": "epipolarpose_amd.utils.cameras",
    "refiner": "epipolarpose_amd.refiner",
    "refiner.model": "epipolarpose_amd.refiner.model",
    "refiner.utils": "epipolarpose_amd.refiner.utils",
    "refiner.main": "epipolarpose_amd.refiner.main",
    "refiner.data": "epipolarpose_amd.refiner.data",
    "lib.utils.augmentation": "epipolarpose_amd.utils.augmentation",
    "lib.dataset": "epipolarpose_amd.dataset",
    "lib.dataset.h36m": "epipolarpose_amd.dataset.h36m",
}


def install_as_lib():
    """Register this package under the reference's import names (``import lib.core.integral_loss as loss`` ...), so a
    script written against the reference (scripts/train.py:14-26) picks up the MI355X path unchanged."""
    for alias, target in _LIB_ALIASES.items():
        sys.modules[alias] = importlib.import_module(target)
    return sys.modules["lib"]
