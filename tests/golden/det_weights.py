"""Deterministic, name-keyed pseudo-random weights so that the golden generator (reference model, this
container) and the tests (our model / the oracle, any box) can build bit-identical state dicts."""
import zlib

import numpy as np
import torch


def _rng(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def fill_state_dict(shapes, seed=0, head_std=None):
    """shapes: ordered {key: shape}.  Returns {key: torch tensor} (fp32; int64 for num_batches_tracked).
    ``head_std``: standard deviation of the deconvolution / final-layer weights (the reference initialises them with
    N(0, 0.001), pose3d_resnet.py:222-239); None = He scaling like the backbone (logits of several hundred: a practically
    one-hot soft-argmax, whose gradient no reduced-precision run can reproduce)."""
    out = {}
    for key, shape in shapes.items():
        r = _rng(key, seed)
        shape = tuple(shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros(shape, dtype=torch.int64)
        elif key.endswith("running_var"):
            out[key] = torch.from_numpy(r.uniform(0.5, 1.5, size=shape).astype(np.float32))
        elif key.endswith("running_mean"):
            out[key] = torch.from_numpy((0.1 * r.standard_normal(size=shape)).astype(np.float32))
        elif len(shape) == 1 and key.endswith("weight"):
            out[key] = torch.from_numpy(r.uniform(0.5, 1.5, size=shape).astype(np.float32))
        elif key.endswith("bias"):
            out[key] = torch.from_numpy((0.1 * r.standard_normal(size=shape)).astype(np.float32))
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if "deconv" in key:                      # ConvTranspose weight is [Cin, Cout, k, k]
                fan_in = shape[0] * shape[2] * shape[3] // 4
            std = (2.0 / max(fan_in, 1)) ** 0.5
            if head_std is not None and ("deconv" in key or key.startswith("final_layer")):
                std = head_std
            out[key] = torch.from_numpy((std * r.standard_normal(size=shape)).astype(np.float32))
    return out


def seeded_array(tag, shape, seed=0, dtype=np.float32, scale=1.0):
    return (scale * _rng(tag, seed).standard_normal(size=tuple(shape))).astype(dtype)
