"""GPU input pipeline on synthetic camera frames.

The reference's loader decodes a JPEG per sample on a CPU worker and runs ``get_single_patch_sample`` on it
(lib/dataset/h36m.py:53-88 -> lib/utils/img_utils.py:246-298: augmentation draw, ``cv2.warpAffine`` crop, BGR -> RGB, optional
synthetic occlusion, colour scaling, normalisation, label arithmetic).  Here the decoded frames live in HBM as uint8 BGR and a whole
multi-view batch goes through ONE launch (``epi_crop_patches_occluded``); the host draws the augmentation parameters in the
reference's order and does the label arithmetic for the batch in vectorised NumPy (a few hundred bytes per sample).

``SyntheticFrames`` renders H36M-like 1000 x 1000 frames of the synthetic multi-view scenes (``epipolarpose_amd.synthetic``): a
textured background with the skeleton drawn on top -- limbs and joints coloured by index, so the image carries the pose.
``FramePatchLoader`` yields ``(data, label, weight, meta)`` batches in the view-major order ``train_integral`` expects, ``data``
already on the device in the network's layout (NHWC bf16 by default).
"""
import random

import numpy as np
import torch

from .. import hip
from ..synthetic import RECT_3D, SyntheticScenes, project
from ..utils import augmentation as aug
from ..utils.img_utils import IMAGENET_MEAN, IMAGENET_STD, do_augmentation, patch_affines_batch, patch_labels_batch

PARENTS_17 = (0, 0, 1, 2, 0, 4, 5, 0, 7, 8, 9, 8, 11, 12, 8, 14, 15)       # lib/dataset/h36m.py:23


def _palette(n):
    rng = np.random.default_rng(12345)
    return rng.integers(40, 256, size=(n, 3)).astype(np.uint8)


def render_frame(uv, size=1000, seed=0):
    """One BGR uint8 frame [size, size, 3]: low-frequency textured background + limbs (thick segments) + joints (discs)."""
    rng = np.random.default_rng(seed)
    coarse = rng.integers(60, 190, size=(size // 50 + 1, size // 50 + 1, 3)).astype(np.float32)
    frame = np.repeat(np.repeat(coarse, 50, axis=0), 50, axis=1)[:size, :size]
    frame += rng.normal(0, 6.0, size=frame.shape).astype(np.float32)
    frame = np.clip(frame, 0, 255).astype(np.uint8)
    j = len(uv)
    pal = _palette(2 * j)

    def disc(cx, cy, r, colour):
        x0, x1, y0, y1 = int(max(0, cx - r)), int(min(size, cx + r + 1)), int(max(0, cy - r)), int(min(size, cy + r + 1))
        if x0 >= x1 or y0 >= y1:
            return
        yy, xx = np.mgrid[y0:y1, x0:x1]
        frame[y0:y1, x0:x1][(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = colour
    parents = PARENTS_17 if j == 17 else tuple(max(0, k - 1) for k in range(j))
    for k in range(1, j):
        a, b = uv[k], uv[parents[k]]
        n = int(max(2, np.hypot(*(a - b)) / 3))
        for t in np.linspace(0.0, 1.0, n):
            p = a * (1 - t) + b * t
            disc(p[0], p[1], 4, pal[j + k])
    for k in range(j):
        disc(uv[k][0], uv[k][1], 9, pal[k])
    return frame


class SyntheticFrames:
    """``n_group`` scenes x ``n_view`` cameras: frames in HBM + the H36M-style records ``get_single_patch_sample`` reads."""

    def __init__(self, n_group, n_view=4, num_joints=17, frame_size=1000, seed=0, device=None):
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.n_group, self.n_view, self.num_joints = n_group, n_view, num_joints
        self.scenes = SyntheticScenes(n_group=n_group, n_view=n_view, num_joints=num_joints, seed=seed, augment=False)
        sc = self.scenes
        b = sc.batch_size
        self.joints = np.zeros((b, num_joints, 3))          # (u, v, root-relative depth mm): db['joints_3d'] (h36m.py:57)
        frames = []
        for i in range(b):
            v, g = divmod(i, n_group)
            uv, xc = project(sc.world[g], sc.cams[v])
            self.joints[i] = np.concatenate([uv, xc[:, 2:3] - xc[0, 2]], axis=1)
            frames.append(render_frame(uv, frame_size, seed=seed * 100003 + i))
        self.frame_hw_host = np.array([[f.shape[0], f.shape[1]] for f in frames], np.int32)
        self.frame_offset_host = (np.arange(b, dtype=np.int64) * frame_size * frame_size * 3)
        self.frames_host = frames
        self.frames = torch.from_numpy(np.stack(frames).reshape(-1)).to(self.device)
        self.joints_vis = np.ones((b, num_joints, 3))

    def index(self, view, group):
        return view * self.n_group + group


class FramePatchLoader:
    """Batches of ``groups_per_batch`` scenes x all views, view-major.  ``augment``: img_utils.py:29-39 per sample; ``occlusion``:
    augmentation.py:61-81 per sample (procedural occluders).  ``np_rng`` / ``py_rng`` default to private seeded generators."""

    def __init__(self, frames, groups_per_batch, patch=256, augment=True, occlusion=False, seed=0, dtype=torch.bfloat16, channels_last=True,
                 shuffle=True, occluders=None):
        self.frames, self.gpb, self.patch = frames, groups_per_batch, int(patch)
        self.augment, self.occlusion = augment, occlusion
        self.np_rng, self.py_rng = np.random.RandomState(seed), random.Random(seed)
        self.dtype, self.channels_last, self.shuffle = dtype, channels_last, shuffle
        self.bank = None
        if occlusion:
            self.bank = aug.OccluderBank(occluders if occluders is not None else aug.load_occluders(seed=seed), frames.device)
        self.dataset = frames            # (train_integral only uses len(loader))
        self._slots = None

    def __len__(self):
        return self.frames.n_group // self.gpb

    def sample_parameters(self, idx):
        """Host side of get_single_patch_sample for the samples ``idx``: -> dict of arrays (scale, rot, color, trans, label, weight, placements)."""
        fr, sc = self.frames, self.frames.scenes
        b = len(idx)
        scale, rot, color = np.ones(b), np.zeros(b), np.ones((b, 3), np.float32)
        place = np.full((b, aug.MAX_OCCLUDERS, 5), -1, np.int32) if self.occlusion else None
        for k in range(b):                      # the draws, sample by sample in the reference's order (augmentation, then occlusion)
            if self.augment:
                scale[k], rot[k], _, cs = do_augmentation(self.np_rng, self.py_rng)
                color[k] = cs
            if self.occlusion:
                place[k] = aug.draw_occlusion((self.patch, self.patch), self.bank.hw_host, self.np_rng, self.py_rng)
        cx, cy = sc.meta["center_x"][idx], sc.meta["center_y"][idx]
        w, h = sc.meta["width"][idx], sc.meta["height"][idx]
        trans = patch_affines_batch(cx, cy, w, h, self.patch, self.patch, scale, rot)
        label, weight = patch_labels_batch(fr.joints[idx], fr.joints_vis[idx], trans, w, scale, self.patch, self.patch, RECT_3D)
        return {"scale": scale, "rot": rot, "color": color, "trans": trans, "label": label, "weight": weight, "place": place}

    def _slot(self, b, j3):
        """Pinned staging + device buffer for one batch's parameters: ONE asynchronous H2D copy per batch (seven small pageable copies, each
        synchronous for the host and ordered behind the previous step's kernels, cost 0.6 ms of a 6.8 ms step).  Four slots in rotation:
        a slot is rewritten only after the copy issued from it four batches ago has completed."""
        layout, off = {}, 0
        for name, dtype, shape in (("offset", np.int64, (b,)), ("trans", np.float64, (b, 2, 3)), ("hw", np.int32, (b, 2)),
                                   ("color", np.float32, (b, 3)), ("place", np.int32, (b, aug.MAX_OCCLUDERS, 5)), ("label", np.float32, (b, j3)),
                                   ("weight", np.float32, (b, j3))):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            layout[name] = (off, n, dtype, shape)
            off = (off + n + 15) // 16 * 16
        if self._slots is None or self._slots["nbytes"] != off:
            self._slots = {"nbytes": off, "next": 0, "layout": layout,
                           "host": [torch.empty(off, dtype=torch.uint8).pin_memory() for _ in range(4)],
                           "dev": [torch.empty(off, dtype=torch.uint8, device=self.frames.device) for _ in range(4)],
                           "event": [None] * 4}
        sl = self._slots
        k = sl["next"]
        sl["next"] = (k + 1) % 4
        if sl["event"][k] is not None:
            sl["event"][k].synchronize()
        return k

    def batch(self, groups):
        fr, sc = self.frames, self.frames.scenes
        idx = np.array([fr.index(v, g) for v in range(fr.n_view) for g in groups])            # view-major (img_utils.py:194-199)
        p = self.sample_parameters(idx)
        b, j3 = len(idx), p["label"].shape[1]
        k = self._slot(b, j3)
        sl = self._slots
        host, dev = sl["host"][k].numpy(), sl["dev"][k]
        values = {"offset": fr.frame_offset_host[idx], "trans": p["trans"], "hw": fr.frame_hw_host[idx], "color": p["color"],
                  "place": p["place"] if self.occlusion else np.full((b, aug.MAX_OCCLUDERS, 5), -1, np.int32), "label": p["label"], "weight": p["weight"]}
        views = {}
        for name, (off, n, dtype, shape) in sl["layout"].items():
            host[off:off + n].view(dtype).reshape(shape)[...] = values[name]
            views[name] = dev[off:off + n].view(getattr(torch, np.dtype(dtype).name)).reshape(shape)
        dev.copy_(sl["host"][k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        sl["event"][k] = ev
        data = hip.crop_patches(fr.frames, views["offset"], views["hw"], views["trans"], self.patch, self.patch, color_scale=views["color"],
                                mean=IMAGENET_MEAN, std=IMAGENET_STD, dtype=self.dtype, channels_last=self.channels_last,
                                occluders=self.bank.tensors() if self.occlusion else None, placements=views["place"] if self.occlusion else None)
        meta = {k2: torch.from_numpy(np.ascontiguousarray(sc.meta[k2][idx])) for k2 in ("center_x", "center_y", "width", "height", "R", "T", "f", "c",
                                                                                           "projection_matrix")}
        meta["scale"], meta["rot"] = torch.from_numpy(p["scale"].copy()), torch.from_numpy(p["rot"].copy())
        # label / weight are CLONES (a few KB): the views into the slot's device buffer would be overwritten when the slot comes round again four
        # batches later, silently corrupting whatever a caller kept (a validation loop collecting labels)
        return data, views["label"].clone(), views["weight"].clone(), meta

    def __iter__(self):
        order = np.arange(self.frames.n_group)
        if self.shuffle:
            self.np_rng.shuffle(order)
        for k in range(len(self)):
            yield self.batch(order[k * self.gpb:(k + 1) * self.gpb])
