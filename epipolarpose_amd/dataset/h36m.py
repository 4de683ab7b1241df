"""The reference's on-disk Human3.6M format: ``<root>/annot/<image_set>.pkl`` + JPEG frames -- ``lib/dataset/h36m.py:91-165``.

* ``read_annotations`` / ``build_db``: the annotation pickle in either of its two forms -- per camera ``{1: [records], ..., 4: [records]}`` or flat
  ``[records]`` -- and the record list(s) ``H36M_Integral._get_train_db`` / ``_get_val_db`` make of it, drawn from the ``numpy.random`` / ``random`` module
  states in the reference's order (so the same seeds give the same ``db``).  A record's ``cam`` is a pickled ``lib.utils.cameras.Camera``:
  ``install_as_lib()`` resolves that class path to ``utils.cameras.Camera`` (unpickling restores attributes, it never calls ``__init__``).
* ``H36MFrames``: every frame decoded ONCE (PIL -> BGR, what ``cv2.imread`` yields) and kept in HBM as uint8, with the per-sample arrays the GPU
  input pipeline reads -- the attribute contract of ``synthetic_frames.SyntheticFrames``, so ``FramePatchLoader`` drives real files unchanged
  (one crop / occlusion / normalise launch per multi-view batch instead of one ``cv2.warpAffine`` per sample on a CPU worker).
* ``H36M_Integral``: the reference's dataset class -- constructor ``(cfg, root, image_set, is_train)``, ``db``, ``__len__``, ``__getitem__`` with the
  single bundle ``(img f32[3,H,W], label f32[3J], weight f32[3J], meta)`` or, with ``is_train and cfg.DATASET.TRI``, ``{'cam_1': bundle, 'cam_2':
  bundle}`` (h36m.py:32-50), ``get_data`` (h36m.py:53-88), ``evaluate`` (GPU, ``h36m_eval.EvalMixin``) -- for code that iterates the data set
  item by item; the image patch of an item comes from the same device crop kernel (a batch of one).
Checked against the LIVE reference class on the same files: tests/golden/make_h36m_fixture.py -> tests/golden/h36m_files.npz, tests/test_h36m_files.py.
JPEG decoding is PIL's libjpeg, the reference's is OpenCV's: both IJG-compatible decoders, not guaranteed bit-identical to each other (unpinned, as every
OpenCV primitive here: DESIGN section 5).
"""
import collections
import copy
import os
import pickle
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ..utils.img_utils import IMAGENET_MEAN, IMAGENET_STD, do_augmentation, patch_affines_batch, patch_labels_batch
from .h36m_eval import EvalMixin

RECT_3D = 2000.0                                             # JointIntegralDataset.py:64-65


def read_annotations(root, image_set):
    """``pkl.load(<root>/annot/<image_set>.pkl)`` (h36m.py:93-98) with the reference's module paths resolvable."""
    from .. import install_as_lib
    install_as_lib()
    with open(os.path.join(root, "annot", image_set + ".pkl"), "rb") as f:
        return pickle.load(f)


def build_db(anno, num_cams, tri, is_train, np_rng=np.random, py_rng=random):
    """h36m.py:91-165.  -> (db, db_length).  Per-camera annotations: a random permutation of the frames (``np.random.permutation``), the same for
    every camera; training with ``tri`` keeps one list per camera, otherwise the lists are concatenated (and shuffled with ``random.shuffle`` when
    training).  Flat annotations: the list as stored, shuffled when training."""
    if isinstance(anno, dict):
        per_cam = [[] for _ in range(num_cams)]
        for idx in np_rng.permutation(len(anno[1])):
            for cid in range(num_cams):
                per_cam[cid].append(anno[cid + 1][idx])
        if is_train and tri:
            return per_cam, len(per_cam[0])
        db = [rec for cam in per_cam for rec in cam]
        if is_train:
            py_rng.shuffle(db)
        return db, len(db)
    db = [anno[i] for i in range(len(anno))]
    if is_train:
        py_rng.shuffle(db)
    return db, len(db)


def decode_bgr(path):
    """``cv2.imread(path, IMREAD_COLOR | IMREAD_IGNORE_ORIENTATION)`` (img_utils.py:251-252): uint8 [H, W, 3] in BGR order, EXIF orientation ignored."""
    from PIL import Image
    try:
        with Image.open(path) as im:
            rgb = np.asarray(im.convert("RGB"))
    except (OSError, ValueError):
        raise IOError("Fail to read %s" % path)                # img_utils.py:254-255
    return np.ascontiguousarray(rgb[:, :, ::-1])


def record_meta(rec, root):
    """The camera / crop entries of ``meta`` (h36m.py:72-86) of one record."""
    cam = rec["cam"]
    return {"image": os.path.join(root, rec["image"]), "center_x": rec["center_x"], "center_y": rec["center_y"], "width": rec["width"],
            "height": rec["height"], "R": np.asarray(cam.R, np.float64), "T": np.asarray(cam.T, np.float64).reshape(3, 1),
            "f": np.asarray(cam.f, np.float64).reshape(-1), "c": np.asarray(cam.c, np.float64).reshape(-1),
            "projection_matrix": np.asarray(cam.projection_matrix, np.float64)}


class _Meta:
    """The ``scenes`` attribute ``FramePatchLoader`` reads: ``meta`` arrays over the frame store's samples."""

    def __init__(self, meta):
        self.meta = meta


class FrameCache:
    """Decoded frames in HBM, least recently used out first, under a byte budget (a ~3 MB frame per record: the full H36M training set is hundreds of
    thousands of them -- an unbounded dict would end in an out-of-memory error after a few hundred thousand items)."""

    def __init__(self, max_bytes=8 << 30):
        self.max_bytes, self.bytes = int(max_bytes), 0
        self._items = collections.OrderedDict()

    def get(self, path, device):
        hit = self._items.get(path)
        if hit is not None:
            self._items.move_to_end(path)
            return hit
        bgr = decode_bgr(path)
        hit = (torch.from_numpy(bgr.reshape(-1)).to(device), bgr.shape[:2])
        self._items[path] = hit
        self.bytes += hit[0].numel()
        while self.bytes > self.max_bytes and len(self._items) > 1:
            _, (old, _) = self._items.popitem(last=False)
            self.bytes -= old.numel()
        return hit

    def __len__(self):
        return len(self._items)


class H36MFrames:
    """Frames of a per-camera ``db`` (``n_view`` lists of ``n_group`` records, the same frame at the same position in each) decoded into HBM.

    Sample index of (view v, group g) = ``v * n_group + g`` (img_utils.py:194-199).  Attributes: ``frames`` (uint8 CUDA, the BGR frames back to back),
    ``frame_offset_host`` int64 [B], ``frame_hw_host`` int32 [B, 2], ``joints`` [B, J, 3] (u, v, root-relative depth in mm: ``db['joints_3d']``),
    ``joints_vis`` [B, J, 3] (z column scaled by ``z_weight``, h36m.py:61-62), ``scenes.meta`` (center_x .. projection_matrix)."""

    def __init__(self, per_cam_db, root, z_weight=1.0, device=None, groups=None, max_bytes=64 << 30):
        """groups: a range / slice of frame positions to decode (a WINDOW of the data set: the loader is then rebuilt per window); max_bytes: refuse to
        decode more than this into HBM (the whole H36M training set does not fit any budget: ~3 MB x 4 cameras x hundreds of thousands of frames)."""
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        if groups is not None:
            per_cam_db = [list(cam[groups] if isinstance(groups, slice) else [cam[i] for i in groups]) for cam in per_cam_db]
        self.n_view, self.n_group = len(per_cam_db), len(per_cam_db[0])
        assert all(len(c) == self.n_group for c in per_cam_db), "every camera must list the same frames"
        est = self.n_view * self.n_group * 3 * 1000 * 1000             # H36M frames are ~1000 x 1000 x 3 bytes
        if est > max_bytes:
            raise ValueError("H36MFrames: %d frames (~%.0f GiB decoded) exceed the %.0f GiB budget -- decode a window (groups=range(a, b)) per pass "
                             "or raise max_bytes" % (self.n_view * self.n_group, est / 2.0 ** 30, max_bytes / 2.0 ** 30))
        recs = [rec for cam in per_cam_db for rec in cam]
        b = len(recs)
        self.records = recs
        self.num_joints = len(recs[0]["joints_3d"])
        frames = [decode_bgr(os.path.join(root, r["image"])) for r in recs]
        self.frame_hw_host = np.array([[f.shape[0], f.shape[1]] for f in frames], np.int32)
        sizes = np.array([f.size for f in frames], np.int64)
        self.frame_offset_host = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        self.frames = torch.from_numpy(np.concatenate([f.reshape(-1) for f in frames])).to(self.device)
        self.joints = np.stack([np.asarray(r["joints_3d"], np.float64) for r in recs])
        self.joints_vis = np.stack([np.asarray(r["joints_3d_vis"], np.float64) for r in recs]).copy()
        self.joints_vis[:, :, 2] *= z_weight
        metas = [record_meta(r, root) for r in recs]
        meta = {k: np.array([float(m[k]) for m in metas]) for k in ("center_x", "center_y", "width", "height")}
        meta.update({k: np.stack([m[k] for m in metas]) for k in ("R", "T", "f", "c", "projection_matrix")})
        meta["scale"], meta["rot"] = np.ones(b), np.zeros(b)
        self.scenes = _Meta(meta)

    def index(self, view, group):
        return view * self.n_group + group


class H36M_Integral(EvalMixin, Dataset):
    """lib/dataset/h36m.py:19-88 on the GPU input pipeline (see the module docstring)."""
    items_use_device = True          # __getitem__ launches the crop kernel: iterate in-process (scripts/train.py gives such a data set num_workers = 0)

    def __init__(self, cfg, root, image_set, is_train, device=None):
        self.cfg, self.root, self.image_set, self.is_train = cfg, root, image_set, is_train
        self.patch_width, self.patch_height = int(cfg.MODEL.IMAGE_SIZE[0]), int(cfg.MODEL.IMAGE_SIZE[1])
        self.rect_3d_width = self.rect_3d_height = RECT_3D
        self.mean, self.std = np.array(IMAGENET_MEAN), np.array(IMAGENET_STD)          # JointIntegralDataset.py:67-68
        self.num_cams = int(cfg.DATASET.NUM_CAMS)
        self.parent_ids = np.array([0, 0, 1, 2, 0, 4, 5, 0, 8, 8, 9, 8, 11, 12, 8, 14, 15])     # h36m.py:23
        self.cam_config = [[1, 2], [0, 3], [0, 3], [1, 2]]                                        # h36m.py:25: camera neighbourhoods
        self.tri = bool(is_train and cfg.DATASET.TRI)
        self.occluders = None
        if cfg.DATASET.OCCLUSION and is_train:                                                    # JointIntegralDataset.py:73
            from ..utils.augmentation import load_occluders
            self.occluders = load_occluders(cfg.DATASET.VOC)
        self.db, self.db_length = build_db(read_annotations(root, image_set), self.num_cams, bool(cfg.DATASET.TRI), is_train)
        self._device = device
        self._frame_cache = FrameCache(int(getattr(cfg.DATASET, "FRAME_CACHE_BYTES", 8 << 30)))

    def __len__(self):
        return self.db_length

    def _frame(self, path):
        """(uint8 CUDA tensor of the decoded frame, (h, w)), decoded and uploaded at first use."""
        return self._frame_cache.get(path, self._device or torch.device("cuda", torch.cuda.current_device()))

    def host_sample(self, the_db):
        """The host side of ``get_data``: the augmentation draw (img_utils.py:257-261, from the module RNG states like the reference), the crop affine,
        labels and weights (img_utils.py:281-296), ``meta`` (h36m.py:72-86).  -> dict(scale, rot, color, trans [1, 2, 3], label, weight, meta)."""
        meta = record_meta(the_db, self.root)
        joints_vis = np.asarray(the_db["joints_3d_vis"], np.float64).copy()
        joints_vis[:, 2] *= self.cfg.DATASET.Z_WEIGHT                                             # h36m.py:61-62
        scale, rot, _, color = do_augmentation() if self.is_train else (1.0, 0, False, [1.0, 1.0, 1.0])
        trans = patch_affines_batch([meta["center_x"]], [meta["center_y"]], [meta["width"]], [meta["height"]], self.patch_width, self.patch_height,
                                    [scale], [rot])
        label, weight = patch_labels_batch(np.asarray(the_db["joints_3d"], np.float64)[None], joints_vis[None], trans, [meta["width"]], [scale],
                                           self.patch_width, self.patch_height, self.rect_3d_width)
        meta["scale"], meta["rot"] = float(scale), float(rot)
        return {"scale": scale, "rot": rot, "color": color, "trans": trans, "label": label[0], "weight": weight[0], "meta": meta}

    def get_data(self, the_db):
        """h36m.py:53-88 -> (img_patch f32 [3, H, W], label f32 [3J], label_weight f32 [3J], meta)."""
        from .. import hip
        hs = self.host_sample(the_db)
        meta = hs["meta"]
        frame, (fh, fw) = self._frame(meta["image"])
        dev = frame.device
        place = bank = None
        if self.occluders:
            from ..utils import augmentation as aug
            if getattr(self, "_bank", None) is None:
                self._bank = aug.OccluderBank(self.occluders, dev)
            bank = self._bank.tensors()
            place = torch.from_numpy(aug.draw_occlusion((self.patch_height, self.patch_width), self._bank.hw_host)[None]).to(dev)
        img = hip.crop_patches(frame, torch.zeros(1, dtype=torch.int64, device=dev), torch.tensor([[fh, fw]], dtype=torch.int32, device=dev),
                               torch.from_numpy(hs["trans"]).to(dev), self.patch_height, self.patch_width,
                               color_scale=torch.tensor([hs["color"]], dtype=torch.float32, device=dev), mean=IMAGENET_MEAN, std=IMAGENET_STD,
                               occluders=bank, placements=place)
        return img[0].float().cpu().numpy(), hs["label"], hs["weight"], meta

    def __getitem__(self, idx):
        if self.tri:                                                                              # h36m.py:33-47
            cam_1 = np.random.randint(self.num_cams)
            cam_2 = self.cam_config[cam_1][0] if random.random() <= 0.5 else self.cam_config[cam_1][1]
            rec_1, rec_2 = copy.deepcopy(self.db[cam_1][idx]), copy.deepcopy(self.db[cam_2][idx])
            return {"cam_1": self.get_data(rec_1), "cam_2": self.get_data(rec_2)}
        return self.get_data(copy.deepcopy(self.db[idx]))

    def frame_store(self, device=None, groups=None, max_bytes=64 << 30):
        """The data set -- or the window ``groups`` of its frame positions -- decoded into HBM for ``synthetic_frames.FramePatchLoader`` (needs the
        per-camera record lists: ``DATASET.TRI`` training).  More than ``max_bytes`` of decoded frames is refused (H36MFrames)."""
        if not self.tri:
            raise ValueError("frame_store needs the per-camera db (is_train and DATASET.TRI)")
        return H36MFrames(self.db, self.root, z_weight=self.cfg.DATASET.Z_WEIGHT, device=device or self._device, groups=groups, max_bytes=max_bytes)


def h36m(cfg, root=None, image_set="valid", is_train=False, **kwargs):
    """``lib.dataset.h36m`` (lib/dataset/__init__.py:11): the reader above where ``<root>/annot/<image_set>.pkl`` exists, the synthetic stand-in with the
    same constructor and item contract where it does not (no H36M on the build / GPU boxes)."""
    if root and os.path.isfile(os.path.join(str(root), "annot", str(image_set) + ".pkl")):
        return H36M_Integral(cfg, root, image_set, is_train, **{k: v for k, v in kwargs.items() if k == "device"})
    from .synthetic import SyntheticH36M
    return SyntheticH36M(cfg, root, image_set, is_train, **{k: v for k, v in kwargs.items() if k != "device"})
