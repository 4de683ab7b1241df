"""Synthetic H36M-shaped dataset with the reference's constructor, item contract and ``db`` record format.

Stands in for ``lib/dataset/h36m.py`` (no H36M images on the build / GPU boxes).  ``SyntheticH36M(cfg, root, image_set,
is_train)`` is what ``scripts/train.py:125-136`` constructs.  ``__getitem__`` returns ``(img f32[3,H,W], label f32[3J], weight
f32[3J], meta)`` exactly as ``H36M_Integral.get_data`` (h36m.py:53-88); with ``is_train and cfg.DATASET.TRI`` it returns the
two-view item ``{'cam_1': bundle, 'cam_2': bundle}`` of h36m.py:32-47 (random camera + random neighbour camera, the same frame
index in both per-camera record lists).  ``db`` holds the fields ``eval_integral`` and ``evaluate`` read (center_x, center_y,
width, height, fl, c_p, pelvis, joints_3d, joints_3d_vis); ``evaluate`` is the GPU evaluation (``h36m_eval.EvalMixin``).
"""
import random

import numpy as np
import torch
from torch.utils.data import Dataset

from ..synthetic import SyntheticScenes, project
from .h36m_eval import EvalMixin


class SyntheticH36M(EvalMixin, Dataset):
    def __init__(self, cfg, root=None, image_set="valid", is_train=False, n_group=8, n_view=None, seed=0):
        self.cfg, self.root, self.image_set, self.is_train = cfg, root, image_set, is_train
        self.patch_width, self.patch_height = int(cfg.MODEL.IMAGE_SIZE[0]), int(cfg.MODEL.IMAGE_SIZE[1])
        self.rect_3d_width = self.rect_3d_height = 2000.                      # JointIntegralDataset.py:64-65
        self.num_cams = int(n_view if n_view is not None else cfg.DATASET.NUM_CAMS)
        self.cam_config = [[1, 2], [0, 3], [0, 3], [1, 2]]                       # h36m.py:25: camera neighbourhoods
        self.parent_ids = np.array([0, 0, 1, 2, 0, 4, 5, 0, 8, 8, 9, 8, 11, 12, 8, 14, 15])     # h36m.py:23
        self.tri = bool(is_train and cfg.DATASET.TRI)
        j = int(cfg.MODEL.NUM_JOINTS)
        n_view = self.num_cams
        self.scenes = SyntheticScenes(n_group=n_group, n_view=n_view, num_joints=j, patch=256, seed=seed, augment=is_train)
        self.images = self.scenes.images(size=self.patch_width, seed=seed + 1)
        sc = self.scenes
        records = []
        for i in range(sc.batch_size):
            v, g = divmod(i, n_group)
            cam = sc.cams[v]
            uv, xc = project(sc.world[g], cam)
            joints = np.concatenate([uv, xc[:, 2:3] - xc[0, 2]], axis=1)
            records.append({"image": "synthetic/%06d.jpg" % i, "center_x": float(sc.meta["center_x"][i]),
                            "center_y": float(sc.meta["center_y"][i]), "width": float(sc.meta["width"][i]),
                            "height": float(sc.meta["height"][i]), "fl": cam["f"].copy(), "c_p": cam["c"].copy(),
                            "pelvis": xc[0].copy(), "joints_3d": joints, "joints_3d_vis": np.ones((j, 3)), "cam": cam,
                            "index": i})
        if self.tri:        # one record list per camera, the same frame (group) at the same position in each (h36m.py:100-108)
            self.db = [records[v * n_group:(v + 1) * n_group] for v in range(n_view)]
            self.db_length = n_group
        else:
            self.db = records
            self.db_length = len(records)

    def __len__(self):
        return self.db_length

    def get_data(self, the_db):
        """h36m.py:53-88: one (image patch, label, weight, meta) bundle."""
        idx, sc = the_db["index"], self.scenes
        meta = {"image": the_db["image"]}
        for k in ("center_x", "center_y", "width", "height", "scale", "rot"):
            meta[k] = float(sc.meta[k][idx])
        for k in ("R", "T", "f", "c", "projection_matrix"):
            meta[k] = sc.meta[k][idx]
        return torch.from_numpy(self.images[idx]), torch.from_numpy(sc.label[idx]), torch.from_numpy(sc.weight[idx]), meta

    def __getitem__(self, idx):
        if self.tri:                                                          # h36m.py:33-47
            cam_1 = np.random.randint(self.num_cams)
            pair = self.cam_config[cam_1 % len(self.cam_config)]
            cam_2 = (pair[0] if random.random() <= 0.5 else pair[1]) % self.num_cams
            if cam_2 == cam_1:
                cam_2 = (cam_1 + 1) % self.num_cams
            return {"cam_1": self.get_data(self.db[cam_1][idx]), "cam_2": self.get_data(self.db[cam_2][idx])}
        return self.get_data(self.db[idx])
