"""Synthetic H36M-shaped dataset with the reference's item contract and ``db`` record format.

Stands in for ``lib/dataset/h36m.py`` (no H36M images on the build / GPU boxes): ``__getitem__`` returns
``(img f32[3,H,W], label f32[3J], weight f32[3J], meta)`` exactly as ``H36M_Integral.get_data`` (h36m.py:53-88); ``db`` holds
the fields ``eval_integral`` and ``evaluate`` read (center_x, center_y, width, height, fl, c_p, pelvis, joints_3d,
joints_3d_vis); ``evaluate`` is the GPU evaluation (``h36m_eval.EvalMixin``).
"""
import numpy as np
import torch
from torch.utils.data import Dataset

from ..synthetic import SyntheticScenes, project
from .h36m_eval import EvalMixin


class SyntheticH36M(EvalMixin, Dataset):
    def __init__(self, cfg, root=None, image_set="valid", is_train=False, n_group=8, n_view=4, seed=0):
        self.cfg, self.root, self.image_set, self.is_train = cfg, root, image_set, is_train
        self.patch_width, self.patch_height = int(cfg.MODEL.IMAGE_SIZE[0]), int(cfg.MODEL.IMAGE_SIZE[1])
        j = int(cfg.MODEL.NUM_JOINTS)
        self.scenes = SyntheticScenes(n_group=n_group, n_view=n_view, num_joints=j, patch=256, seed=seed, augment=is_train)
        self.images = self.scenes.images(size=self.patch_width, seed=seed + 1)
        sc = self.scenes
        self.db = []
        for i in range(sc.batch_size):
            v, g = divmod(i, n_group)
            cam = sc.cams[v]
            uv, xc = project(sc.world[g], cam)
            joints = np.concatenate([uv, xc[:, 2:3] - xc[0, 2]], axis=1)
            self.db.append({"image": "synthetic/%06d.jpg" % i, "center_x": float(sc.meta["center_x"][i]),
                            "center_y": float(sc.meta["center_y"][i]), "width": float(sc.meta["width"][i]),
                            "height": float(sc.meta["height"][i]), "fl": cam["f"].copy(), "c_p": cam["c"].copy(),
                            "pelvis": xc[0].copy(), "joints_3d": joints, "joints_3d_vis": np.ones((j, 3)), "cam": cam})
        self.db_length = len(self.db)

    def __len__(self):
        return self.db_length

    def __getitem__(self, idx):
        sc = self.scenes
        meta = {"image": self.db[idx]["image"]}
        for k in ("center_x", "center_y", "width", "height", "scale", "rot"):
            meta[k] = float(sc.meta[k][idx])
        for k in ("R", "T", "f", "c", "projection_matrix"):
            meta[k] = sc.meta[k][idx]
        return torch.from_numpy(self.images[idx]), torch.from_numpy(sc.label[idx]), torch.from_numpy(sc.weight[idx]), meta
