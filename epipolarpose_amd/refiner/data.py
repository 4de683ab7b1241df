"""Dataset of the refinement MLP (reference ``refiner/data.py``).

* ``Human36M(is_train)`` -- the class the reference's ``refiner/main.py:13`` imports: reads the reference's pickles ``refiner/data/{train,valid}.pkl``
  (``{'inp': ..., 'out': ...}`` pose pairs, hip joint removed, inputs always and training targets standardised, ``norm.pkl`` written by the training split:
  data.py:40-70) where they exist, and falls back to seeded synthetic pairs with the same item contract where they do not (no such data on the build /
  GPU boxes).  ``evaluate`` = MPJPE and Procrustes-aligned MPJPE over the 15 joints (data.py:78-158).
* ``SyntheticLift`` -- the seeded stand-in itself."""
import os
import pickle

import numpy as np
from torch.utils.data import Dataset

from ..utils.prep_h36m import compute_similarity_transform


class SyntheticLift(Dataset):
    def __init__(self, is_train, n=512, joints=15, noise_mm=40.0, seed=0, norm=None):
        rng = np.random.default_rng(seed + (0 if is_train else 1))
        self.is_train = is_train
        labels = rng.normal(0, 250.0, size=(n, joints * 3)).astype(np.float32)
        data = labels + rng.normal(0, noise_mm, size=labels.shape).astype(np.float32)
        if norm is None:
            norm = (data.mean(0), data.std(0), labels.mean(0), labels.std(0))
        self.data_mean, self.data_std, self.labels_mean, self.labels_std = norm
        self.data = (data - self.data_mean) / self.data_std
        self.labels = (labels - self.labels_mean) / self.labels_std if is_train else labels

    def norm(self):
        return self.data_mean, self.data_std, self.labels_mean, self.labels_std

    def __getitem__(self, index):
        return self.data[index], self.labels[index]

    def __len__(self):
        return len(self.labels)

    def evaluate(self, preds):
        """data.py:78-158: de-standardise, per-sample mean joint error and its Procrustes-aligned variant."""
        preds = (preds * self.labels_std + self.labels_mean).reshape(preds.shape[0], -1, 3)
        gt = self.labels.reshape(self.labels.shape[0], -1, 3)
        dist, dist_align = [], []
        for p, g in zip(preds, gt):
            _, _, t, b, c = compute_similarity_transform(g, p, compute_optimal_scale=True)
            dist.append(np.linalg.norm(g - p, axis=1).mean())
            dist_align.append(np.linalg.norm(g - (b * p.dot(t) + c), axis=1).mean())
        return float(np.mean(dist)), float(np.mean(dist_align))


class Human36M(SyntheticLift):
    """refiner/data.py:31-76 (see the module docstring).  ``root``: where ``refiner/data/*.pkl`` live (the reference uses the working directory)."""

    def __init__(self, is_train, root="."):
        fname = os.path.join(root, "refiner", "data", "train.pkl" if is_train else "valid.pkl")
        if not os.path.isfile(fname):
            super().__init__(is_train)
            return
        self.is_train = is_train
        with open(fname, "rb") as f:
            anno = pickle.load(f)
        data = np.asarray(anno["inp"], dtype=np.float32).reshape(len(anno["inp"]), -1)
        labels = np.asarray(anno["out"], dtype=np.float32).reshape(len(anno["out"]), -1)
        data, labels = np.delete(data, np.s_[18:21], axis=1), np.delete(labels, np.s_[18:21], axis=1)          # remove the hip joint (:50-52)
        norm_file = os.path.join(root, "refiner", "data", "norm.pkl")
        if os.path.exists(norm_file):
            with open(norm_file, "rb") as f:
                norm = pickle.load(f)
        elif is_train:
            norm = (data.mean(axis=0), data.std(axis=0), labels.mean(axis=0), labels.std(axis=0))
            with open(norm_file, "wb") as f:
                pickle.dump(norm, f)
        else:
            raise IOError("refiner/data/norm.pkl is missing: construct the training split first (refiner/data.py:54-62)")
        self.data_mean, self.data_std, self.labels_mean, self.labels_std = norm
        data = (data - self.data_mean) / self.data_std
        if is_train:
            labels = (labels - self.labels_mean) / self.labels_std
            rnd = np.random.permutation(labels.shape[0])                                                     # :69-71
            data, labels = data[rnd], labels[rnd]
        self.data, self.labels = data, labels
