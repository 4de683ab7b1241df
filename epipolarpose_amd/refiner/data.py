"""Dataset of the refinement MLP -- the reference's ``refiner/data.py`` reads pickled (noisy 3-D pose, ground truth) pairs of
Human3.6M; no such data exists on the build / GPU boxes, so ``SyntheticLift`` supplies seeded pairs with the same item contract
(``(inp f32 [45], out f32 [45])``, hip joint removed, inputs always and training targets standardised, data.py:40-70) and the same
``evaluate`` protocol (MPJPE and Procrustes-aligned MPJPE over the 15 joints, data.py:78-158).  The reference's pickle-reading ``Human36M`` class is
data plumbing outside the hot-path scope and is not mirrored."""
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
