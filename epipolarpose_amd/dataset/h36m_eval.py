"""H36M pose evaluation on the GPU -- mirror of ``H36M_Integral.evaluate`` (reference ``lib/dataset/h36m.py:168-378``).

MPJPE, PA-MPJPE (Procrustes-aligned), N-MPJPE (scale-normalised), their 14-joint variants and the per-axis errors are
computed by one HIP kernel (``epi_evaluate_poses``: one thread per sample, back-projection, 3x3 Jacobi SVD, alignment,
root centring) instead of a per-sample / per-joint Python loop.  ``evaluate_arrays`` works on plain arrays (there is no H36M
data on the build / GPU boxes); ``EvalMixin.evaluate`` keeps the reference's ``imdb.evaluate(preds, save_path, debug)``
contract for datasets that hold reference-format ``db`` records.
"""
import numpy as np
import torch

from .. import hip

H36M_NAMES = ['Hip', 'RHip', 'RKnee', 'RFoot', 'LHip', 'LKnee', 'LFoot', 'Spine', 'Thorax', 'Neck/Nose', 'Head',
              'LShoulder', 'LElbow', 'LWrist', 'RShoulder', 'RElbow', 'RWrist']                     # prep_h36m.py:6-23
MPII_NAMES = ['RFoot', 'RKnee', 'RHip', 'LHip', 'LKnee', 'LFoot', 'Hip', 'Thorax', 'Neck/Nose', 'Head', 'RWrist', 'RElbow',
              'RShoulder', 'LShoulder', 'LElbow', 'LWrist']                                        # prep_h36m.py:26-42
H36M_TO_MPII_PERM = np.array([H36M_NAMES.index(h) for h in MPII_NAMES])                           # prep_h36m.py:44
J14_H36M = [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15]                                          # h36m.py:186
J14_MPII = [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15]
METRIC_NAMES = ("hm36_17j      :", "hm36_17j_align:", "hm36_17j_norm:", "hm36_17j_14   :", "hm36_17j_14_al:",
                "hm36_17j_14_nm:", "hm36_17j_x    :", "hm36_17j_y    :", "hm36_17j_z    :")       # h36m.py:365-375


def evaluate_arrays(preds, gt_joints, pelvis, fl, c_p, mpii_order=False, device=None):
    """preds / gt_joints [N, J, >=3] (u, v, root-relative depth mm, image coordinates), pelvis [N, 3], fl / c_p [N, 2].
    -> (name_value list, perf = MPJPE, per-sample metrics [N, 9] ndarray, per-joint errors [N, J] ndarray)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    as_t = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), device=device)
    p = as_t(np.asarray(preds)[:, :, 0:3])
    g = as_t(np.asarray(gt_joints)[:, :, 0:3])
    per_sample, per_joint = hip.evaluate_poses(p, g, as_t(np.asarray(pelvis)[:, 2]), as_t(fl), as_t(c_p), 6 if mpii_order else 0,
                                               J14_MPII if mpii_order else J14_H36M)
    per_sample = per_sample.cpu().numpy()
    metrics = per_sample.mean(axis=0)
    return list(zip(METRIC_NAMES, metrics.tolist())), float(metrics[0]), per_sample, per_joint.cpu().numpy()


class EvalMixin:
    """``evaluate(preds, save_path=None, debug=False)`` for a dataset whose ``self.db`` holds reference-format records
    (keys ``fl``, ``c_p``, ``pelvis``, ``joints_3d``) and whose ``self.cfg.DATASET.MPII_ORDER`` selects the joint order."""

    def evaluate(self, preds, save_path=None, debug=False, actionwise=False):
        mpii = bool(self.cfg.DATASET.MPII_ORDER)
        gt = np.stack([np.asarray(r["joints_3d"], dtype=np.float64) for r in self.db])
        if mpii and gt.shape[1] == 17:
            gt = gt[:, H36M_TO_MPII_PERM, :]                       # h36m.py:219-220
        name_value, perf, _, per_joint = evaluate_arrays(
            preds, gt, np.stack([np.reshape(r["pelvis"], 3) for r in self.db]), np.stack([np.asarray(r["fl"])[0:2] for r in self.db]),
            np.stack([np.asarray(r["c_p"])[0:2] for r in self.db]), mpii_order=mpii)
        names = MPII_NAMES if mpii else H36M_NAMES
        for idx, err in enumerate(per_joint.mean(axis=0).tolist()):   # h36m.py:351-355
            print(names[idx], err)
        return name_value, perf
