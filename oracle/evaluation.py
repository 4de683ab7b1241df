"""Oracle (test infrastructure): H36M pose evaluation -- MPJPE, PA-MPJPE (Procrustes), N-MPJPE, 14-joint and per-axis errors.

float64 NumPy restatement of ``/root/reference/lib/dataset/h36m.py:168-378`` (``H36M_Integral.evaluate``) and
``lib/utils/prep_h36m.py:85-89,108-168`` (``CamBackProj``, ``compute_similarity_transform``).  Not imported by the product.
"""
import numpy as np

J14_H36M = [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 15]        # h36m.py:186 (H36M order)
J14_MPII = [0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 12, 13, 14, 15]       # h36m.py:186 (MPII order)
METRIC_NAMES = ("hm36_17j      :", "hm36_17j_align:", "hm36_17j_norm:", "hm36_17j_14   :", "hm36_17j_14_al:",
                "hm36_17j_14_nm:", "hm36_17j_x    :", "hm36_17j_y    :", "hm36_17j_z    :")      # h36m.py:365-375


def cam_back_proj(uvz, fl, c_p):
    """prep_h36m.py:85-89: (u, v, depth) -> camera coordinates."""
    uvz = np.asarray(uvz, np.float64)
    out = np.empty_like(uvz)
    out[..., 0] = (uvz[..., 0] - c_p[0]) / fl[0] * uvz[..., 2]
    out[..., 1] = (uvz[..., 1] - c_p[1]) / fl[1] * uvz[..., 2]
    out[..., 2] = uvz[..., 2]
    return out


def similarity_transform(x, y):
    """prep_h36m.py:108-168 with compute_optimal_scale=True: rotation T, scale b, translation c aligning y to x."""
    mu_x, mu_y = x.mean(0), y.mean(0)
    x0, y0 = x - mu_x, y - mu_y
    norm_x, norm_y = np.sqrt((x0 ** 2).sum()), np.sqrt((y0 ** 2).sum())
    x0, y0 = x0 / norm_x, y0 / norm_y
    u, s, vt = np.linalg.svd(x0.T @ y0, full_matrices=False)
    v = vt.T
    t = v @ u.T
    det = np.sign(np.linalg.det(t))                # :150-153 make it a rotation
    v[:, -1] *= det
    s[-1] *= det
    t = v @ u.T
    b = s.sum() * norm_x / norm_y                  # :158
    c = mu_x - b * (mu_y @ t)                      # :166
    return t, b, c


def evaluate(preds, gt_joints, pelvis, fl, c_p, mpii_order=False):
    """h36m.py:168-378.  preds / gt_joints: [N, J, >=3] (u, v, root-relative depth in mm, image coordinates);
    pelvis: [N, 3] camera-space root; fl / c_p: [N, 2].  Returns (metrics [9], per-sample metrics [N, 9],
    per-sample per-joint errors [N, J])."""
    preds = np.asarray(preds, np.float64)[:, :, 0:3]
    gt_joints = np.asarray(gt_joints, np.float64)[:, :, 0:3]
    root = 6 if mpii_order else 0                                  # :182
    j14 = J14_MPII if mpii_order else J14_H36M
    per_sample, per_joint = [], []
    for n in range(preds.shape[0]):
        p2, g2 = preds[n].copy(), gt_joints[n].copy()
        p2[:, 2] += pelvis[n][2]                                   # :222-223
        g2[:, 2] += pelvis[n][2]
        p3, g3 = cam_back_proj(p2, fl[n], c_p[n]), cam_back_proj(g2, fl[n], c_p[n])   # :231-237
        t, b, c = similarity_transform(g3, p3)                     # :240
        p_al = b * (p3 @ t) + c                                    # :241
        p_nm = b * p3                                              # :242
        p3, g3 = p3 - p3[root], g3 - g3[root]                      # :245-248
        p_al, p_nm = p_al - p_al[root], p_nm - p_nm[root]
        e = np.linalg.norm(g3 - p3, axis=1)
        e_al = np.linalg.norm(g3 - p_al, axis=1)
        e_nm = np.linalg.norm(g3 - p_nm, axis=1)
        ax = np.abs(g3 - p3)
        per_sample.append([e.mean(), e_al.mean(), e_nm.mean(), e[j14].mean(), e_al[j14].mean(), e_nm[j14].mean(),
                           ax[:, 0].mean(), ax[:, 1].mean(), ax[:, 2].mean()])
        per_joint.append(e)
    per_sample = np.asarray(per_sample)
    return per_sample.mean(axis=0), per_sample, np.asarray(per_joint)
