"""Camera-frame helpers -- mirror of the reference's ``lib/utils/prep_h36m.py`` (the functions the hot path and the evaluation
use: ``CamProj`` / ``CamBackProj`` :85-89,170-175, ``from_worldjt_to_imagejt`` :177-204, ``compute_similarity_transform``
:108-168).  Dataset-time host code (NumPy); the training step uses the batched device versions (``epi_reproject_labels``,
``epi_evaluate_poses``).
"""
import numpy as np

H36M_NAMES = ['Hip', 'RHip', 'RKnee', 'RFoot', 'LHip', 'LKnee', 'LFoot', 'Spine', 'Thorax', 'Neck/Nose', 'Head',
              'LShoulder', 'LElbow', 'LWrist', 'RShoulder', 'RElbow', 'RWrist']                      # prep_h36m.py:6-23
MPII_NAMES = ['RFoot', 'RKnee', 'RHip', 'LHip', 'LKnee', 'LFoot', 'Hip', 'Thorax', 'Neck/Nose', 'Head', 'RWrist', 'RElbow',
              'RShoulder', 'LShoulder', 'LElbow', 'LWrist']                                         # prep_h36m.py:26-42
H36M_TO_MPII_PERM = np.array([H36M_NAMES.index(h) for h in MPII_NAMES])                            # prep_h36m.py:44


def CamProj(x, y, z, fx, fy, u, v):
    """prep_h36m.py:170-175: pinhole projection without distortion."""
    return x / z * fx + u, y / z * fy + v


def CamBackProj(cam_x, cam_y, depth, fx, fy, u, v):
    """prep_h36m.py:85-89."""
    return (cam_x - u) / fx * depth, (cam_y - v) / fy * depth, depth


def from_worldjt_to_imagejt(joint_num, rot, keypoints, trans, fl, c_p, rect_3d_width, rect_3d_height, mpii=False):
    """prep_h36m.py:177-204: X_c = R (X - T), projection, root-relative depth, and the image rectangle of the
    ``rect_3d`` box centred on the root joint.  -> (l, r, t, b, pt_2d [J,3], pt_3d [J,3], vis [J,3], pelvis3d)."""
    root_joint = 6 if mpii else 0
    rot = np.asarray(rot, np.float64).reshape(3, 3)
    pt_3d = (np.asarray(keypoints, np.float64)[:joint_num] - np.asarray(trans, np.float64).reshape(3)) @ rot.T
    pt_2d = np.zeros((joint_num, 3), dtype=np.float64)
    pt_2d[:, 0], pt_2d[:, 1] = CamProj(pt_3d[:, 0], pt_3d[:, 1], pt_3d[:, 2], fl[0], fl[1], c_p[0], c_p[1])
    pelvis3d = pt_3d[root_joint].copy()
    half = np.array([rect_3d_width / 2, rect_3d_height / 2, 0])
    lt, rb = pelvis3d - half, pelvis3d + half
    rect2d_l, rect2d_t = CamProj(lt[0], lt[1], lt[2], fl[0], fl[1], c_p[0], c_p[1])
    rect2d_r, rect2d_b = CamProj(rb[0], rb[1], rb[2], fl[0], fl[1], c_p[0], c_p[1])
    pt_2d[:, 2] = pt_3d[:, 2] - pelvis3d[2]
    return rect2d_l, rect2d_r, rect2d_t, rect2d_b, pt_2d, pt_3d, np.ones((joint_num, 3), dtype=np.float64), pelvis3d


def compute_similarity_transform(X, Y, compute_optimal_scale=False):
    """prep_h36m.py:108-168 (Procrustes): the similarity (scale b, rotation T, translation c) that maps Y onto X.
    -> (d, Z, T, b, c) with Z = b * Y @ T + c the aligned points and d the normalised residual."""
    X, Y = np.asarray(X, np.float64), np.asarray(Y, np.float64)
    muX, muY = X.mean(0), Y.mean(0)
    X0, Y0 = X - muX, Y - muY
    ssX, ssY = (X0 ** 2.).sum(), (Y0 ** 2.).sum()
    normX, normY = np.sqrt(ssX), np.sqrt(ssY)
    X0, Y0 = X0 / normX, Y0 / normY
    U, s, Vt = np.linalg.svd(np.dot(X0.T, Y0), full_matrices=False)
    V = Vt.T
    T = np.dot(V, U.T)
    if np.linalg.det(T) < 0:            # no reflections
        V[:, -1] *= -1
        s[-1] *= -1
        T = np.dot(V, U.T)
    traceTA = s.sum()
    if compute_optimal_scale:
        b = traceTA * normX / normY
        d = 1 - traceTA ** 2
        Z = normX * traceTA * np.dot(Y0, T) + muX
    else:
        b = 1
        d = 1 + ssY / ssX - 2 * traceTA * normY / normX
        Z = normY * np.dot(Y0, T) + muX
    c = muX - b * np.dot(muY, T)
    return d, Z, T, b, c
