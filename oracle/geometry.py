"""Oracle (test infrastructure): crop affine, camera model, pseudo-label re-projection, SS step.

float64 NumPy restatement of ``/root/reference/lib/utils/img_utils.py:63-111,141-243``,
``lib/utils/prep_h36m.py:170-204`` and ``lib/utils/cameras.py:120-131,149-150``.
``cv2.getAffineTransform`` (img_utils.py:101,103; OpenCV: exact affine through 3 point pairs,
solved in float64 from float32 points) is restated with ``np.linalg.solve``.
Not imported by the product.
"""
import numpy as np

from . import integral, triangulation

PATCH = 256.0          # img_utils.py:169,178 hard-coded patch size
RECT_3D = 2000.0       # img_utils.py:179,230 hard-coded depth box (mm)


def rotate_2d(pt, rot_rad):
    """img_utils.py:63-69 -- float64 arithmetic, result rounded to float32."""
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    x, y = float(pt[0]), float(pt[1])
    return np.array([x * cs - y * sn, x * sn + y * cs], dtype=np.float32)


def affine_from_3pts(src, dst):
    """cv2.getAffineTransform(src, dst): 2x3 M with M @ [sx, sy, 1] = [dx, dy] for 3 pairs (float64 solve)."""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    a = np.concatenate([src, np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, dst).T


def gen_trans_from_patch(c_x, c_y, src_width, src_height, dst_width, dst_height, scale, rot, inv=False):
    """img_utils.py:72-105.  Point triples are float32 (:93,98); box half-extents go through float32 (:82-83)."""
    src_w = float(src_width) * float(scale)
    src_h = float(src_height) * float(scale)
    center = np.array([float(c_x), float(c_y)])
    rot_rad = np.pi * float(rot) / 180.0
    down = rotate_2d(np.array([0, src_h * 0.5], dtype=np.float32), rot_rad)
    right = rotate_2d(np.array([src_w * 0.5, 0], dtype=np.float32), rot_rad)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0] = center
    src[1] = center + down
    src[2] = center + right
    dcen = np.array([dst_width * 0.5, dst_height * 0.5], dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    dst[0] = dcen
    dst[1] = dcen + np.array([0, dst_height * 0.5], dtype=np.float32)
    dst[2] = dcen + np.array([dst_width * 0.5, 0], dtype=np.float32)
    return affine_from_3pts(dst, src) if inv else affine_from_3pts(src, dst)


def trans_points2d(pts, trans):
    """img_utils.py:108-111 applied to an [N,2] array."""
    pts = np.asarray(pts, np.float64)
    return pts @ trans[:, :2].T + trans[:, 2]


def trans_coords_from_patch_to_org_3d(coords, c_x, c_y, bb_w, bb_h, patch_w, patch_h, rect_w, rect_h,
                                      scale=1.0, rot=0.0):
    """img_utils.py:141-155: inverse affine on (x,y); z_img = z_patch / patch_w * rect_3d_width; score kept."""
    coords = np.asarray(coords, np.float64)
    out = coords.copy()
    t = gen_trans_from_patch(c_x, c_y, bb_w, bb_h, patch_w, patch_h, scale, rot, inv=True)
    out[:, 0:2] = trans_points2d(coords[:, 0:2], t)
    out[:, 2] = coords[:, 2] / patch_w * rect_w
    return out


def projection_matrix(r, t, f, c):
    """cameras.py:120-131,149-150: P = K [R | -R T], float64."""
    r = np.asarray(r, np.float64).reshape(3, 3)
    t = np.asarray(t, np.float64).reshape(3, 1)
    f = np.asarray(f, np.float64).reshape(-1)
    c = np.asarray(c, np.float64).reshape(-1)
    k = np.array([[f[0], 0.0, c[0]], [0.0, f[1], c[1]], [0.0, 0.0, 1.0]])
    return k @ np.concatenate([r, r @ (-t)], axis=1)


def world_to_image_joints(keypoints, r, t, f, c, root_joint=0):
    """prep_h36m.py:177-204 (+CamProj :170-175): X_c = R (X - T); u = x/z*f + c; z -= root z.

    Returns (pt_2d [J,3] with root-relative depth, pt_3d camera coords [J,3]).  ``root_joint`` is 0
    because the reference's caller never passes ``mpii=True`` (img_utils.py:230).
    """
    x = np.asarray(keypoints, np.float64)
    r = np.asarray(r, np.float64).reshape(3, 3)
    t = np.asarray(t, np.float64).reshape(3)
    f = np.asarray(f, np.float64).reshape(-1)
    c = np.asarray(c, np.float64).reshape(-1)
    cam = (x - t) @ r.T
    pt2d = np.empty_like(cam)
    pt2d[:, 0] = cam[:, 0] / cam[:, 2] * f[0] + c[0]
    pt2d[:, 1] = cam[:, 1] / cam[:, 2] * f[1] + c[1]
    pt2d[:, 2] = cam[:, 2] - cam[root_joint, 2]
    return pt2d, cam


def labels_from_global_coords(coords_3d, meta):
    """img_utils.py:212-243 ``get_batch_labels_from_global_coords`` -> (label f32 [B,3J], weight f32 [B,3J])."""
    labels, weights = [], []
    for i in range(coords_3d.shape[0]):
        joints, _ = world_to_image_joints(coords_3d[i], meta["R"][i], meta["T"][i], meta["f"][i], meta["c"][i])
        scale = float(meta["scale"][i])
        trans = gen_trans_from_patch(meta["center_x"][i], meta["center_y"][i], meta["width"][i],
                                     meta["height"][i], PATCH, PATCH, scale, meta["rot"][i], inv=False)
        joints[:, 0:2] = trans_points2d(joints[:, 0:2], trans)                 # :235
        joints[:, 2] = joints[:, 2] / (RECT_3D * scale) * PATCH               # :236
        lab, w = integral.generate_joint_location_label(PATCH, PATCH, joints, np.ones_like(joints))
        labels.append(lab)
        weights.append(w)
    return np.asarray(labels, dtype=np.float32), np.asarray(weights, dtype=np.float32)


def decode_to_image(coords_patch, meta):
    """img_utils.py:171-185: per sample patch -> original image coordinates, depth to mm."""
    out = []
    for n in range(coords_patch.shape[0]):
        out.append(trans_coords_from_patch_to_org_3d(
            coords_patch[n], meta["center_x"][n], meta["center_y"][n], meta["width"][n], meta["height"][n],
            PATCH, PATCH, RECT_3D, RECT_3D, scale=meta["scale"][n], rot=meta["rot"][n]))
    return np.asarray(out)


def self_supervision(logits, meta, n_view=2, method="iterative", num_joints=None, coords_patch=None):
    """img_utils.py:166-190: decode -> patch-to-image -> triangulate -> re-project into every view.

    ``coords_patch`` (float64 [B,J,4]) may be supplied instead of logits (to isolate the geometry).
    """
    if coords_patch is None:
        coords_patch = integral.get_joint_location_result(PATCH, PATCH, logits, num_joints=num_joints)
    kps_img = decode_to_image(coords_patch, meta)
    x_world = triangulation.triangulate_pairs(kps_img, meta["projection_matrix"], n_view=n_view, method=method)
    label, weight = labels_from_global_coords(x_world, meta)
    return label, weight, x_world, kps_img
