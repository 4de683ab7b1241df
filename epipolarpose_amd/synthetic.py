"""Seeded synthetic H36M-shaped multi-view scenes (cameras, poses, crops, labels).

There is no dataset on the build/GPU boxes, so the training hot path is exercised on synthetic
batches that honour the reference's item contract
(``lib/dataset/h36m.py:53-88``: ``(img f32[3,H,W], label f32[3J], weight f32[3J], meta)`` with
``meta`` keys ``center_x center_y width height scale rot R T f c projection_matrix``).
Geometry follows SURVEY.md section 8(d): four H36M-like cameras on a 5 m circle looking at a
subject whose pelvis sits near the world origin.  Host-side NumPy only (dataset-time code).
"""
import math

import numpy as np

AZIMUTHS = (0.3, 1.9, 3.4, 5.0)
FOCAL = (1145.0, 1145.0)
CENTER = (512.0, 512.0)
RECT_3D = 2000.0


def look_at_camera(azimuth, radius=5000.0, height=1500.0, target=(0.0, 0.0, 900.0)):
    """World->camera rotation R (rows = right, down, forward) and centre T, H36M convention X_c = R (X - T)."""
    pos = np.array([radius * math.cos(azimuth), radius * math.sin(azimuth), height])
    fwd = np.asarray(target) - pos
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.stack([right, down, fwd]), pos.reshape(3, 1)


def projection_matrix(r, t, f, c):
    """P = K [R | -R T] (the quantity ``Camera.projection_matrix`` holds, reference cameras.py:126-131)."""
    k = np.array([[f[0], 0.0, c[0]], [0.0, f[1], c[1]], [0.0, 0.0, 1.0]])
    return k @ np.concatenate([r, -r @ t.reshape(3, 1)], axis=1)


def make_cameras(n_view=4, jitter=None, rng=None):
    cams = []
    for v in range(n_view):
        az = AZIMUTHS[v % len(AZIMUTHS)] + (0.0 if jitter is None else float(rng.normal(0, jitter)))
        r, t = look_at_camera(az)
        f = np.array(FOCAL)
        c = np.array(CENTER)
        cams.append({"R": r, "T": t, "f": f, "c": c, "projection_matrix": projection_matrix(r, t, f, c)})
    return cams


def project(x_world, cam):
    xc = (x_world - cam["T"].reshape(3)) @ cam["R"].T
    uv = xc[:, :2] / xc[:, 2:3] * cam["f"] + cam["c"]
    return uv, xc


def patch_affine(c_x, c_y, bb_w, bb_h, patch_w, patch_h, scale, rot_deg, inverse=False):
    """2x3 crop affine with the reference's float32 point rounding (``gen_trans_from_patch_cv``).

    Closed form of the 3-point affine: the destination triple is axis aligned, so the map is
    ``src0 + (px - dcx)/hw * (src2 - src0) + (py - dcy)/hh * (src1 - src0)`` and its inverse.
    """
    rad = math.pi * rot_deg / 180.0
    sn, cs = math.sin(rad), math.cos(rad)
    hw = float(np.float32(bb_w * scale * 0.5))
    hh = float(np.float32(bb_h * scale * 0.5))
    s0 = np.array([c_x, c_y], dtype=np.float64).astype(np.float32).astype(np.float64)
    down = np.array([-hh * sn, hh * cs]).astype(np.float32).astype(np.float64)
    right = np.array([hw * cs, hw * sn]).astype(np.float32).astype(np.float64)
    s1 = (np.array([c_x, c_y]) + down).astype(np.float32).astype(np.float64)
    s2 = (np.array([c_x, c_y]) + right).astype(np.float32).astype(np.float64)
    dcx, dcy, dhw, dhh = patch_w * 0.5, patch_h * 0.5, patch_w * 0.5, patch_h * 0.5
    ex = (s2 - s0) / dhw            # image of the patch x axis
    ey = (s1 - s0) / dhh            # image of the patch y axis
    inv = np.array([[ex[0], ey[0], s0[0] - ex[0] * dcx - ey[0] * dcy],
                    [ex[1], ey[1], s0[1] - ex[1] * dcx - ey[1] * dcy]])
    if inverse:
        return inv
    lin = np.linalg.inv(inv[:, :2])
    return np.concatenate([lin, (-lin @ inv[:, 2]).reshape(2, 1)], axis=1)


class SyntheticScenes:
    """``G`` groups x ``V`` views.  Batch index of (view v, group g) is ``v*G + g`` (reference: first half of
    the batch is view 1, second half view 2 -- ``img_utils.py:194-199``)."""

    def __init__(self, n_group, n_view=4, num_joints=17, patch=256, seed=0, noise_px=0.0, augment=True,
                 pose_sigma=250.0):
        rng = np.random.default_rng(seed)
        self.n_group, self.n_view, self.num_joints, self.patch = n_group, n_view, num_joints, patch
        self.cams = make_cameras(n_view)
        pelvis = np.array([0.0, 0.0, 900.0]) + rng.normal(0, 100.0, size=(n_group, 1, 3))
        off = rng.normal(0, pose_sigma, size=(n_group, num_joints, 3))
        off[:, 0] = 0.0
        self.world = pelvis + off                                   # [G,J,3] mm
        b = n_group * n_view
        self.meta = {k: np.zeros((b,)) for k in ("center_x", "center_y", "width", "height", "scale", "rot")}
        self.meta.update({"R": np.zeros((b, 3, 3)), "T": np.zeros((b, 3, 1)), "f": np.zeros((b, 2)),
                          "c": np.zeros((b, 2)), "projection_matrix": np.zeros((b, 3, 4))})
        self.kps_img = np.zeros((b, num_joints, 2))                 # observed 2-D (with noise)
        self.label = np.zeros((b, 3 * num_joints), dtype=np.float32)
        self.weight = np.ones((b, 3 * num_joints), dtype=np.float32)
        for v in range(n_view):
            cam = self.cams[v]
            for g in range(n_group):
                i = v * n_group + g
                uv, xc = project(self.world[g], cam)
                zp = xc[0, 2]
                self.meta["center_x"][i] = uv[0, 0]
                self.meta["center_y"][i] = uv[0, 1]
                self.meta["width"][i] = RECT_3D * cam["f"][0] / zp        # prep_h36m.py:192-197
                self.meta["height"][i] = RECT_3D * cam["f"][1] / zp
                if augment:                                               # img_utils.py:29-39
                    self.meta["scale"][i] = np.clip(rng.normal(), -1.0, 1.0) * 0.25 + 1.0
                    self.meta["rot"][i] = np.clip(rng.normal(), -2.0, 2.0) * 30.0 if rng.random() <= 0.6 else 0.0
                else:
                    self.meta["scale"][i] = 1.0
                for k in ("R", "T", "f", "c", "projection_matrix"):
                    self.meta[k][i] = cam[k]
                self.kps_img[i] = uv + (rng.normal(0, noise_px, size=uv.shape) if noise_px > 0 else 0.0)
                fwd = patch_affine(uv[0, 0], uv[0, 1], self.meta["width"][i], self.meta["height"][i], patch, patch,
                                   self.meta["scale"][i], self.meta["rot"][i])
                pxy = uv @ fwd[:, :2].T + fwd[:, 2]
                pz = (xc[:, 2] - zp) / (RECT_3D * self.meta["scale"][i]) * patch
                lab = np.stack([pxy[:, 0] / patch - 0.5, pxy[:, 1] / patch - 0.5, pz / patch], axis=1)
                self.label[i] = lab.reshape(-1).astype(np.float32)

    @property
    def batch_size(self):
        return self.n_group * self.n_view

    def patch_coords(self, noise_px=0.0, seed=1):
        """float64 [B,J,4] patch-pixel coordinates + score (what ``get_joint_location_result`` yields) derived
        from the labels, optionally perturbed -- input for the self-supervision geometry tests."""
        rng = np.random.default_rng(seed)
        lab = self.label.astype(np.float64).reshape(self.batch_size, self.num_joints, 3)
        out = np.ones((self.batch_size, self.num_joints, 4))
        out[:, :, 0] = (lab[:, :, 0] + 0.5) * self.patch
        out[:, :, 1] = (lab[:, :, 1] + 0.5) * self.patch
        out[:, :, 2] = lab[:, :, 2] * self.patch
        if noise_px > 0:
            out[:, :, :2] += rng.normal(0, noise_px, size=(self.batch_size, self.num_joints, 2))
        return out

    def peaked_logits(self, depth, hm, gain=20.0, sigma=2.0, dtype=np.float32):
        """[B, J*D, hm, hm] logits with a Gaussian blob (x ``gain``) at each label's voxel."""
        b, j = self.batch_size, self.num_joints
        lab = self.label.reshape(b, j, 3).astype(np.float64)
        cx = (lab[:, :, 0] + 0.5) * hm
        cy = (lab[:, :, 1] + 0.5) * hm
        cz = (lab[:, :, 2] + 0.5) * depth
        ax = np.arange(hm)[None, None, :]
        az = np.arange(depth)[None, None, :]
        gx = np.exp(-0.5 * ((ax - cx[:, :, None]) / sigma) ** 2)
        gy = np.exp(-0.5 * ((ax - cy[:, :, None]) / sigma) ** 2)
        gz = np.exp(-0.5 * ((az - cz[:, :, None]) / sigma) ** 2)
        vol = gain * gz[:, :, :, None, None] * gy[:, :, None, :, None] * gx[:, :, None, None, :]
        return vol.reshape(b, j * depth, hm, hm).astype(dtype)

    def images(self, size=None, seed=2):
        size = size or self.patch
        rng = np.random.default_rng(seed)
        return rng.standard_normal((self.batch_size, 3, size, size), dtype=np.float32)
