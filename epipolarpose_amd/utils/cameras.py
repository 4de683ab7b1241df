"""Camera model -- mirror of the reference's ``lib/utils/cameras.py:5-150`` (``Camera``).

Dataset-time host code (NumPy, float64): the training step only ever sees the arrays a ``Camera`` carries (``R T f c`` and
``projection_matrix``, collated into ``meta`` by the dataset, h36m.py:73-86).  OpenCV-backed members of the reference are
provided without OpenCV: ``project_points`` = ``project_point_radial``; ``get_fundamental_matrix`` uses the batched normalised
8-point estimate on the GPU (``epi_fundamental_8point``) where the reference runs ``cv2.findFundamentalMat(FM_LMEDS)``, whose
random sampling is not reproducible -- every match is reported as an inlier.
"""
import numpy as np


class Camera:
    def __init__(self, cam_params):
        self.cam_params = cam_params
        self.R, self.T, self.f, self.c, self.k, self.p, self.name = cam_params          # cameras.py:8
        self.R = np.asarray(self.R, np.float64).reshape(3, 3)
        self.T = np.asarray(self.T, np.float64).reshape(3, 1)
        self.f = np.asarray(self.f, np.float64).reshape(-1)[:2] if np.size(self.f) > 1 else np.array([float(np.ravel(self.f)[0])] * 2)
        self.c = np.asarray(self.c, np.float64).reshape(-1)[:2]
        self.camera_matrix = self.get_intrinsic_matrix()
        if self.k is not None or self.p is not None:
            self.dist_coeffs = self.get_dist_coeffs()
        self.tvec = self.get_tvec()
        self.projection_matrix = self.get_projection_matrix()

    def get_intrinsic_matrix(self):
        """cameras.py:120-124."""
        return np.array([[self.f[0], 0., self.c[0]], [0., self.f[1], self.c[1]], [0., 0., 1.]], dtype=np.double)

    def get_tvec(self):
        """cameras.py:149-150: t = -R T (T = camera centre in the world)."""
        return np.dot(self.R, np.negative(self.T))

    def get_projection_matrix(self):
        """cameras.py:126-131: P = K [R | -R T]."""
        return np.dot(self.get_intrinsic_matrix(), np.concatenate((self.R, self.get_tvec().reshape(3, 1)), axis=1))

    def get_disp_matrix(self):
        return np.concatenate((self.R, self.get_tvec().reshape(3, 1)), axis=1)

    def get_dist_coeffs(self):
        k = np.zeros(3) if self.k is None else np.ravel(self.k)
        p = np.zeros(2) if self.p is None else np.ravel(self.p)
        return np.array([k[0], k[1], p[0], p[1], k[2]])

    def get_essential_matrix(self, fundamental_mat):
        """cameras.py:133-134."""
        return np.dot(np.dot(self.camera_matrix.T, fundamental_mat), self.camera_matrix)

    def get_fundamental_matrix(self, u1, u2):
        """cameras.py:136-143: integer pixel coordinates as the reference casts them, the least-median-of-squares estimate
        (``cv2.FM_LMEDS``: utils/triangulation.py find_fundamental_mat_lmeds) and the inlier pairs it selects."""
        from .triangulation import find_fundamental_mat_lmeds
        u1, u2 = np.int32(u1), np.int32(u2)
        f, mask = find_fundamental_mat_lmeds(u1, u2)
        keep = mask.ravel() == 1
        return f, (u1[keep], u2[keep])

    def world_to_camera_frame(self, P):
        assert len(P.shape) == 2 and P.shape[1] == 3
        return self.R.dot(P.T - self.T).T

    def camera_to_world_frame(self, P):
        assert len(P.shape) == 2 and P.shape[1] == 3
        return (self.R.T.dot(P.T) + self.T).T

    def project_point_radial(self, P):
        """cameras.py:18-60: projection with radial and tangential distortion -> (Proj [N,2], D, radial, tan, r2)."""
        assert len(P.shape) == 2 and P.shape[1] == 3
        X = self.R.dot(P.T - self.T)
        XX = X[:2, :] / X[2, :]
        r2 = XX[0, :] ** 2 + XX[1, :] ** 2
        k = np.zeros(3) if self.k is None else np.ravel(self.k)
        p = np.zeros(2) if self.p is None else np.ravel(self.p)
        radial = 1 + k[0] * r2 + k[1] * r2 ** 2 + k[2] * r2 ** 3
        tan = p[0] * XX[1, :] + p[1] * XX[0, :]
        XXX = XX * (radial + tan) + np.outer(np.array([p[1], p[0]]), r2)
        Proj = (self.f.reshape(2, 1) * XXX + self.c.reshape(2, 1)).T
        return Proj, X[2], radial, tan, r2

    def project_points(self, P):
        return self.project_point_radial(np.asarray(P, np.float64))[0]
