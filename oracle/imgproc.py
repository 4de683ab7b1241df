"""Oracle (test infrastructure): the reference's patch generation on the host -- NumPy restatement.

Follows ``/root/reference/lib/utils/img_utils.py:114-127`` (``generate_patch_image_cv``: optional horizontal flip, ``gen_trans_from_patch_cv``,
``cv2.warpAffine(img, trans, (pw, ph), flags=cv2.INTER_LINEAR)``) and ``:265-279`` (BGR -> RGB, colour scaling, clip, mean / std).

``cv2.warpAffine`` is third-party code that is NOT installed here (conda pin ``opencv=4.1.0``, environment.yml:96); its published
algorithm (OpenCV 4.1 ``modules/imgproc/src/imgwarp.cpp``: ``cv::warpAffine`` + ``remapBilinear<FixedPtCast<int, uchar, 15>>`` +
``initInterTab2D``) is restated: the forward matrix is inverted in double, destination pixels address the source in fixed point with
``AB_BITS = 10`` / ``INTER_BITS = 5`` (``round_delta = 16``), the 2 x 2 taps are blended with int16 weights of scale ``2**15`` built from
float32 products (sum corrected to ``2**15`` on the largest / smallest entry), ``(sum + 2**14) >> 15``; ``BORDER_CONSTANT`` 0.
**Parity unpinned** for this layer: no OpenCV build is available to produce golden vectors; the tests pin the restatement by its
defining properties (identity, integer translations and flips are exact; a 2x up-scaling equals the closed-form fixed-point blend).
The NumPy arithmetic of ``:275-279`` is float32, as NumPy 1.16 (the reference's pin) evaluates float32-array-with-scalar
expressions.  Not imported by the product.
"""
import numpy as np

from . import geometry


def _itab():
    """[32][32][4] int32 weights of the bilinear table (initInterTab2D for INTER_LINEAR, scale 2**15)."""
    tab = np.zeros((32, 32, 4), np.int32)
    for ay in range(32):
        for ax in range(32):
            fx, fy = np.float32(ax) * np.float32(1.0 / 32), np.float32(ay) * np.float32(1.0 / 32)
            tx = (np.float32(1) - fx, fx)
            ty = (np.float32(1) - fy, fy)
            w = []
            for k1 in range(2):
                for k2 in range(2):
                    v = np.float32(ty[k1] * tx[k2]) * np.float32(32768.0)
                    w.append(int(np.clip(np.rint(v), -32768, 32767)))
            diff = sum(w) - 32768
            if diff:
                mk = 0
                for k in range(1, 4):
                    if (w[k] > w[mk]) if diff < 0 else (w[k] < w[mk]):
                        mk = k
                w[mk] -= diff
            tab[ay, ax] = w
    return tab


_ITAB = None


def warp_affine_linear(img, trans, dsize):
    """cv2.warpAffine(img, trans, dsize=(w, h), flags=cv2.INTER_LINEAR): uint8 [H, W, C] -> uint8 [h, w, C]."""
    global _ITAB
    if _ITAB is None:
        _ITAB = _itab()
    img = np.asarray(img, np.uint8)
    h_src, w_src, ch = img.shape
    pw, ph = int(dsize[0]), int(dsize[1])
    m = np.asarray(trans, np.float64).reshape(2, 3).copy()
    d = m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[1, 1] * d, m[0, 0] * d
    m[0, 0] = a11
    m[0, 1] *= -d
    m[1, 0] *= -d
    m[1, 1] = a22
    b1 = -m[0, 0] * m[0, 2] - m[0, 1] * m[1, 2]
    b2 = -m[1, 0] * m[0, 2] - m[1, 1] * m[1, 2]
    m[0, 2], m[1, 2] = b1, b2
    xs = np.arange(pw, dtype=np.float64)
    ys = np.arange(ph, dtype=np.float64)
    adelta = np.rint(m[0, 0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(m[1, 0] * xs * 1024.0).astype(np.int64)
    x0 = np.rint((m[0, 1] * ys + m[0, 2]) * 1024.0).astype(np.int64) + 16
    y0 = np.rint((m[1, 1] * ys + m[1, 2]) * 1024.0).astype(np.int64) + 16
    big_x = (x0[:, None] + adelta[None, :]) >> 5
    big_y = (y0[:, None] + bdelta[None, :]) >> 5
    sx, sy, ax, ay = big_x >> 5, big_y >> 5, big_x & 31, big_y & 31
    w = _ITAB[ay, ax].astype(np.int64)                       # [ph, pw, 4]
    acc = np.zeros((ph, pw, ch), np.int64)
    for k1 in range(2):
        for k2 in range(2):
            yy, xx = sy + k1, sx + k2
            ok = (yy >= 0) & (yy < h_src) & (xx >= 0) & (xx < w_src)
            px = img[np.clip(yy, 0, h_src - 1), np.clip(xx, 0, w_src - 1)].astype(np.int64)
            acc += np.where(ok[:, :, None], px, 0) * w[:, :, k1 * 2 + k2][:, :, None]
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)


def generate_patch_image(cvimg, c_x, c_y, bb_width, bb_height, patch_width, patch_height, do_flip, scale, rot):
    """img_utils.py:114-127 -> (patch uint8 [ph, pw, 3] BGR, trans 2x3 float64)."""
    img = np.asarray(cvimg).copy()
    if do_flip:
        img = img[:, ::-1, :]
        c_x = img.shape[1] - c_x - 1
    trans = geometry.gen_trans_from_patch(c_x, c_y, bb_width, bb_height, patch_width, patch_height, scale, rot, inv=False)
    return warp_affine_linear(img, trans, (int(patch_width), int(patch_height))), trans


def normalized_patch(cvimg, c_x, c_y, bb_width, bb_height, patch_width, patch_height, do_flip=False, scale=1.0, rot=0.0,
                     color_scale=(1.0, 1.0, 1.0), mean=None, std=None):
    """img_utils.py:263-279 without the occlusion augmentation: -> float32 [3, ph, pw] (RGB)."""
    patch, trans = generate_patch_image(cvimg, c_x, c_y, bb_width, bb_height, patch_width, patch_height, do_flip, scale, rot)
    out = np.transpose(patch[:, :, ::-1], (2, 0, 1)).astype(np.float32)
    for c in range(3):
        out[c] = np.clip(out[c] * np.float32(color_scale[c]), 0, 255)
        if mean is not None and std is not None:
            out[c] = (out[c] - np.float32(mean[c])) / np.float32(std[c])
    return out, trans
