"""Synthetic-occlusion augmentation -- mirror of the reference's ``lib/utils/augmentation.py``.

The reference pastes segmented Pascal-VOC objects over the person patch (``load_occluders`` :9-58, ``occlude_with_objects`` :61-81,
``paste_over`` :84-114, ``resize_by_factor`` :117-123).  ``load_occluders(root)`` reads a Pascal-VOC tree in the reference's on-disk format
(``Annotations/*.xml``, ``JPEGImages``, ``SegmentationObject``; decoding is PIL's as in the reference, the two OpenCV calls -- the 8 x 8 elliptic
erosion and the INTER_AREA halving -- are restated).  There is no Pascal VOC on the build / GPU boxes, so without a tree it falls back
to procedural RGBA occluders with the reference's alpha convention (255 inside the object, 192 on the ring its 8 x 8 erosion removes,
0 outside) and the same half-size down-scaling; everything downstream is the reference's pipeline: the random draws are made on the host
in the reference's order (``draw_occlusion``), the resize (box-filter average = ``cv2.resize(INTER_AREA)`` restated) and the float32
alpha blend with uint8 truncation run inside the batched crop kernel (``epi_crop_patches_occluded``, csrc/patch_crop.hip).
"""
import os
import random

import numpy as np
import torch

MAX_OCCLUDERS = 7            # occlude_with_objects draws count = randint(1, 8)


def _procedural_occluder(rng):
    """One RGBA object: a filled super-ellipse with a textured colour; alpha 255 inside, 192 on the border ring (load_occluders :44-49
    erodes the mask with an 8 x 8 ellipse and sets the removed ring to 192), 0 outside; then halved like ``resize_by_factor(.., 0.5)`` (:52)."""
    h, w = int(rng.integers(80, 260)), int(rng.integers(80, 260))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    p = rng.uniform(1.5, 4.0)
    r = (np.abs((xx - (w - 1) / 2) / (w / 2)) ** p + np.abs((yy - (h - 1) / 2) / (h / 2)) ** p) ** (1.0 / p)
    inside = r <= 1.0
    ring = inside & (r > 1.0 - 8.0 / min(h, w))
    base = rng.integers(30, 226, size=3).astype(np.float64)
    stripes = 25.0 * np.sin(xx * rng.uniform(0.05, 0.4) + yy * rng.uniform(0.05, 0.4) + rng.uniform(0, 6.28))
    noise = rng.normal(0, 8.0, size=(h, w))
    rgb = np.clip(base[None, None, :] + (stripes + noise)[:, :, None], 0, 255)
    alpha = np.where(ring, 192, np.where(inside, 255, 0))
    img = np.concatenate([rgb, alpha[:, :, None]], axis=2).astype(np.uint8)
    # half size, as the reference stores its occluders: 2 x 2 box average (rounded half up)
    h2, w2 = h // 2, w // 2
    q = img[:2 * h2, :2 * w2].reshape(h2, 2, w2, 2, 4).astype(np.int32).sum(axis=(1, 3))
    return ((q + 2) // 4).astype(np.uint8)


def _ellipse_element(width, height):
    """The ones of ``cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (width, height))`` as (row offset, column offset) pairs from its anchor, the
    element's centre (augmentation.py:11; OpenCV: row i spans [c - dx, c + dx], dx = round(c * sqrt(1 - (i - r)^2 / r^2)), r = height // 2,
    c = width // 2)."""
    r, c = height // 2, width // 2
    taps = []
    for i in range(height):
        dy = i - r
        if abs(dy) > r:
            continue
        dx = int(np.rint(c * np.sqrt((r * r - dy * dy) / float(r * r)))) if r else 0
        taps += [(dy, j - c) for j in range(max(c - dx, 0), min(c + dx + 1, width))]
    return taps


def _erode(mask, taps):
    """``cv2.erode`` of a uint8 [h, w] mask by the element ``taps``: the minimum over the element, pixels outside the image not taking part."""
    h, w = mask.shape
    reach = max(max(abs(a), abs(b)) for a, b in taps)
    framed = np.full((h + 2 * reach, w + 2 * reach), 255, np.uint8)
    framed[reach:reach + h, reach:reach + w] = mask
    out = mask.copy()
    for dy, dx in taps:
        np.minimum(out, framed[reach + dy:reach + dy + h, reach + dx:reach + dx + w], out=out)
    return out


def _cover(src, dst):
    """[dst, src] int64: how many 1/dst-ths of source pixel k fall under destination pixel i when ``src`` pixels shrink onto ``dst`` (rows sum to src)."""
    edges = np.arange(dst + 1, dtype=np.int64) * src                    # destination pixel i covers [i*src, (i+1)*src) in units of 1/dst source pixel
    lo = np.maximum(edges[:-1, None], np.arange(src, dtype=np.int64)[None, :] * dst)
    hi = np.minimum(edges[1:, None], (np.arange(src, dtype=np.int64)[None, :] + 1) * dst)
    return np.maximum(hi - lo, 0)


def resize_area(im, new_size):
    """``cv2.resize(im, new_size, interpolation=cv2.INTER_AREA)`` for a down-scale of a uint8 [h, w, C] image: every destination pixel is the
    area-weighted mean of the source pixels under it, rounded half up -- the same box filter the crop kernel applies to the drawn factor."""
    h, w, _ = im.shape
    nw, nh = int(new_size[0]), int(new_size[1])
    acc = np.einsum("ik,kjc,lj->ilc", _cover(h, nh), im.astype(np.int64), _cover(w, nw))
    return ((2 * acc + h * w) // (2 * h * w)).astype(np.uint8)


def _read_voc(root):
    """augmentation.py:13-56: the segmented, non-person, non-difficult, non-truncated objects of a Pascal-VOC tree as half-size RGBA cut-outs."""
    import xml.etree.ElementTree
    import PIL.Image
    element = _ellipse_element(8, 8)
    ann = os.path.join(root, "Annotations")
    occluders = []
    for path in sorted(p for p in (os.path.join(ann, n) for n in os.listdir(ann)) if os.path.isfile(p)):           # list_filepaths :126-129
        node = xml.etree.ElementTree.parse(path).getroot()
        if node.find("segmented").text == "0":
            continue
        wanted = []
        for i_obj, obj in enumerate(node.findall("object")):
            if obj.find("name").text == "person" or obj.find("difficult").text != "0" or obj.find("truncated").text != "0":
                continue
            box = obj.find("bndbox")
            wanted.append((i_obj, [int(box.find(k).text) for k in ("xmin", "ymin", "xmax", "ymax")]))
        if not wanted:
            continue
        name = node.find("filename").text
        with PIL.Image.open(os.path.join(root, "JPEGImages", name)) as f:
            image = np.asarray(f)
        with PIL.Image.open(os.path.join(root, "SegmentationObject", name.replace("jpg", "png"))) as f:
            labels = np.asarray(f)                                                                                   # palette indices = instance labels
        for i_obj, (x0, y0, x1, y1) in wanted:
            mask = np.where(labels[y0:y1, x0:x1] == i_obj + 1, 255, 0).astype(np.uint8)
            if np.count_nonzero(mask) < 500:                                                                        # :44-46 small objects
                continue
            mask[_erode(mask, element) < mask] = 192                                                                # :49-50 softer border
            rgba = np.concatenate([image[y0:y1, x0:x1], mask[:, :, None]], axis=2)
            half = np.round(np.array([rgba.shape[1], rgba.shape[0]]) * 0.5).astype(int)                             # resize_by_factor(.., 0.5) :52, :121
            occluders.append(resize_area(rgba, half))
    return occluders


def load_occluders(pascal_voc_root_path=None, count=16, seed=0):
    """augmentation.py:9-58.  With a Pascal-VOC tree: its occluders, in the reference's order.  Without one (none on the boxes; the reference's
    default path is the author's disk) -> ``count`` procedural occluders, seeded."""
    if pascal_voc_root_path and os.path.isdir(os.path.join(str(pascal_voc_root_path), "Annotations")):
        return _read_voc(str(pascal_voc_root_path))
    rng = np.random.default_rng(seed)
    return [_procedural_occluder(rng) for _ in range(count)]


class OccluderBank:
    """The occluder images in device memory, as ``epi_crop_patches_occluded`` reads them."""

    def __init__(self, occluders, device):
        self.hw_host = np.array([[o.shape[0], o.shape[1]] for o in occluders], np.int32)
        sizes = [o.size for o in occluders]
        self.offset_host = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        flat = np.concatenate([np.ascontiguousarray(o, dtype=np.uint8).reshape(-1) for o in occluders])
        self.bank = torch.from_numpy(flat).to(device)
        self.offset = torch.from_numpy(self.offset_host).to(device)
        self.hw = torch.from_numpy(self.hw_host).to(device)
        self.count = len(occluders)

    def tensors(self):
        return self.bank, self.offset, self.hw


def draw_occlusion(patch_hw, bank_hw, np_rng=np.random, py_rng=random):
    """The random draws of ``occlude_with_objects`` (augmentation.py:61-81) in the reference's order, turned into what the kernel needs:
    int32 [MAX_OCCLUDERS, 5] rows (occluder index | -1, pasted width, pasted height, x0, y0)."""
    out = np.full((MAX_OCCLUDERS, 5), -1, np.int32)
    width_height = np.asarray([patch_hw[1], patch_hw[0]])
    im_scale_factor = min(width_height) / 256
    count = np_rng.randint(1, 8)
    for k in range(count):
        idx = py_rng.choice(range(len(bank_hw)))
        factor = np_rng.uniform(0.2, 1.0) * im_scale_factor
        center = np_rng.uniform([0, 0], width_height)
        h, w = int(bank_hw[idx][0]), int(bank_hw[idx][1])
        # resize_by_factor :121-123: new size = round(size * factor); the kernel resizes with INTER_AREA when that shrinks the occluder and with
        # INTER_LINEAR when it grows it (factor > 1: patches larger than 256 px, e.g. the 384 px configuration's im_scale_factor = 1.5)
        w, h = (int(v) for v in np.round(np.array([w, h]) * factor).astype(int))
        if w < 1 or h < 1:
            continue
        c = np.round(center).astype(np.int32)                # paste_over :101-103
        out[k] = (idx, w, h, int(c[0]) - w // 2, int(c[1]) - h // 2)
    # rows skipped above would end the kernel's list early: compact
    keep = out[out[:, 0] >= 0]
    out[:] = -1
    out[:len(keep)] = keep
    return out
