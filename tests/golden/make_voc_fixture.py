#!/usr/bin/env python
"""Write a small Pascal-VOC tree in the on-disk format the reference's ``load_occluders`` reads and record what the LIVE reference makes of it.

Run in the build container only (needs ``/root/reference``):

    python tests/golden/make_voc_fixture.py

Files written under ``tests/golden/voc_fixture/`` (``lib/utils/augmentation.py:9-58``): ``Annotations/<id>.xml`` (``segmented``, ``filename`` and per
``object``: ``name``, ``difficult``, ``truncated``, ``bndbox``), ``JPEGImages/<id>.jpg``, ``SegmentationObject/<id>.png`` (palette PNG whose indices
are the instance labels 1, 2, ... in object order, 0 background, 255 void border -- as VOC stores them).  The cases: a segmented image with two
usable objects and a person; an unsegmented image (skipped); an image whose objects are truncated / difficult / too small / usable with odd
box sizes; an image with only a person (skipped).  ``tests/golden/voc_occluders.npz`` holds the occluder list of the reference's own
``load_occluders`` on this tree; ``cv2`` is not installed here, so its three calls are served by the oracle's restatements
(``getStructuringElement`` / ``erode`` / ``resize(INTER_AREA)``, oracle/imgproc.py) and ``countNonZero`` by NumPy -- everything else (the XML
walk, the filters, the cut-outs, the order, PIL's decoding) is the reference's code running.
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_shims                                   # noqa: E402

OUT = os.environ.get("EPI_GOLDEN_OUT") or HERE          # (tests/test_golden_regeneration.py regenerates into a scratch directory)
ROOT = os.path.join(OUT, "voc_fixture")


def blob(h, w, cy, cx, ry, rx, power):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    return (np.abs((yy - cy) / ry) ** power + np.abs((xx - cx) / rx) ** power) <= 1.0


def write_case(name, size, objects, segmented, seed):
    """objects: (class name, difficult, truncated, (cy, cx, ry, rx, power)) -- instance label = position + 1."""
    from PIL import Image
    h, w = size
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.stack([120 + 60 * np.sin(xx * 0.07 + c) + 40 * np.cos(yy * 0.05 - c) for c in range(3)], axis=2) + rng.normal(0, 6.0, (h, w, 3))
    labels = np.zeros((h, w), np.uint8)
    xml = ["<annotation>", "<folder>VOC2012</folder>", "<filename>%s.jpg</filename>" % name,
           "<size><width>%d</width><height>%d</height><depth>3</depth></size>" % (w, h), "<segmented>%d</segmented>" % int(segmented)]
    for i, (cls, difficult, truncated, shape) in enumerate(objects):
        m = blob(h, w, *shape)
        colour = rng.integers(20, 236, 3).astype(np.float64)
        img[m] = colour + 18.0 * np.sin(xx[m] * 0.3 + i)[:, None]
        labels[m] = i + 1
        ys, xs = np.nonzero(m)
        xml += ["<object>", "<name>%s</name>" % cls, "<pose>Unspecified</pose>", "<truncated>%d</truncated>" % int(truncated),
                "<difficult>%d</difficult>" % int(difficult),
                "<bndbox><xmin>%d</xmin><ymin>%d</ymin><xmax>%d</xmax><ymax>%d</ymax></bndbox>" % (xs.min(), ys.min(), xs.max() + 1, ys.max() + 1),
                "</object>"]
    xml.append("</annotation>")
    with open(os.path.join(ROOT, "Annotations", name + ".xml"), "w") as f:
        f.write("\n".join(xml) + "\n")
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(ROOT, "JPEGImages", name + ".jpg"), quality=90)
    if segmented:
        edge = np.zeros_like(labels, bool)                                     # VOC draws a void (255) contour around every instance
        edge[1:, :] |= labels[1:, :] != labels[:-1, :]
        edge[:, 1:] |= labels[:, 1:] != labels[:, :-1]
        seg = labels.copy()
        seg[edge] = 255
        pal = Image.fromarray(seg, mode="P")
        pal.putpalette([v for k in range(256) for v in ((k * 37) % 256, (k * 91) % 256, (k * 151) % 256)])
        pal.save(os.path.join(ROOT, "SegmentationObject", name + ".png"))


def main():
    if os.path.isdir(ROOT):
        shutil.rmtree(ROOT)
    for d in ("Annotations", "JPEGImages", "SegmentationObject"):
        os.makedirs(os.path.join(ROOT, d))
    write_case("2008_000002", (150, 200), [("dog", 0, 0, (60, 60, 40, 45, 2.0)), ("person", 0, 0, (80, 150, 60, 30, 2.5)),
                                           ("chair", 0, 0, (118, 70, 25, 55, 4.0))], True, 1)
    write_case("2008_000003", (120, 160), [("cat", 0, 0, (60, 80, 40, 50, 2.0))], False, 2)
    write_case("2008_000007", (161, 187), [("car", 0, 1, (40, 50, 30, 40, 2.0)), ("bottle", 1, 0, (40, 140, 30, 25, 2.0)),
                                           ("bird", 0, 0, (110, 30, 9, 11, 2.0)), ("sofa", 0, 0, (115, 120, 37, 52, 3.0))], True, 3)
    write_case("2008_000009", (100, 100), [("person", 0, 0, (50, 50, 40, 30, 2.0))], True, 4)
    with open(os.path.join(ROOT, "README.txt"), "w") as f:          # (a non-annotation file beside the tree must not disturb list_filepaths' directory walk)
        f.write("synthetic Pascal-VOC layout for tests/test_voc_occluders.py; written by tests/golden/make_voc_fixture.py\n")

    ref_shims.load_reference()
    from oracle import imgproc as o_img
    cv2 = sys.modules["cv2"]
    cv2.MORPH_ELLIPSE, cv2.INTER_LINEAR, cv2.INTER_AREA = 2, 1, 3
    cv2.getStructuringElement = lambda shape, ksize: o_img.structuring_ellipse(ksize)
    cv2.erode = lambda src, kernel: o_img.erode(src, kernel)
    cv2.countNonZero = lambda a: int(np.count_nonzero(a))

    def resize(im, new_size, fx=None, fy=None, interpolation=None):
        assert interpolation == cv2.INTER_AREA
        return o_img.resize_area(im, new_size)
    cv2.resize = resize
    import importlib
    aug = importlib.import_module("lib.utils.augmentation")
    occluders = aug.load_occluders(ROOT)
    out = {"count": np.int64(len(occluders))}
    for i, oc in enumerate(occluders):
        out["occluder%d" % i] = np.ascontiguousarray(oc)
    path = os.path.join(OUT, "voc_occluders.npz")
    np.savez_compressed(path, **out)
    size = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(ROOT) for f in fs)
    print("wrote %s (%d occluders: %s) and %s (%.1f KB)" % (path, len(occluders), [o.shape for o in occluders], ROOT, size / 1024.0))


if __name__ == "__main__":
    main()
