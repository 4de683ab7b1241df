#!/usr/bin/env python
"""Weight-gradient plans of every ResNet convolution behind the stem (host only, no GPU): tile configuration, reduction splits and
the fp32 slab volume each launch writes and the deferred sum reads back.
    python tools/wgrad_plan_report.py [--layers 50] [--batch 32] [--image 256]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from epipolarpose_amd import hip  # noqa: E402

SPEC = {18: ((2, 2, 2, 2), False), 34: ((3, 4, 6, 3), False), 50: ((3, 4, 6, 3), True), 101: ((3, 4, 23, 3), True), 152: ((3, 8, 36, 3), True)}


def convs(layers, image):
    units, bottleneck = SPEC[layers]
    h, inpl, exp = image // 4, 64, 4 if bottleneck else 1
    for st, (pl, n) in enumerate(zip((64, 128, 256, 512), units), 1):
        for u in range(n):
            s = 2 if (u == 0 and st > 1) else 1
            stages = ((("c1", inpl, pl, 1, 1, h), ("c2", pl, pl, 3, s, h), ("c3", pl, pl * 4, 1, 1, h // s)) if bottleneck else
                      (("c1", inpl, pl, 3, s, h), ("c2", pl, pl, 3, 1, h // s)))
            for name, cin, cout, k, ss, hh in stages:
                yield "l%d.%d.%s" % (st, u, name), cin, cout, k, ss, hh
            if s != 1 or inpl != pl * exp:
                yield "l%d.%d.ds" % (st, u), inpl, pl * exp, 1, s, h
            inpl, h = pl * exp, h // s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--image", type=int, default=256)
    a = ap.parse_args()
    total, n = 0, 0
    print("# weight-gradient plans, ResNet-%d, batch %d, %dx%d: dW[Cout][k*k*Cin] = sum over R rows" % (a.layers, a.batch, a.image, a.image))
    print("%-10s %7s %5s %6s | %-9s %5s %6s %6s %8s" % ("layer", "R", "Cout", "kkCin", "tile", "tiles", "splits", "rows", "slab MB"))
    for name, cin, cout, k, s, h in convs(a.layers, a.image):
        ho = (h + 2 * (k // 2) - k) // s + 1
        r = a.batch * ho * ho
        p = hip.gemm_tn_plan(r, cout, cin, k * k)
        total += p["slab_bytes"]
        n += 1
        print("%-10s %7d %5d %6d | %-9s %5d %6d %6d %8.1f" % (name, r, cout, k * k * cin, ("128x128", "64x128", "256x256")[p["cfg"]], p["tiles"], p["nsplit"],
                                                              p["rows_per_split"], p["slab_bytes"] / 1e6))
    print("# %d layers, %.0f MB of fp32 slabs written and read back per backward pass (one launch per layer, EPI_WGRAD_GROUP=0)" % (n, total / 1e6))
    # grouped launches (round 3, the default): the weight gradients of a whole stage leave together (csrc/torch_glue.cpp side_group_flush:
    # behind the unit that carries the downsample projection, or when one more unit would not fit into epi_wgrad_group_max() items)
    cap = hip.wgrad_group_max()
    groups, cur = [], []
    units = {}
    for name, cin, cout, k, s, h in convs(a.layers, a.image):
        units.setdefault(name.rsplit(".", 1)[0], []).append((name, cin, cout, k, s, h))
    for uname in reversed(list(units)):                    # backward order
        cur += units[uname]
        if any(n.endswith(".ds") for n, *_ in units[uname]) or len(cur) + 4 > cap:
            groups.append(cur)
            cur = []
    if cur:
        groups.append(cur)
    print("#\n# grouped plan (EPI_WGRAD_GROUP=2): one launch per tile class and group")
    print("%-22s %6s %9s %9s %10s" % ("group (backward order)", "items", "unsplit", "max split", "slab MB"))
    gtotal = 0
    for g in groups:
        shapes = [("conv", a.batch, h, h, cin, cout, k, s, k // 2) for _, cin, cout, k, s, h in g]
        gtotal_prev = gtotal
        slab, ns = 0, []
        for b in range(0, len(shapes), cap):
            sb, nn = hip.wgrad_group_plan(shapes[b:b + cap])
            slab += sb
            ns += nn
        gtotal += slab
        # workgroups per launch: tiles x splits, by tile class (64 x 128 tiles for <= 64 output channels on 768 slots, 128 x 128 on 512)
        wg = [0, 0]
        for (_, cin, cout, k, s, h), v in zip(g, ns):
            cls = 1 if cout <= 64 else 0
            wg[cls] += -(-cout // (64 if cls else 128)) * -(-(k * k * cin) // 128) * v
        print("%-22s %6d %9d %9d %10.1f   workgroups: %d of 128x128 (512 slots), %d of 64x128 (768 slots)" % (
            g[0][0].rsplit(".", 1)[0] + " .. " + g[-1][0].rsplit(".", 1)[0], len(g), sum(1 for v in ns if v == 1), max(ns), slab / 1e6, wg[0], wg[1]))
    print("# %d groups, %.0f MB of fp32 slabs per backward pass" % (len(groups), gtotal / 1e6))


if __name__ == "__main__":
    main()
