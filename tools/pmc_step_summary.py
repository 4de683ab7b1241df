#!/usr/bin/env python
"""HBM traffic of the bench step per kernel and per kernel family from two rocprofv3 counter_collection.csv files (FETCH_SIZE pass,
WRITE_SIZE pass) of the same `bench.py --steps K --warmup W` command.  Totals are divided by the number of steps the process ran
(warm-up included: every step launches the same kernels).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the
bytes of wide coalesced reads -> read bytes = 2 * FETCH_SIZE KiB * 1024; WRITE_SIZE is taken at face value (KiB, uncalibrated).
Usage: pmc_step_summary.py fetch.csv write.csv <steps run> out.csv out_families.json"""
import csv
import json
import sys
from collections import defaultdict

FAMILIES = [
    ("implicit_gemm", ("head_gemm_kernel", "head_gemm_tn_kernel", "head_gemm_tn_group_kernel", "head_gemm_astat_kernel", "conv_patch_kernel",
                       "splitk_finish", "slab_reduce")),
    ("batchnorm", ("bn_",)),
    ("softargmax", ("softargmax",)),
    ("adam", ("adam_",)),
]


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def load(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[name][0] += 1
            acc[name][1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, steps, out, out_json = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    fams = defaultdict(lambda: {"launches_per_step": 0.0, "read_bytes_per_step": 0.0, "write_bytes_per_step": 0.0})
    rows = []
    for k in set(f) | set(w):
        nf, sf = f.get(k, [0, 0.0])
        nw, sw = w.get(k, [0, 0.0])
        n = max(nf, nw)
        rb, wb = 2.0 * sf * 1024.0 / steps, sw * 1024.0 / steps
        rows.append((rb + wb, k, n / steps, rb, wb))
        fm = fams[family(k)]
        fm["launches_per_step"] += n / steps
        fm["read_bytes_per_step"] += rb
        fm["write_bytes_per_step"] += wb
    rows.sort(reverse=True)
    with open(out, "w", newline="") as fo:
        wr = csv.writer(fo)
        wr.writerow(["kernel", "launches_per_step", "read_MB_per_step(2x FETCH_SIZE)", "write_MB_per_step", "hbm_MB_per_step", "hbm_MB_per_launch"])
        for tot, k, n, rb, wb in rows:
            wr.writerow([k, round(n, 1), round(rb / 1e6, 2), round(wb / 1e6, 2), round(tot / 1e6, 2), round(tot / 1e6 / max(n, 1e-9), 3)])
    for fm in fams.values():
        fm["hbm_bytes_per_step"] = fm["read_bytes_per_step"] + fm["write_bytes_per_step"]
        fm["hbm_bytes_per_launch"] = fm["hbm_bytes_per_step"] / max(fm["launches_per_step"], 1e-9)
    meta = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (two passes) of bench.py, %d steps" % steps,
            "correction": "read bytes = 2 x FETCH_SIZE (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE at face value", "families": fams}
    with open(out_json, "w") as fo:
        json.dump(meta, fo, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
