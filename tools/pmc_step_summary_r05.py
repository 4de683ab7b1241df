#!/usr/bin/env python
"""Per kernel CLASS of the bench step, from three rocprofv3 --pmc passes of the same `bench.py --steps K --warmup W` command (tools/gpu_pmc_step_r05.sh):
HBM traffic (FETCH_SIZE pass, WRITE_SIZE pass; gfx950: read bytes = 2 x FETCH_SIZE, MI355X_MICROARCH.md HBM section) and matrix-pipe occupancy
(SQ pass: SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 XCDs = the share of the kernel's own cycles a SIMD's matrix pipe was busy).
A class = kernel template + the arguments that decide its work (tile configuration, addressing mode, RED).  Every per-launch figure divides by the launches
THAT PASS counted.  Usage: pmc_step_summary_r05.py fetch.csv write.csv sq.csv <steps run> out.csv out.json"""
import csv
import json
import re
import sys
from collections import defaultdict

FAMILIES = [("implicit_gemm", ("head_gemm_kernel", "head_gemm_tn_kernel", "head_gemm_tn_group_kernel", "head_gemm_astat_kernel", "conv_patch_kernel",
                               "splitk_finish", "slab_reduce")),
            ("batchnorm", ("bn_",)), ("softargmax", ("softargmax",)), ("adam", ("adam_",))]


def family(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return "other"


def klass(name):
    n = name.split("(")[0].replace("void ", "").replace("epi::", "")
    return re.sub(r"\s+", "", n)[:110]


def load(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = klass(r["Kernel_Name"])
            c = acc[k][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch, write, sq, steps, out, out_json = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]
    f, w, s = load(fetch), load(write), load(sq)
    rows, fams = [], defaultdict(lambda: defaultdict(float))
    for k in sorted(set(f) | set(w) | set(s)):
        nf, sf = f.get(k, {}).get("FETCH_SIZE", [0, 0.0])
        nw, sw = w.get(k, {}).get("WRITE_SIZE", [0, 0.0])
        sqk = s.get(k, {})
        ns = sqk.get("SQ_BUSY_CYCLES", [0, 0.0])[0]
        rb, wb = 2.0 * sf * 1024.0, sw * 1024.0                       # totals over the pass
        mfma, gui = sqk.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1], sqk.get("GRBM_GUI_ACTIVE", [0, 0.0])[1]
        wave, wait_any, wait_inst, wait_lds = (sqk.get(c, [0, 0.0])[1] for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"))
        busy = (mfma / 1024.0) / (gui / 8.0) if gui > 0 else 0.0
        row = {"kernel_class": k, "family": family(k), "launches_per_step": round(max(nf, nw, ns) / steps, 2),
               "read_MB_per_launch": round(rb / max(nf, 1) / 1e6, 3), "write_MB_per_launch": round(wb / max(nw, 1) / 1e6, 3),
               "hbm_MB_per_step": round((rb / max(nf, 1) * nf + wb / max(nw, 1) * nw) / steps / 1e6, 2),
               "mfma_busy": round(busy, 4), "kernel_cycles_per_launch": round(gui / 8.0 / max(ns, 1)),
               "wait_any_of_wave": round(wait_any / wave, 3) if wave else 0.0, "wait_inst_of_wave": round(wait_inst / wave, 3) if wave else 0.0,
               "wait_lds_of_wave": round(wait_lds / wave, 3) if wave else 0.0}
        rows.append(row)
        fm = fams[row["family"]]
        fm["launches_per_step"] += row["launches_per_step"]
        fm["read_bytes_per_step"] += rb / steps
        fm["write_bytes_per_step"] += wb / steps
        fm["mfma_busy_cycles_per_step"] += mfma / steps
        fm["gui_active_per_step"] += gui / steps
        fm["pmc_launches_fetch"] += nf
        fm["pmc_launches_write"] += nw
    rows.sort(key=lambda r: -r["hbm_MB_per_step"])
    with open(out, "w", newline="") as fo:
        wr = csv.DictWriter(fo, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        wr.writerows(rows)
    famj = {}
    for name, fm in fams.items():
        hb = fm["read_bytes_per_step"] + fm["write_bytes_per_step"]
        famj[name] = {"launches_per_step": round(fm["launches_per_step"], 2), "read_bytes_per_step": fm["read_bytes_per_step"],
                      "write_bytes_per_step": fm["write_bytes_per_step"], "hbm_bytes_per_step": hb,
                      "hbm_bytes_per_launch": hb / max(fm["launches_per_step"], 1e-9),
                      "mfma_busy": round((fm["mfma_busy_cycles_per_step"] / 1024.0) / (fm["gui_active_per_step"] / 8.0), 4) if fm["gui_active_per_step"] else 0.0}
    meta = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / {SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS "
                      "GRBM_GUI_ACTIVE} (three passes) of bench.py, %d steps" % steps,
            "correction": "read bytes = 2 x FETCH_SIZE (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE at face value; per-launch figures divide by the "
                          "launches the SAME pass counted; mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)",
            "families": famj}
    with open(out_json, "w") as fo:
        json.dump(meta, fo, indent=1, sort_keys=True)
    print(json.dumps(famj, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
