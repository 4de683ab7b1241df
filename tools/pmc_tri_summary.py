#!/usr/bin/env python
"""Per triangulation kernel: share of a SIMD's cycles its vector ALU was issuing, from one rocprofv3 --pmc pass over tools/bench_kernels.py tri
(SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY count quad-cycles summed over waves; GRBM_GUI_ACTIVE counts cycles summed over the 8 XCDs).
    python tools/pmc_tri_summary.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "triangulate" not in k:
        continue
    key = ("staged  " if "staged" in k else "per-item") + (" f32 " if "<float" in k else " f64 ") + {"0": "iterative", "1": "ls", "2": "dlt", "3": "poly"}.get(k.split(",")[2].strip(" >)")[:1], "?")
    rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[key] += 1
print("%-28s %8s %12s %10s %10s %10s %10s" % ("kernel", "launches", "cycles/launch", "valu_busy", "wait_any", "wait_inst", "waves/SIMD"))
for key, c in sorted(rows.items()):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0 / max(n[key], 1)                 # kernel duration in shader cycles
    simd_cycles = cyc * 1024.0 * max(n[key], 1)                       # 256 CUs x 4 SIMDs
    wave = c["SQ_WAVE_CYCLES"] * 4.0
    print("%-28s %8d %12.0f %10.3f %10.3f %10.3f %10.2f" % (key, n[key], cyc, c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, c["SQ_WAIT_ANY"] * 4.0 / max(wave, 1),
                                                         c["SQ_WAIT_INST_ANY"] * 4.0 / max(wave, 1), wave / simd_cycles))
