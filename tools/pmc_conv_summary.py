#!/usr/bin/env python
"""Compact per-kernel table from a rocprofv3 counter_collection.csv: mean counter values per (kernel, grid) of the epi:: kernels.
    python tools/pmc_conv_summary.py <counter_collection.csv> <out.txt>"""
import csv
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if "epi::" not in name or "head_gemm" not in name:
            continue
        short = name.split("epi::")[1].split("(")[0].replace("epi::", "")[:60]
        key = (short, r["Grid_Size"], r["LDS_Block_Size"])
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[(key, r["Dispatch_Id"])] = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])]
with open(sys.argv[2], "w") as out:
    for key, cs in sorted(rows.items()):
        d = [v[0] for (k, _), v in dur.items() if k == key]
        line = "%-62s grid %-8s lds %-6s n=%d  dur %.1f us |" % (key[0], key[1], key[2], len(d), sum(d) / len(d) / 1e3)
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        for c in sorted(m):
            line += " %s=%.3g" % (c.replace("SQ_", ""), m[c])
        line += " | of wave-cycles: wait_any %.2f wait_inst %.2f active %.2f" % (m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc,
                                                                            m.get("SQ_ACTIVE_INST_ANY", 0) / wc)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
            line += " mfma_busy/busy %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(m["SQ_BUSY_CYCLES"], 1.0))
        if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
            line += " lds_conflict %.3f" % (m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0))
        if "TCC_HIT_sum" in m:
            line += " | L2 hit rate %.3f, fabric reads %.1f MB (RDREQ x 64 B x 2, the gfx950 correction)" % (
                m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0.0), 1.0), m.get("TCC_EA0_RDREQ_sum", 0.0) * 128 / 1e6)
        out.write(line + "\n")
print(open(sys.argv[2]).read())
