#!/usr/bin/env python
"""Every launch of ONE steady-state training step, in start order, from a rocprofv3 kernel trace: short kernel name, queue, grid
(workgroups), LDS bytes, start offset, duration and the idle gap on the launch's own queue -- the true per-layer durations inside
the step (the per-layer micro-benchmark is host-bound below ~12 us per call) and how launches of the two streams overlap.
    python tools/trace_step_sequence.py <kernel_trace.csv> <marker substring> <out.txt> [k]     (the step that ends with the k-th marker
    launch, Python index; default -2)"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("epi::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    path, marker, out = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            wg = [int(r.get("Workgroup_Size_" + a, r.get("Workgroup_Size", 1)) or 1) for a in "XYZ"]
            gr = [int(r.get("Grid_Size_" + a, r.get("Grid_Size", 1)) or 1) for a in "XYZ"]
            nwg = 1
            for g, w in zip(gr, wg):
                nwg *= max(1, g // max(1, w))
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), nwg,
                         int(r.get("LDS_Block_Size", 0) or 0)))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        raise SystemExit("marker %r seen %d times" % (marker, len(marks)))
    k = int(sys.argv[4]) if len(sys.argv) > 4 else -2
    lo, hi = marks[k - 1], marks[k]
    window = rows[lo + 1:hi + 1]
    t0 = window[0][0]
    queues = sorted({r[3] for r in window})
    last_end = {}
    busy = 0
    # union of busy intervals (overlap between queues counted once)
    union, cur_s, cur_e = 0, None, None
    for s, e, *_ in window:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    with open(out, "w") as f:
        f.write("# one step: %d launches, %.3f ms wall, %.3f ms summed kernel time, %.3f ms with at least one kernel running; queues %s\n"
                % (len(window), (window[-1][1] - t0) / 1e6, sum(e - s for s, e, *_ in window) / 1e6, union / 1e6, queues))
        f.write("%9s %8s %7s %2s %6s %6s  %s\n" % ("start_us", "dur_us", "gap_us", "q", "wgs", "lds", "kernel"))
        for s, e, name, q, nwg, lds in window:
            gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
            last_end[q] = e
            f.write("%9.1f %8.2f %7.2f %2d %6d %6d  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, gap, queues.index(q), nwg, lds, short(name)))


if __name__ == "__main__":
    main()
