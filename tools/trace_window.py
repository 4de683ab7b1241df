#!/usr/bin/env python
"""Steady-state kernel summary from a rocprofv3 kernel trace: aggregate by kernel name over the window spanned by the
last N launches of a marker kernel (one launch per training step), dropping warm-up / MIOpen find traffic.
    python tools/trace_window.py <kernel_trace.csv> <marker substring> <n_steps> <out.csv>"""
import csv
import sys
from collections import defaultdict


def main():
    path, marker, n_steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < n_steps + 1:
        raise SystemExit("marker %r seen %d times, need %d" % (marker, len(marks), n_steps + 1))
    lo, hi = marks[-(n_steps + 1)], marks[-1]          # n_steps full step periods
    window = rows[lo + 1:hi + 1]
    span = rows[hi][1] - rows[lo][1]
    agg = defaultdict(lambda: [0, 0, 1 << 62, 0])
    busy = 0
    for s, e, name in window:
        a = agg[name]
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
        busy += e - s
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["# steady-state window: %d steps, %.3f ms/step wall, %.3f ms/step kernel-busy, %d launches/step"
                    % (n_steps, span / n_steps / 1e6, busy / n_steps / 1e6, len(window) // n_steps)])
        w.writerow(["Name", "CallsPerStep", "TotalUsPerStep", "AverageUs", "MinUs", "MaxUs", "PercentOfBusy"])
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name[:160], round(a[0] / n_steps, 2), round(a[1] / n_steps / 1e3, 2), round(a[1] / a[0] / 1e3, 2),
                        round(a[2] / 1e3, 2), round(a[3] / 1e3, 2), round(100.0 * a[1] / busy, 2)])


if __name__ == "__main__":
    main()
