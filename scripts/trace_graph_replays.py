#!/usr/bin/env python3
"""One replay of a captured graph, kernel by kernel, from a rocprofv3 --kernel-trace CSV: the trace is cut at every launch of
`--marker` (the first kernel of a replay) and the LAST complete segment is listed in start order with durations and idle gaps.
usage: trace_graph_replays.py <kernel_trace.csv> --marker hand_frame_kernel"""
import argparse
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from trace_window import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", required=True)
    ap.add_argument("--skip-last", type=int, default=1, help="segments to skip from the end (the last one may be cut off)")
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", ""))))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.skip_last + 2:
        sys.exit("not enough replays in the trace")
    lo, hi = marks[-(a.skip_last + 2)], marks[-(a.skip_last + 1)]
    step = rows[lo:hi]
    busy = sum(e - s for s, e, _, _ in step)
    w = csv.writer(sys.stdout)
    w.writerow(["# launches", len(step), "span_us", round((step[-1][1] - step[0][0]) * 1e-3, 1), "kernel_busy_us", round(busy * 1e-3, 1)])
    w.writerow(["i", "start_us", "dur_us", "gap_us", "grid", "kernel"])
    prev_end = step[0][0]
    for i, (s, e, n, g) in enumerate(step):
        w.writerow([i, round((s - step[0][0]) * 1e-3, 1), round((e - s) * 1e-3, 2), round((s - prev_end) * 1e-3, 2), g, short(n)])
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
