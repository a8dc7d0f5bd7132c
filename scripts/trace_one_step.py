#!/usr/bin/env python3
"""One training step, kernel by kernel, from a rocprofv3 --kernel-trace CSV of `bench_train.py --graph`: the launches between
the last two fused-Adam kernels, in start order, with their duration and the idle gap in front of them.
usage: trace_one_step.py <kernel_trace.csv> [--marker adam_advance_kernel]  -> CSV on stdout."""
import argparse
import csv
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from trace_window import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="adam_advance_kernel", help="kernel that ends a step (hotrack_amd.optim.FusedAdam: adam_advance_kernel; torch's fused Adam: multi_tensor_apply_kernel)")
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", r.get("Grid_Size_X", "")),
                         r.get("Queue_Id", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    # the optimiser may be several consecutive launches: a step ends at the LAST marker of a run
    ends = [i for j, i in enumerate(marks) if j + 1 == len(marks) or marks[j + 1] - i > 3]
    if len(ends) < 3:
        sys.exit("fewer than three steps in the trace")
    lo, hi = ends[-3] + 1, ends[-2] + 1
    w = csv.writer(sys.stdout)
    step = rows[lo:hi]
    busy = sum(e - s for s, e, *_ in step)
    queues = {q: i for i, q in enumerate(sorted({r[4] for r in step}))}  # hardware queue = stream (geometry prefetch runs on its own)
    w.writerow(["# launches", len(step), "span_us", round((step[-1][1] - step[0][0]) * 1e-3, 1), "kernel_busy_us", round(busy * 1e-3, 1)])
    w.writerow(["i", "start_us", "dur_us", "gap_us", "grid", "queue", "kernel"])
    prev_end = step[0][0]
    for i, (s, e, n, g, q) in enumerate(step):
        w.writerow([i, round((s - step[0][0]) * 1e-3, 1), round((e - s) * 1e-3, 2), round((s - prev_end) * 1e-3, 2), g, queues[q], short(n)])
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
