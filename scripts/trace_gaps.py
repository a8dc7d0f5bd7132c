#!/usr/bin/env python3
"""Idle intervals of the dense queue over the last steps of a rocprofv3 --kernel-trace CSV of `bench_train.py --graph`:
for each of the last `--steps` steps (delimited by the optimiser's last kernel) every interval > `--min-us` in which the
step's queue ran nothing, with the kernels on either side.  Answers "where does the step's stream sit idle" (graph-launch
boundaries, cross-stream waits) -- a sum of kernel durations cannot.
usage: trace_gaps.py <kernel_trace.csv> [--steps 4] [--min-us 3]"""
import argparse
import csv
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from trace_window import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="adam_advance_kernel")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--min-us", type=float, default=3.0)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.steps + 2:
        sys.exit("too few steps in the trace")
    dense_q = rows[marks[-1]][3]
    dense = [r for r in rows if r[3] == dense_q]
    dmarks = [i for i, r in enumerate(dense) if a.marker in r[2]]
    for k in range(a.steps, 0, -1):
        lo, hi = dmarks[-k - 1], dmarks[-k]
        step = dense[lo:hi + 1]
        t0 = step[0][1]
        period = (dense[hi][1] - dense[lo][1]) * 1e-3
        busy = sum(e - s for s, e, *_ in step[1:]) * 1e-3
        print(f"step -{k}: period {period:.1f} us, dense-queue busy {busy:.1f} us, idle {period - busy:.1f} us, launches {len(step) - 1}")
        prev = step[0]
        for r in step[1:]:
            gap = (r[0] - prev[1]) * 1e-3
            if gap > a.min_us:
                print(f"    idle {gap:7.1f} us at +{(prev[1] - t0) * 1e-3:8.1f}:  {short(prev[2])[:50]}  ->  {short(r[2])[:50]}")
            if r[1] > prev[1]:
                prev = r


if __name__ == "__main__":
    main()
