#!/usr/bin/env python3
"""Steady-state per-kernel summary from a rocprofv3 --kernel-trace CSV: keeps only launches that START inside the last
`--last` seconds-fraction of the traced interval (drops warm-up, library solution search, graph capture), groups by kernel
name.  usage: trace_window.py <kernel_trace.csv> --frac 0.5 --steps-in-window N  -> CSV on stdout."""
import argparse
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    if name.startswith("Cijk_"):
        m = re.search(r"(Cijk_\w+?_SB)_(MT\d+x\d+x\d+)", name)
        return f"{m.group(1)}_{m.group(2)}..." if m else name[:60]
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
    tag = ""
    if name.startswith("at::native::") and m:
        inner = re.findall(r"at::native::(?:\(anonymous namespace\)::)?(\w+)", name)
        tag = " [" + ",".join(dict.fromkeys(inner[:4])) + "]"
    name = re.sub(r"\(.*", "", name)
    return (name[:80] + tag)[:130]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--frac", type=float, default=0.5, help="fraction of the traced time span to keep (from the end)")
    ap.add_argument("--steps-in-window", type=float, default=0, help="if given: also print launches and microseconds per step")
    ap.add_argument("--top", type=int, default=70)
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
    cut = t1 - (t1 - t0) * a.frac
    by = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, n in rows:
        if s >= cut:
            by[short(n)][0] += 1
            by[short(n)][1] += e - s
            busy += e - s
    w = csv.writer(sys.stdout)
    n_launch = sum(v[0] for v in by.values())
    w.writerow(["# window_s", round((t1 - cut) * 1e-9, 4), "kernel_busy_s", round(busy * 1e-9, 4), "launches", n_launch,
                "launches_per_step", round(n_launch / a.steps_in_window, 1) if a.steps_in_window else "",
                "kernel_us_per_step", round(busy * 1e-3 / a.steps_in_window, 1) if a.steps_in_window else ""])
    w.writerow(["kernel", "calls", "total_us", "avg_us", "pct", "calls_per_step", "us_per_step"])
    for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:a.top]:
        w.writerow([k, c, round(t * 1e-3, 1), round(t * 1e-3 / c, 2), round(100.0 * t / busy, 2),
                    round(c / a.steps_in_window, 2) if a.steps_in_window else "",
                    round(t * 1e-3 / a.steps_in_window, 1) if a.steps_in_window else ""])


if __name__ == "__main__":
    main()
