"""Per-kernel stats over the LAST fraction of a rocprofv3 --kernel-trace csv (steady state, skips warm-up / MIOpen find).
usage: trace_tail_stats.py <kernel_trace.csv> <window_ms> <ms_per_step> [top]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
win_ms, ms_per_step = float(sys.argv[2]), float(sys.argv[3])
steps = win_ms / ms_per_step
t0 = min(int(r["Start_Timestamp"]) for r in rows)
t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - win_ms * 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) >= cut:
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print("window %.1f ms, kernel time %.2f ms per step over %.0f steps, %d dispatches per step" % ((t1 - cut) / 1e6, tot / steps / 1e6, steps, sum(a[0] for a in agg.values()) / steps))
for name, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print("%6.2f%%  %7.1f/step  avg %8.1f us  %s" % (ns / tot * 100, c / steps, ns / c / 1e3, name[:110]))
