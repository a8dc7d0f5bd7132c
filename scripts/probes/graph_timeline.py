"""Print the kernel timeline (start offset, duration, gap, name) of the LAST graph replay in a rocprofv3
--kernel-trace csv of scripts/bench_latency.py-like runs.  usage: graph_timeline.py <kernel_trace.csv> <kernels_per_replay>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:100]))
    prev_end = e
    tot += e - s
print("span %.1f us, kernel sum %.1f us" % ((prev_end - t0) / 1e3, tot / 1e3))
