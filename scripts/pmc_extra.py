"""Per-kernel averages of one rocprofv3 --pmc pass (counter_collection csv), pn2:: kernels only.
usage: pmc_extra.py <counter_collection.csv>"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "pn2::" in n:
        agg[n.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(v) / len(v), 3) for c, v in agg[k].items()}, "launches", len(next(iter(agg[k].values()))))
