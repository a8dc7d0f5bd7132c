"""Per-kernel averages of rocprofv3 --pmc passes (counter_collection csv, one counter per pass), pn2:: kernels only.
usage: pmc_extra.py <counter_collection.csv> [<counter_collection.csv> ...]
With SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE given, also prints MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * 1024 SIMDs)
(MI355X: 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3, hence / 8 first)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "pn2::" in n:
            agg[(n.split("(")[0].replace("void ", ""), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    c = {name: sum(v) / len(v) for name, v in agg[k].items()}
    line = "%s grid=%d %s launches=%d" % (k[0], k[1], {n: round(v, 1) for n, v in c.items()}, len(next(iter(agg[k].values()))))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
        line += "  MfmaUtil(all 1024 SIMDs, GUI_ACTIVE/8 XCDs) = %.1f %%" % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
        line += "  raw ratio MFMA_BUSY/GUI_ACTIVE = %.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"])
    print(line)
