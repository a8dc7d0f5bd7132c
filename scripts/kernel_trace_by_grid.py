"""Per (kernel, grid size) launch statistics from a rocprofv3 --kernel-trace csv -- the --stats summary merges launches
of one kernel symbol that differ only in grid (e.g. the K=64 and K=16 launches of one sa_mlp_max_kernel instance)."""
import collections
import csv
import sys

agg = collections.defaultdict(list)
def _prod(r, base):
    if base in r:
        return int(r[base])
    return int(r[base + "_X"]) * int(r.get(base + "_Y", 1) or 1) * int(r.get(base + "_Z", 1) or 1)


for r in csv.DictReader(open(sys.argv[1])):
    agg[(r["Kernel_Name"], _prod(r, "Grid_Size"), _prod(r, "Workgroup_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("kernel,grid_threads,workgroup,calls,avg_us,min_us,max_us")
for (name, grid, wg), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('"%s",%d,%d,%d,%.2f,%.2f,%.2f' % (name, grid, wg, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
