#!/usr/bin/env python3
"""Host API calls against device kernels over the last steps of a `rocprofv3 --kernel-trace --hip-trace` run of
bench_train.py --graph: when does the host call hipGraphLaunch / hipEventSynchronize / hipStreamWaitEvent relative to the
step's first and last kernels?  usage: trace_host_vs_device.py <dir with *_kernel_trace.csv and *_hip_api_trace.csv>"""
import csv
import glob
import sys

d = sys.argv[1]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
ht = glob.glob(d + "/**/*hip_api_trace.csv", recursive=True)[0]
K = []
for r in csv.DictReader(open(kt, newline="")):
    K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
K.sort()
H = []
for r in csv.DictReader(open(ht, newline="")):
    H.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
H.sort()
marks = [i for i, r in enumerate(K) if "adam_advance_kernel" in r[2]]
lo, hi = K[marks[-4]][1], K[marks[-1]][1]
ev = [(s, "K  %-8s q%s %s" % ("%.1f" % ((e - s) * 1e-3), q[-2:], n[:60])) for s, e, n, q in K if lo - 200000 <= s <= hi and
      ("adam_advance" in n or "copyBuffer" in n or "FillFunctor<double>" in n or "hand_frame" in n or "multi_tensor" in n)]
names = ("hipGraphLaunch", "hipEventSynchronize", "hipStreamWaitEvent", "hipEventRecord", "hipStreamSynchronize", "hipDeviceSynchronize")
ev += [(s, "H  %-8s %s" % ("%.1f" % ((e - s) * 1e-3), f)) for s, e, f in H if lo - 200000 <= s <= hi and any(f.startswith(n) for n in names)]
ev.sort()
for t, txt in ev:
    print("%10.1f  %s" % ((t - lo) * 1e-3, txt))
