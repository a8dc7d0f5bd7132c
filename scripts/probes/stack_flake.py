"""How often does test_mlp_stack_matches_torch fail over different seeded draws (seed_offset = 0 .. reps - 1), per shape and per
switch of hotrack_amd.train_stack?  Round 4: with unseeded conv weights the suite showed ~1 failure in 12 runs, on the round-3
build too: single ReLU mask bits flipping between the fp32 path and the fp64 reference (the fused path is bit-for-bit repeatable:
stack_determinism.py).  The test now only takes draws whose reference keeps every ReLU input KINK_MARGIN away from zero; set
PN2_KINK_MARGIN=0 to see the old rate.  usage: python scripts/probes/stack_flake.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "network"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import test_gpu_train as T  # noqa: E402

if "PN2_KINK_MARGIN" in os.environ:
    T.KINK_MARGIN = float(os.environ["PN2_KINK_MARGIN"])
from hotrack_amd import train_stack as TS  # noqa: E402

CASES = [(32 * 64, [32, 32, 64], 32), (4000, [64, 64, 128], 0), (21 * 16 * 5, [128, 128, 192], 16), (21 * 64 * 2, [128, 128, 192], 64),
         (1500, [128, 128, 512], 0), (999, [256, 256], 0), (2048, [128, 128, 384], 0), (128 * 3, [128, 128, 512], 128)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
SWITCHES = [("default", {}), ("DEFER_REDUCE=0", {"DEFER_REDUCE": False}), ("FUSED_BWD=0", {"FUSED_BWD": False}), ("FWD2=0", {"FWD2": False})]
for name, sw in SWITCHES:
    saved = {k: getattr(TS, k) for k in sw}
    for k, v in sw.items():
        setattr(TS, k, v)
    line = []
    for R, widths, K in CASES:
        bad, worst = 0, ""
        for off in range(reps):
            try:
                T.test_mlp_stack_matches_torch(R, widths, K, seed_offset=off)
            except AssertionError as e:
                bad += 1
                worst = str(e)[:60].replace("\n", " ")
        line.append("%d/%d %s" % (bad, reps, worst if bad else ""))
    for k, v in saved.items():
        setattr(TS, k, v)
    print("%-16s" % name, " | ".join(line), flush=True)
