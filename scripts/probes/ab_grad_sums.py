"""A/B of linear_dw.FUSE_GRAD_SUMS on the graph-captured training step (same box, alternating): python scripts/probes/ab_grad_sums.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = ("import sys; sys.argv = ['bench_train.py', '--graph']; sys.path[:0] = [%r, %r + '/network', %r + '/scripts'];"
        "from hotrack_amd import linear_dw; linear_dw.FUSE_GRAD_SUMS = %s; import runpy; runpy.run_path(%r + '/scripts/bench_train.py', run_name='__main__')")
for rep in range(3):
    for fuse in (True, False):
        out = subprocess.run([sys.executable, "-c", code % (ROOT, ROOT, ROOT, fuse, ROOT)], capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print("fuse" if fuse else "autograd adds", json.loads(line[-1])["ms_per_step"] if line else out.stderr[-300:])
