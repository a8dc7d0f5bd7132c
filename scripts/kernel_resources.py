"""Per-kernel register / scratch / LDS / occupancy table of every HIP source under hotrack_amd/csrc (hipcc
-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).  `--scratch` lists only the kernels that use
scratch memory (the build itself refuses them: hotrack_amd/_build.py)."""
import argparse
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hotrack_amd import _build  # noqa: E402

_FIELD = re.compile(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"^void ", "", o).split("(")[0] for o in out]


def parse_remarks(text):
    """[(mangled name, {field: value})] from the stderr of a -Rpass-analysis=kernel-resource-usage compile."""
    rows, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = (m.group(1), {})
            rows.append(cur)
            continue
        m = _FIELD.search(line)
        if m and cur is not None:
            cur[1][m.group(1).strip()] = m.group(2)
    return rows


def analyse(src):
    cmd = [_build._hipcc(), *_build.HIPCC_FLAGS, *_build.PER_FILE_FLAGS.get(os.path.basename(src), []),
           "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.devnull]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        raise RuntimeError(p.stderr[-2000:])
    return os.path.basename(src), parse_remarks(p.stderr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scratch", action="store_true")
    ap.add_argument("files", nargs="*")
    a = ap.parse_args()
    srcs = [s for s in _build.sources() if not a.files or os.path.basename(s) in a.files]
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(analyse, srcs))
    bad = 0
    for f, rows in res:
        names = demangle([r[0] for r in rows])
        for n, (_, d) in zip(names, rows):
            scr = int(d.get("ScratchSize [bytes/lane]", 0))
            bad += scr != 0
            if a.scratch and scr == 0:
                continue
            print("%-20s %-90s V=%3s A=%3s scr=%4d occ=%s lds=%s" % (f, n[:90], d.get("VGPRs"), d.get("AGPRs"), scr,
                  d.get("Occupancy [waves/SIMD]"), d.get("LDS Size [bytes/block]")))
    print("%d kernels with scratch" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
