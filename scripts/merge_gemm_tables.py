"""Merge TunableOp result files: `python scripts/merge_gemm_tables.py BASE NEW... --out OUT`.  Rows are keyed by (operator, shape);
rows of BASE win (the shipped table's entries were recorded with longer tuning), new shapes are appended.  All files must carry the
same Validator lines (same hipBLASLt / rocBLAS build): a mismatch aborts."""
import argparse


def read(path):
    lines = [l for l in open(path).read().splitlines() if l.strip()]
    return [l for l in lines if l.startswith("Validator")], [l for l in lines if not l.startswith("Validator")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("base")
    ap.add_argument("new", nargs="+")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    head, rows = read(a.base)
    table = {",".join(r.split(",")[:2]): r for r in rows}
    n0 = len(table)
    for p in a.new:
        h, r = read(p)
        if h != head:
            raise SystemExit(f"{p}: validators differ from {a.base}: recorded on another library build")
        for row in r:
            table.setdefault(",".join(row.split(",")[:2]), row)
    with open(a.out, "w") as f:
        f.write("\n".join(head + list(table.values())) + "\n")
    print(f"{a.out}: {n0} -> {len(table)} rows")


if __name__ == "__main__":
    main()
