"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel stats table."""
import sqlite3
import sys


def main(path, top=60):
    db = sqlite3.connect(path)
    rows = list(db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# {path}: {sum(r[1] for r in rows)} dispatches, {tot/1e6:.3f} ms total kernel time")
    print("# pct  calls  avg_us  min_us  max_us  total_ms  name")
    for r in rows[:top]:
        print(f"{r[2]/tot*100:6.2f} {r[1]:6d} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {r[5]/1e3:9.2f} {r[2]/1e6:9.3f}  {r[0][:140]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)
