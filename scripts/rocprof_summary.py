"""Summarise a rocprofv3 results.db (sqlite, --kernel-trace) into a per-kernel stats table."""
import sqlite3
import sys


def main(db_path, title=""):
    cur = sqlite3.connect(db_path).cursor()
    print(f"# {title}")
    print("# per-kernel durations in microseconds (rocprofv3 --kernel-trace)")
    print("%-104s %6s %12s %12s %12s %14s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_us"))
    q = ("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e3 "
         "from kernels group by name order by 6 desc")
    for r in cur.execute(q).fetchall():
        print("%-104s %6d %12.1f %12.1f %12.1f %14.1f" % (r[0][:104], r[1], r[2], r[3], r[4], r[5]))


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))
