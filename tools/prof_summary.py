#!/usr/bin/env python3
"""Turns a rocprofv3 results .db (--kernel-trace --stats) into the per-kernel text summary kept under profiles/."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), avg(end-start)/1e3, sum(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size) "
                       "from kernels group by name order by 4 desc").fetchall()
    tot = sum(r[3] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary:", " ".join(sys.argv[2:]))
    print("%-72s %7s %12s %12s %7s %10s %10s %5s %5s %7s %7s" % ("kernel", "calls", "avg_us", "total_us", "pct", "min_us", "max_us", "vgpr", "agpr", "lds", "scratch"))
    for r in rows:
        print("%-72s %7d %12.1f %12.1f %6.1f%% %10.1f %10.1f %5d %5d %7d %7d" % (r[0][:72], r[1], r[2], r[3], 100 * r[3] / tot, r[4], r[5], r[6], r[7], r[8], r[9]))


if __name__ == "__main__":
    main()
