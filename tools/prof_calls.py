#!/usr/bin/env python3
"""Per-call durations (us) of the kernels whose name contains argv[2], in launch order, from a rocprofv3 .db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
pat = sys.argv[2]
d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
print(" ".join(f"{x:.0f}" for x in d))
