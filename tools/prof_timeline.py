#!/usr/bin/env python3
"""Timeline (us) of the kernels between two k_restore_state launches (one bench step) from a rocprofv3 .db: start, end, duration, gap to the
latest end of the kernels that started before it."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
marker = sys.argv[3] if len(sys.argv) > 3 else "k_restore_state"  # (a frame of tools/dev_frame.py: "k_build_tables", ovgpu_set_state's last launch)
marks = [i for i, r in enumerate(rows) if marker in r[0]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
a, b = marks[which], marks[which + 1]
t0 = rows[a][1]
latest = 0.0
for n, s, e in rows[a:b + 1]:
    n = n.split("(")[0].replace("void ", "").replace("ovg::", "")
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  gap {max(0.0, (s - t0) / 1e3 - latest):6.1f}  {n[:60]}")
    latest = max(latest, (e - t0) / 1e3)
