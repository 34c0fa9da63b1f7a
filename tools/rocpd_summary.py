#!/usr/bin/env python3
"""Summarise a rocprofv3 result (rocpd SQLite .db, the ROCm 7.2 default output) as a per-kernel stats table.
usage: tools/rocpd_summary.py <results.db> [> profiles/rNN_kernel_stats.md]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = list(cur.execute(
    "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(sgpr_count), "
    "max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows) or 1
print("| kernel | calls | total us | avg us | min us | max us | % | VGPR | SGPR | LDS B | grid | wg |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %s | %s | %s | %s | %s |" % (
        r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6], r[7], r[8], r[9], r[10]))
