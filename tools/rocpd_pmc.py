#!/usr/bin/env python3
"""Per-kernel averages of the hardware counters in a rocprofv3 --pmc result (rocpd SQLite .db).
usage: tools/rocpd_pmc.py <results.db>  -> markdown table: kernel, counter, dispatches, mean value per dispatch"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                       "group by kernel_name, counter_name order by kernel_name, counter_name"))
print("| kernel | counter | dispatches | mean | min | max |")
print("|---|---|---|---|---|---|")
for r in rows:
    print("| %s | %s | %d | %.1f | %.1f | %.1f |" % r)
