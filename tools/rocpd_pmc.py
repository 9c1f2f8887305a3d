#!/usr/bin/env python3
"""Per-kernel mean of one PMC counter from a rocprofv3 --pmc run (rocpd sqlite):
   python tools/rocpd_pmc.py gpurun_out/pmc_fetch/f_results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
q = ("select kernel_name, counter_name, grid_size_x, count(*), avg(value), min(value), max(value), max(description) "
     "from counters_collection group by kernel_name, counter_name, grid_size_x order by kernel_name, grid_size_x")
desc = None
print("%-52s %-12s %9s %7s %14s %14s %14s" % ("kernel", "counter", "grid_x", "calls", "avg", "min", "max"))
for name, cn, gx, cnt, avg, mn, mx, d in db.execute(q):
    desc = desc or d
    print("%-52s %-12s %9d %7d %14.1f %14.1f %14.1f" % (name.split("(")[0].replace("void ", "")[:52], cn, gx, cnt, avg, mn, mx))
print("counter description:", desc)
