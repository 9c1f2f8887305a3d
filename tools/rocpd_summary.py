#!/usr/bin/env python3
"""Dumps the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as text:
   python tools/rocpd_summary.py gpurun_out/prof1/p1_results.db > profiles/r01_xxx.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("%-44s %8s %14s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, tot, avg, pct in rows:
    short = name.split("(")[0].replace("void ", "")
    print("%-44s %8d %14.1f %10.3f %7.2f" % (short, calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
print("total_us %.1f" % sum(r[2] for r in rows))
