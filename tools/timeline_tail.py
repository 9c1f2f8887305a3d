#!/usr/bin/env python3
"""Timeline of the last kernels and copies of a rocprofv3 --kernel-trace --memory-copy-trace run (rocpd sqlite):
   python tools/timeline_tail.py <results.db> [count]   -- start / end relative to the first listed event, in us."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
if "kernels" in tabs:
    for name, s, e in db.execute("select name, start, end from kernels"):
        ev.append((s, e, name.split("(")[0].replace("void ", "")[:40]))
if "memory_copies" in tabs:
    cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
    namecol = "name" if "name" in cols else cols[0]
    sizecol = "size" if "size" in cols else None
    q = "select %s, start, end%s from memory_copies" % (namecol, (", " + sizecol) if sizecol else "")
    for row in db.execute(q):
        ev.append((row[1], row[2], "COPY %s %s" % (row[0], row[3] if sizecol else "")))
ev.sort()
ev = ev[-count:]
t0 = ev[0][0]
prev_end = None
for s, e, name in ev:
    gap = "" if prev_end is None else "gap %7.2f" % ((s - prev_end) / 1e3)
    print("%10.2f %10.2f  dur %8.2f  %-12s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, name))
    prev_end = e
