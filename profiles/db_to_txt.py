#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> plain-text kernel summary: python db_to_txt.py results.db"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in cur.execute(
        "select name, total_calls, total_duration, average, percentage from top_kernels"):
    print("%-90s %8d %14.1f %12.2f %7.2f" % (name[:90], calls, total, avg, pct))
