#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite) -> timeline of ONE steady-state step of the bench loop: every kernel dispatch between
the starts of two consecutive Newton launches (the 8th and 9th k_thorough_dna of the trace), with start offset,
duration, stream/queue and the gap to the previous kernel's end.   python step_timeline.py results.db [n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kt[0])]
print("# table", kt[0], cols, file=sys.stderr)
name_c = "name" if "name" in cols else [c for c in cols if "name" in c][0]
q_c = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
s_c = "stream_id" if "stream_id" in cols else None
sel = "select %s, start, end%s%s from %s order by start" % (name_c, (", " + q_c) if q_c else "", (", " + s_c) if s_c else "", kt[0])
rows = list(cur.execute(sel))
th = [i for i, r in enumerate(rows) if "k_thorough" in r[0]]
a, b = th[nth], th[nth + 1]
t0 = rows[a][1]
prev_end = None
print("%-58s %10s %10s %9s  %s" % ("kernel", "start_us", "dur_us", "gap_us", "queue/stream"))
for r in rows[a:b + 1]:
    gap = (r[1] - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%-58s %10.1f %10.1f %9.1f  %s" % (r[0][:58], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap, " ".join(str(x) for x in r[3:])))
    prev_end = max(prev_end, r[2]) if prev_end is not None else r[2]
