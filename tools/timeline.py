#!/usr/bin/env python3
"""Timeline of ONE pass out of a rocprofv3 --kernel-trace CSV: from the n-th launch of `anchor` to the next one.
usage: timeline.py DIR [anchor=field_tile_kernel] [n=-2]   -> start us / gap before us / duration us / kernel name"""
import csv, glob, sys
d = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else "field_tile_kernel"
n = int(sys.argv[3]) if len(sys.argv) > 3 else -2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
at = [i for i, r in enumerate(rows) if anchor in r[2]]
a, b = at[n], at[n + 1] if n + 1 != 0 and n + 1 < len(at) else len(rows)
t0, prev, busy = rows[a][0], rows[a][0], 0
for s, e, name in rows[a:b]:
    print("%9.1f  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name[:110]))
    prev, busy = max(prev, e), busy + (e - s)
print("busy %.1f us, span %.1f us" % (busy / 1e3, (rows[b][0] - t0) / 1e3 if b < len(rows) else (prev - t0) / 1e3))
