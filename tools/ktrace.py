#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name, calls / mean / min us.  usage: ktrace.py DIR [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if sub not in n: continue
    acc.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in acc.items():
    print("%-60s calls %3d mean %9.1f us min %9.1f us" % (n[:60], len(v), sum(v) / len(v), min(v)))
