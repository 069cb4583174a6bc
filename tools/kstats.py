#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV directory: calls, mean us, ms per pass.  usage: kstats.py DIR PASSES"""
import csv, glob, sys, collections
d = sys.argv[1]; passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    acc.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in acc.values())
print("| kernel | calls | avg us | ms per pass | % |\n|---|---|---|---|---|")
for n, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print("| `%s` | %d | %.2f | %.3f | %.2f |" % (n[:70], len(v), sum(v) / len(v), sum(v) / passes / 1e3, 100 * sum(v) / tot))
print("total kernel time per pass: %.3f ms" % (tot / passes / 1e3))
