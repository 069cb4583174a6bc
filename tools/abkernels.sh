#!/bin/bash
# Run ON THE GPU BOX (through gpurun): per-kernel A/B of two builds of libtomo_hip.so on the SAME box (box-to-box spread is
# +-5 %, so numbers from different gpurun calls do not compare): rocprofv3 --kernel-trace --stats of the short bench for each
# library (TOMO_LIB), then the average kernel times side by side.
#   usage: bash tools/abkernels.sh <libA.so> <libB.so> [extra bench args]
set -e
A=$1; B=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for tag in A B A2 B2; do
  lib=$A; [ "${tag:0:1}" = "B" ] && lib=$B
  TOMO_LIB=$ROOT/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o x -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras "$@" > $OUT/$tag.log 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
t = {}
for tag in ("A", "B", "A2", "B2"):
    d = {}
    for r in csv.DictReader(open("%s/%s/x_kernel_stats.csv" % (out, tag))):
        d[r["Name"][:48]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    t[tag] = d
names = [n for n in t["A"] if not n.startswith(("void at::", "__amd", "void (anonymous"))]
print("%-50s %9s %9s %9s %9s" % ("kernel (avg us)", "A", "B", "A again", "B again"))
tot = collections.defaultdict(float)
for n in sorted(names, key=lambda n: -t["A"][n][0]):
    row = [t[k].get(n, (float("nan"), 0))[0] for k in ("A", "B", "A2", "B2")]
    per_pass = t["A"][n][1] / 12.0
    for k, v in zip(("A", "B", "A2", "B2"), row):
        tot[k] += v * per_pass
    print("%-50s %9.1f %9.1f %9.1f %9.1f" % (n, *row))
print("%-50s %9.1f %9.1f %9.1f %9.1f" % ("sum per pass", tot["A"], tot["B"], tot["A2"], tot["B2"]))
for tag in ("A", "B", "A2", "B2"):
    import json
    line = [l for l in open("%s/%s.log" % (out, tag)) if l.startswith("{")][-1]
    print(tag, "ms_per_step", json.loads(line)["ms_per_step"])
PY
