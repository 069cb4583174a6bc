#!/bin/bash
# Run ON THE GPU BOX: SQ counter passes for one small driver script; prints per-kernel sums.
#   usage: bash tools/pmc_kernel.sh <tag> <script.py> [kernel-name-substring]
set -e
TAG=$1; SCRIPT=$2; PAT=${3:-morph}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM" "SQ_BUSY_CYCLES SQ_WAVES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TCP_GATE_EN1_sum"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $OUT/$N -o x -- python3 $ROOT/$SCRIPT > $OUT/$N.log 2>&1 || echo "pass $C failed"
done
cd $ROOT
python3 - "$OUT" "$PAT" <<'PY'
import sys, glob, csv, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
