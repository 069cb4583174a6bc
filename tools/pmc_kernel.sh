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
        if any(q in r["Kernel_Name"] for q in pat.split("|")):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    g = lambda c: (sum(d[c]) / len(d[c])) if c in d and d[c] else float("nan")
    wc = g("SQ_WAVE_CYCLES")
    print("   waves %.0f  wave_cycles %.3g  active %.0f%%  wait_mem %.0f%%  wait_issue %.0f%%  valu/active %.0f%%  TA_busy %.3g  busy_cycles %.3g  vmem_insts %.3g  valu_insts %.3g  tcc_rd %.3g tcc_wr %.3g"
          % (g("SQ_WAVES"), wc, 100 * g("SQ_ACTIVE_INST_ANY") / wc, 100 * g("SQ_WAIT_ANY") / wc, 100 * g("SQ_WAIT_INST_ANY") / wc,
             100 * g("SQ_ACTIVE_INST_VALU") / g("SQ_ACTIVE_INST_ANY"), g("TA_BUSY_avr"), g("SQ_BUSY_CYCLES"), g("SQ_INSTS_VMEM"), g("SQ_INSTS_VALU"),
             g("TCP_TCC_READ_REQ_sum"), g("TCP_TCC_WRITE_REQ_sum")))
PY
