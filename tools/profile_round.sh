#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel trace + two separate PMC passes of the default bench command,
# then tools/profile_summary.py writes the round's files under gpurun_out/prof_<tag>/ (copy them into profiles/).
#   usage: bash tools/profile_round.sh r01
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o x -- $CMD > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o x -- $CMD > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o x -- $CMD > $OUT/bench_write.log 2>&1
# the same passes with every kernel on ONE stream: kernel durations without a neighbour on the second stream
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o x -- $CMD --one-stream > $OUT/bench_trace1.log 2>&1
cd $ROOT
python3 bench.py > $OUT/bench_plain.log 2>&1
python3 tools/profile_summary.py $OUT $TAG
