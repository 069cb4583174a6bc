#!/bin/bash
set -e
O=gpurun_out/r3j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
DUMP_SEGS=1 python tools/slab_selfloop_bench.py 512 2048 2048 10 8 3 2>&1 | grep -E "segments:|ms per pass" | cut -c1-700
python tools/slab_selfloop_bench.py 1024 1024 1024 10 2>&1 | grep -E "ms per pass" | cut -c1-700
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3:', d['ms_per_step'], 'ms')"
python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 single-GPU:', d['ms_per_step'], 'ms')"
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_cfg5 -o x -- python3 $R/tools/slab_selfloop_bench.py 512 2048 2048 5 8 3 > $R/$O/cfg5.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_cfg5single -o x -- python3 $R/bench.py --workload cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/cfg5single.log 2>&1
cd $R
python tools/timeline.py $O/tr_cfg5 field_tile_kernel 5 | cut -c1-100
echo == cfg5 single; python tools/ktrace.py $O/tr_cfg5single | cut -c1-40,60-120 | sort -k6 -n -r | head -24
