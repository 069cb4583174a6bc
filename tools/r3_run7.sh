#!/bin/bash
set -e
O=gpurun_out/r3i; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_cfg5 -o x -- python3 $R/tools/slab_selfloop_bench.py 512 2048 2048 5 8 3 > $R/$O/cfg5.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_1024 -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/$O/b1024.log 2>&1
cd $R
for d in tr_cfg5 tr_1024; do echo == $d; python tools/ktrace.py $O/$d | grep -E "rocprim|mc3_bands|mc3_scan_kernel|uq3|morph|pack_close|field_tile" | cut -c1-30,60-120; done
grep -h "ms per pass" $O/cfg5.log | cut -c1-140
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3:', d['ms_per_step'], 'ms')"
python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 single-GPU:', d['ms_per_step'], 'ms')"
python bench.py --workload cfg4 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4 single-GPU:', d['ms_per_step'], 'ms')"
