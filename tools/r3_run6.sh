#!/bin/bash
set -e
O=gpurun_out/r3h; mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
for cfg in narrow wide; do
  TOMO_EXP_SORT_CFG=$cfg rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_cfg5_$cfg -o x -- python3 $R/tools/slab_selfloop_bench.py 512 2048 2048 5 8 3 > $R/$O/cfg5_$cfg.log 2>&1
  TOMO_EXP_SORT_CFG=$cfg rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_1024_$cfg -o x -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/$O/b1024_$cfg.log 2>&1
done
cd $R
for d in tr_cfg5_narrow tr_cfg5_wide tr_1024_narrow tr_1024_wide; do echo == $d; python tools/ktrace.py $O/$d rocprim | cut -c1-20,60-120; done
grep -h "ms per pass" $O/cfg5_*.log | cut -c1-140
python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 single-GPU (auto cfg):', d['ms_per_step'], 'ms')"
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
