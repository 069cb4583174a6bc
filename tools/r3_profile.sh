#!/bin/bash
# Round-3 evidence in one GPU call: tests, round profile (kernel trace + PMC passes + plain bench), slab rehearsals with
# kernel traces (1024^3 per rank; BASELINE configs[4] per rank), other single-GPU workloads.
set -e
R=$PWD
O=gpurun_out/r3prof; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1 || { tail -30 $O/profile_round.log; exit 1; }
tail -3 $O/profile_round.log
python tools/slab_selfloop_bench.py 1024 1024 1024 10 > $O/self1024.log 2>&1
TOMO_READ_EVERY_PASS=1 python tools/slab_selfloop_bench.py 1024 1024 1024 10 > $O/self1024_serial.log 2>&1
python tools/slab_selfloop_bench.py 512 2048 2048 10 8 3 > $O/self_cfg5.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_1024 -o x -- python3 $R/tools/slab_selfloop_bench.py 1024 1024 1024 6 > $R/$O/tr_1024.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_cfg5 -o x -- python3 $R/tools/slab_selfloop_bench.py 512 2048 2048 5 8 3 > $R/$O/tr_cfg5.log 2>&1
cd $R
python tools/timeline.py $O/tr_1024 field_tile_kernel 6 > $O/timeline_1024.txt
python tools/timeline.py $O/tr_cfg5 field_tile_kernel 5 > $O/timeline_cfg5.txt
python bench.py --workload cfg4 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_cfg4.json 2>/dev/null
python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_cfg5.json 2>/dev/null
python bench.py --rehearse-dist --no-cpu-baseline --no-extras > $O/bench_rehearse.json 2>/dev/null || true
grep -h "ms per pass" $O/self*.log | cut -c1-420
tail -1 gpurun_out/prof_r03/bench_plain.log | cut -c1-900
