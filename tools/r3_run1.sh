#!/bin/bash
# one GPU call: tests, bench (pipelined / read-every-pass), slab rehearsal both ways, kernel trace of the rehearsal
set -e
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python bench.py --no-cpu-baseline --no-extras --read-every-pass > $O/bench_serial.json 2>> $O/bench.err
python tools/slab_selfloop_bench.py 1024 1024 1024 10 > $O/self1024.log 2>&1
TOMO_READ_EVERY_PASS=1 python tools/slab_selfloop_bench.py 1024 1024 1024 10 > $O/self1024_serial.log 2>&1
python tools/slab_selfloop_bench.py 512 2048 2048 10 > $O/self_cfg5.log 2>&1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -o x -- python3 $R/tools/slab_selfloop_bench.py 1024 1024 1024 6 > $R/$O/trace.log 2>&1
cd $R
python tools/timeline.py $O/trace field_tile_kernel -9 > $O/timeline.txt
grep -h "ms per pass" $O/self*.log | cut -c1-400
python - <<'P'
import json
for f in ("bench.json","bench_serial.json"):
    d=json.loads(open("gpurun_out/r3b/"+f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d.get("read_every_pass_ms_per_step"), d["roofline"]["kernel_ms"], d.get("host_to_host_runs_ms"))
P
