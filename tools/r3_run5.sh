#!/bin/bash
set -e
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/morphtime.py 2>&1 | grep -v amdgpu.ids | tee $O/morph.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python tools/slab_selfloop_bench.py 1024 1024 1024 10 > $O/self1024.log 2>&1
python tools/slab_selfloop_bench.py 512 2048 2048 10 8 3 > $O/self_cfg5.log 2>&1
python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/bench_cfg5.json 2>> $O/bench.err
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_cfg5 -o x -- python3 $R/tools/slab_selfloop_bench.py 512 2048 2048 5 8 3 > $R/$O/trace_cfg5.log 2>&1
cd $R
python tools/timeline.py $O/trace_cfg5 field_tile_kernel 5 | cut -c1-110 > $O/timeline_cfg5.txt
grep -h "ms per pass" $O/self*.log | cut -c1-330
python - <<'P'
import json
for f in ("bench.json","bench_cfg5.json"):
    d=json.loads(open("gpurun_out/r3g/"+f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d.get("read_every_pass_ms_per_step"), d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("host_to_host_runs_ms"))
P
cat $O/timeline_cfg5.txt
