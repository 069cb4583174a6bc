#!/bin/bash
set -e
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/pintime.py 2>&1 | grep -v amdgpu.ids | tee $O/pintime.log | tail -12
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r3e/bench.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "serial", d.get("read_every_pass_ms_per_step"), "roofline", {k: d["roofline"][k] for k in ("achieved","frac","frac_survey","kernel_ms")})
print("h2h", d.get("host_to_host_runs_ms"))
P
