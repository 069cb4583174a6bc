#!/bin/bash
set -e
O=gpurun_out/r3f; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/stagetest.py 2>&1 | grep -v amdgpu.ids | tee $O/stagetest.log
python tools/h2hprof.py 2>&1 | grep -v amdgpu.ids | tee $O/h2hprof.log
