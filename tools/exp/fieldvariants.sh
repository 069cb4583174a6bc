#!/bin/bash
# experiment: field kernel time for build variants in tools/scratch/libtomo_<tag>.so against the default build
for r in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then L=""; else L="TOMO_LIB=$PWD/tools/scratch/libtomo_$v.so"; fi
  env $L python bench.py --no-cpu-baseline --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s' % '$v', 'field kernel', d['roofline']['kernel_ms'], 'ms; pass', d['ms_per_step'], 'ms; mesh', d['config']['n_vertices'], d['config']['n_faces'])"
done; done
