#!/bin/bash
set -e
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/exp/run_exp_write2.py > $O/exp_write2.log 2>&1
cat $O/exp_write2.log
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_sparse -o x -- python3 $R/tools/sparsetime.py > $R/$O/sparsetime.log 2>&1
TOMO_EXP_NEAR_DENSE=1 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_near -o x -- python3 $R/tools/sparsetime.py > $R/$O/sparsetime_near.log 2>&1
for tp in 8 32; do TOMO_SPARSE_TP=$tp TOMO_EXP_NEAR_DENSE=1 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_near$tp -o x -- python3 $R/tools/sparsetime.py > $R/$O/sparsetime_near$tp.log 2>&1; done
cd $R
for d in tr_sparse tr_near tr_near8 tr_near32; do echo == $d; python tools/ktrace.py $O/$d field_; done
cat $O/sparsetime.log $O/sparsetime_near.log | grep field
python tools/slab_selfloop_bench.py 1024 1024 1024 10 2>&1 | grep "ms per pass" | cut -c1-300
