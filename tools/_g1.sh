set -x
mkdir -p gpurun_out/r2a
python tools/rccl_selfloop.py > gpurun_out/r2a/selfloop.log 2>&1; echo "selfloop rc $?" >> gpurun_out/r2a/selfloop.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r2a/bench1.log 2>&1; echo "rc $?" >> gpurun_out/r2a/bench1.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --size 256 512 512 --steps 5 > gpurun_out/r2a/bench2gloo.log 2>&1; echo "rc $?" >> gpurun_out/r2a/bench2gloo.log
timeout -k 10 300 python bench.py --workload cfg4 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2a/bench_cfg4.log 2>&1; echo "rc $?" >> gpurun_out/r2a/bench_cfg4.log
tail -3 gpurun_out/r2a/*.log
