#!/usr/bin/env python3
"""Throughput with several stacks in flight on ONE GPU: T host threads, each with its own HIP stream, each running whole
passes of the hot path on its own 1024^3 ellipsoid stack.  (bench.py's headline runs the passes one after the other.)
usage: inflightbench.py [threads] [passes_per_thread]"""
import os, sys, threading, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tomography_3d_reconstructor_amd import pipeline
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1024
dev = torch.device("cuda:0")
masks = [pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8) for _ in range(T)]
depths = np.full(n, 1.0)
for m in masks:
    for _ in range(3): bench.one_pass(m, depths)
torch.cuda.synchronize()
bar = threading.Barrier(T + 1)
def worker(i):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        for _ in range(2): bench.one_pass(masks[i], depths)
        s.synchronize()
        bar.wait()
        for _ in range(K): res = bench.one_pass(masks[i], depths)
        s.synchronize()
    bar.wait()
th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
for t in th: t.start()
bar.wait(); t0 = time.perf_counter()
bar.wait(); dt = time.perf_counter() - t0
for t in th: t.join()
print("%d stacks in flight: %.3f ms per pass, %.0f Mvoxels/s" % (T, dt / (T * K) * 1e3, n ** 3 * T * K / dt / 1e6))
