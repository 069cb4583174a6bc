"""How launch-bound is a pass on a small stack?  ms per pass (pipelined submission, one stream) and host time to enqueue one."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tomography_3d_reconstructor_amd import pipeline
dev = torch.device("cuda:0")
for shape in ((64, 128, 128), (128, 256, 256), (256, 512, 512)):
    nz, ny, nx = shape
    mask = pipeline.ellipsoid_mask(nz, ny, nx, dev).view(torch.uint8)
    d = np.full(nz, 1.0)
    for _ in range(5):
        bench.one_pass_submit(mask, d, False).result()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter(); pend = None; tenq = 0.0
    for _ in range(K):
        a = time.perf_counter(); nxt = bench.one_pass_submit(mask, d, False); tenq += time.perf_counter() - a
        if pend is not None: pend.result()
        pend = nxt
    pend.result(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K * 1e3
    print("%-16s %.3f ms per pass (%.0f Mvoxels/s), host enqueue %.3f ms per pass" % (shape, dt, nz * ny * nx / dt / 1e3, tenq / K * 1e3))
