"""A few whole passes of the hot path on the 1024^3 ellipsoid (driver for counter collection: tools/pmc_kernel.sh)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from tomography_3d_reconstructor_amd import pipeline
n = 1024
dev = torch.device("cuda:0")
mask = pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8)
depths = np.full(n, 1.0)
for _ in range(4):
    res = bench.one_pass(mask, depths)
torch.cuda.synchronize()
print(res[0].shape, res[1].shape)
