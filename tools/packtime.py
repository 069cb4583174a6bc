import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline
n = 1024; dev = torch.device("cuda:0")
m = pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8)
for _ in range(3): pipeline.pack(m)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): v = pipeline.pack(m)
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20
print(os.path.basename(os.environ.get("TOMO_LIB", "default")), "pack %.3f ms  %.0f GB/s" % (t, n ** 3 / t / 1e6))
