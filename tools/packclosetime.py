"""Times pipeline.pack_closed (np.stack + close ends in one pass over the mask) on the 1024^3 ellipsoid; TOMO_LIB selects the build."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
m = pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8)
for _ in range(3): pipeline.pack_closed(m)
torch.cuda.synchronize()
a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): v = pipeline.pack_closed(m)
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20
print("%-22s pack_closed %.3f ms  %.0f GB/s of mask" % (os.path.basename(os.environ.get("TOMO_LIB", "default")), t, n ** 3 / t / 1e6), "checksum", int(v.bits.sum().item()))
