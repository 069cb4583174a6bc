#!/usr/bin/env python3
"""Where does a workgroup of uq3_sortrank_kernel spend its time?  Builds a PROFILING copy of the library (-DSR_PROFILE: s_memtime
stamps per workgroup and phase, tools/scratch/libtomo_srprof.so), runs one pass on the 1024^3 ellipsoid and prints per-phase
statistics for the two variants.  usage: sortprof.py build | run"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "scratch", "libtomo_srprof.so")
CS = os.path.join(ROOT, "tomography_3d_reconstructor_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    fl = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -DSR_PROFILE".split()
    objs = []
    for f in ("bits.hip", "field.hip", "mc.hip", "mesh.hip", "volume.hip"):
        o = os.path.join(os.path.dirname(SO), f + ".srprof.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + fl + ["-c", os.path.join(CS, f), "-o", o])
        objs.append(o)
    for f in ("host_shim.cpp", "obj_writer.cpp", "host_hash.cpp"):
        o = os.path.join(os.path.dirname(SO), f + ".srprof.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-pthread", "-x", "c++", "-c",
                               os.path.join(CS, f), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", SO] + objs)
    sys.exit(0)
os.environ["TOMO_LIB"] = SO
import numpy as np, torch
sys.path.insert(0, ROOT)
from tomography_3d_reconstructor_amd import _lib, pipeline
L = _lib.lib()
L.tomo_sort_profile_buffer.argtypes = [ctypes.c_void_p]
L.tomo_sort_profile_buffer.restype = None
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
vol = pipeline.smooth(pipeline.pack_closed(pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8)), 3, True)
depths = np.full(n, 1.0)
for _ in range(3):
    pipeline.extract_surface(vol, depths, 1.0, 1.0)
nseg = int(L.tomo_mc3_sort_segments(n + 2, n + 2))
prof = torch.zeros(2 * nseg * 8, dtype=torch.int64, device=dev)
L.tomo_sort_profile_buffer(prof.data_ptr())
pipeline.extract_surface(vol, depths, 1.0, 1.0)
torch.cuda.synchronize()
L.tomo_sort_profile_buffer(None)
P = prof.cpu().numpy().reshape(2, nseg, 8)
tick = 1e-2                              # s_memtime ticks at 100 MHz on gfx9: 10 ns
for var, name in ((0, "<2048>"), (1, "<4096>")):
    Q = P[var]
    t0 = Q[:, 0][Q[:, 0] != 0].min()
    worked = Q[:, 4] != 0
    print("%s: %d workgroups, %d with work; kernel span %.1f us (first start -> last stamp)" % (
        name, (Q[:, 0] != 0).sum(), worked.sum(), (Q[:, :5].max() - t0) * tick))
    idle = Q[~worked]
    if len(idle):
        print("   idle workgroups: start..decision mean %.2f us; their starts span %.1f us" % (
            ((idle[:, 1] - idle[:, 0]) * tick).mean(), (idle[:, 0].max() - t0) * tick))
    W = Q[worked]
    if not len(W):
        continue
    names = ["prelude", "sort", "gather->LDS", "rows/table"]
    for k, nm in enumerate(names):
        d = (W[:, k + 1] - W[:, k]) * tick
        print("   %-12s mean %7.2f us  median %7.2f  p90 %7.2f  max %7.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90), d.max()))
    tot = (W[:, 4] - W[:, 0]) * tick
    print("   total        mean %7.2f us  median %7.2f  p90 %7.2f  max %7.2f; n mean %.0f max %d; starts span %.1f us; in flight on average %.0f" % (
        tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), W[:, 7].mean(), W[:, 7].max(), (W[:, 0].max() - t0) * tick,
        tot.sum() / ((W[:, 4].max() - t0) * tick)))
    big = W[np.argsort(-W[:, 7])[:3]]
    for b in big:
        print("   largest: n %d: prelude %.1f sort %.1f gather %.1f rows %.1f us" % (b[7], *[(b[k + 1] - b[k]) * tick for k in range(4)]))
