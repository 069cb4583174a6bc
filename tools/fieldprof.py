#!/usr/bin/env python3
"""Where does a workgroup of field_tile_kernel spend its time?  Builds a PROFILING copy of the library (-DFT_PROFILE: s_memtime
stamps per block and phase, tools/scratch/libtomo_prof.so), runs the dense / tile-sparse field fill on the 1024^3 ellipsoid and
prints per-phase statistics.  usage: fieldprof.py build | run [dense|sparse|near]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "scratch", "libtomo_prof.so")
CS = os.path.join(ROOT, "tomography_3d_reconstructor_amd", "csrc")
if sys.argv[1] == "build":
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    fl = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -DFT_PROFILE".split()
    objs = []
    for f in ("bits.hip", "field.hip", "mc.hip", "mesh.hip", "volume.hip"):
        o = os.path.join(os.path.dirname(SO), f + ".prof.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + fl + ["-c", os.path.join(CS, f), "-o", o])
        objs.append(o)
    for f in ("host_shim.cpp", "obj_writer.cpp", "host_hash.cpp"):
        o = os.path.join(os.path.dirname(SO), f + ".prof.o")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-pthread", "-x", "c++", "-c",
                               os.path.join(CS, f), "-o", o])
        objs.append(o)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", SO] + objs)
    sys.exit(0)
os.environ["TOMO_LIB"] = SO
mode = sys.argv[2] if len(sys.argv) > 2 else "dense"
if mode == "near":
    os.environ["TOMO_EXP_NEAR_DENSE"] = "1"
import numpy as np, torch
sys.path.insert(0, ROOT)
from tomography_3d_reconstructor_amd import _lib, pipeline
L = _lib.lib()
L.tomo_field_profile_buffer.argtypes = [ctypes.c_void_p]
L.tomo_field_profile_buffer.restype = None
n = 1024; dev = torch.device("cuda:0")
vol = pipeline.smooth(pipeline.pack_closed(pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8)), 3, True)
for _ in range(2):
    pipeline.make_field(vol, True, True, sparse=(mode != "dense"))
nb = 1 << 18
prof = torch.zeros(nb * 8, dtype=torch.int64, device=dev)
L.tomo_field_profile_buffer(prof.data_ptr())
pipeline.make_field(vol, True, True, sparse=(mode != "dense"))
torch.cuda.synchronize()
L.tomo_field_profile_buffer(None)
P = prof.cpu().numpy().reshape(nb, 8)
live = P[:, 0] != 0
P = P[live]
print(mode, "blocks that ran:", len(P))
t0 = P[:, 0].min()
done = P[:, 4] != 0                     # reached the end of the mixed phase (not an early exit)
Q = P[done]
tick = 1e-2                              # s_memtime ticks at 100 MHz on gfx9: 10 ns
names = ["stage", "reduce+classify", "const stores", "mixed", "sync", "signs"]
dur = np.stack([(Q[:, k + 1] - Q[:, k]) for k in range(6)], 1) * tick
nm = Q[:, 7] & 0xffffffff
print("working blocks %d, span of the kernel %.1f us" % (len(Q), (P[:, 1:7].max() - t0) * tick))
for k, nme in enumerate(names):
    d = dur[:, k]
    print("%-16s mean %7.2f us  median %7.2f  p90 %7.2f  max %7.2f  sum %9.0f us" % (nme, d.mean(), np.median(d), np.percentile(d, 90), d.max(), d.sum()))
tot = (Q[:, 6] - Q[:, 0]) * tick
print("block total      mean %7.2f us  median %7.2f  p90 %7.2f  max %7.2f  sum %9.0f us  -> average blocks in flight %.0f" % (
    tot.mean(), np.median(tot), np.percentile(tot, 90), tot.max(), tot.sum(), tot.sum() / ((P[:, 1:7].max() - t0) * tick)))
print("mixed tiles: total %d, per working block mean %.1f; blocks with 0: %d, 1-6: %d, 7-12: %d, 13-24: %d, >24: %d" % (
    nm.sum(), nm.mean(), (nm == 0).sum(), ((nm > 0) & (nm <= 6)).sum(), ((nm > 6) & (nm <= 12)).sum(), ((nm > 12) & (nm <= 24)).sum(), (nm > 24).sum()))
for lo, hi in ((0, 0), (1, 6), (7, 12), (13, 24), (25, 999)):
    sel = (nm >= lo) & (nm <= hi)
    if sel.any():
        print("  nmixed %3d-%3d: %6d blocks, mixed phase mean %6.2f us, block total mean %6.2f us" % (lo, hi, sel.sum(), dur[sel, 3].mean(), tot[sel].mean()))
