#!/usr/bin/env python3
"""np.stack of 1024 separate 1 MiB masks -> device: variants of the host staging (which memory, who brings its pages in)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline
L = _lib.lib()
n = 1024
dev = torch.device("cuda:0")
stack = pipeline.ellipsoid_mask(n, n, n, dev).cpu().numpy()
masks = [stack[i].copy() for i in range(n)]
del stack
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
d = torch.empty((n, n, n), dtype=torch.bool, device=dev)
torch.cuda.synchronize()
def T(label, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print("%-70s %7.1f ms" % (label, (time.perf_counter() - t0) * 1e3), flush=True); return r
def fill(stage, lo, hi):
    for i in range(lo, hi):
        stage[i] = masks[i]
def fill_all(stage, workers, fine=16):
    with ThreadPoolExecutor(workers) as ex:
        list(ex.map(lambda lo: fill(stage, lo, min(n, lo + fine)), range(0, n, fine)))
def pipelined(stage, workers, up=128, fine=16):
    with ThreadPoolExecutor(workers) as ex:
        pieces = [(lo, min(n, lo + up), [ex.submit(fill, stage, a, min(min(n, lo + up), a + fine)) for a in range(lo, min(n, lo + up), fine)]) for lo in range(0, n, up)]
        for lo, hi, futs in pieces:
            for f in futs: f.result()
            d[lo:hi].copy_(torch.from_numpy(stage[lo:hi]))
for workers in (8, 16):
    s = np.empty((n, n, n), np.bool_)
    T("fill only, fresh np.empty, %d workers" % workers, lambda: fill_all(s, workers))
    T("fill only again (pages present), %d workers" % workers, lambda: fill_all(s, workers))
    del s
s = np.empty((n, n, n), np.bool_)
T("touch 16 threads", lambda: L.tomo_host_touch(s.ctypes.data, s.nbytes, 16))
T("fill only after touch, 8 workers", lambda: fill_all(s, 8))
T("upload only, 8 pieces, from that array", lambda: [d[lo:lo + 128].copy_(torch.from_numpy(s[lo:lo + 128])) for lo in range(0, n, 128)])
T("upload only, 1 piece", lambda: d.copy_(torch.from_numpy(s)))
del s
for workers in (8, 16):
    s = np.empty((n, n, n), np.bool_)
    T("V1 pipelined fill + upload, fresh np.empty, %d workers" % workers, lambda: pipelined(s, workers))
    T("V1 again (pages present)", lambda: pipelined(s, workers))
    del s
s = np.empty((n, n, n), np.bool_)
def v2():
    L.tomo_host_touch(s.ctypes.data, s.nbytes, 16); pipelined(s, 8)
T("V2 touch(16) then pipelined fill + upload, 8 workers", v2)
del s
s = np.empty((n, n, n), np.bool_)
T("V3 fill all (8 workers) then ONE upload, fresh", lambda: (fill_all(s, 8), d.copy_(torch.from_numpy(s))))
del s
def v5():
    st = torch.empty((n, n, n), dtype=torch.bool, pin_memory=True); a = st.numpy()
    with ThreadPoolExecutor(8) as ex:
        futs = [(lo, min(n, lo + 32), ex.submit(fill, a, lo, min(n, lo + 32))) for lo in range(0, n, 32)]
        for lo, hi, f in futs:
            f.result(); d[lo:hi].copy_(st[lo:hi], non_blocking=True)
    return st
st = T("V5 page-locked staging (rounds 1-2), cold", v5); del st
st = T("V5 warm", v5); del st
# single big memcpy-style: np.stack
T("np.stack(masks) alone (one thread)", lambda: np.stack(masks))
