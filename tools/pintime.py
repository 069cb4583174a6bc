#!/usr/bin/env python3
"""How long does page-locking host memory take on this box, and does it run in parallel?  (cold host-to-host call)"""
import sys, time, threading
import torch
torch.cuda.init()
GiB = 1 << 30
def pin(n):
    t = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    return t
def timed(label, fn):
    t0 = time.perf_counter(); r = fn(); dt = (time.perf_counter() - t0) * 1e3
    print("%-52s %7.1f ms" % (label, dt), flush=True); return r
a = timed("1 GiB, one call", lambda: pin(GiB))
b = timed("1 GiB again (new block)", lambda: pin(GiB))
def par(k, n):
    out = [None] * k
    th = [threading.Thread(target=lambda i=i: out.__setitem__(i, pin(n))) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
    return out
c = timed("4 x 1 GiB on 4 threads", lambda: par(4, GiB))
d = timed("8 x 128 MiB on 8 threads", lambda: par(8, GiB // 8))
e = timed("16 x 64 MiB on 16 threads", lambda: par(16, GiB // 16))
f = timed("8 x 128 MiB, one thread", lambda: [pin(GiB // 8) for _ in range(8)])
del a
g = timed("1 GiB from torch's pinned cache", lambda: pin(GiB))
dev = torch.device("cuda:0")
x = torch.empty(GiB, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
def d2h():
    g.copy_(x, non_blocking=True); torch.cuda.synchronize()
timed("D2H 1 GiB into pinned", d2h); timed("D2H 1 GiB into pinned (again)", d2h)
def h2d():
    x.copy_(g, non_blocking=True); torch.cuda.synchronize()
timed("H2D 1 GiB from pinned", h2d)
import numpy as np
h = timed("np.empty 1 GiB + touch", lambda: np.ones(GiB, dtype=np.uint8))
hp = torch.from_numpy(h)
def d2h_pageable():
    hp.copy_(x); torch.cuda.synchronize()
timed("D2H 1 GiB into pageable", d2h_pageable)
try:
    rt = torch.cuda.cudart()
    r = timed("cudaHostRegister 1 GiB of touched pageable", lambda: rt.cudaHostRegister(h.ctypes.data, GiB, 0))
    print("register rc", r)
    timed("D2H 1 GiB into registered", d2h_pageable)
except Exception as ex:
    print("hostRegister failed", ex)
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib
L = _lib.lib()
for nt in (1, 4, 16, 32):
    a = np.empty(GiB, dtype=np.uint8)
    timed("np.empty 1 GiB + tomo_host_touch, %d threads" % nt, lambda: L.tomo_host_touch(a.ctypes.data, GiB, nt))
    if nt == 16:
        ap = torch.from_numpy(a)
        timed("D2H 1 GiB into that array", lambda: (ap.copy_(x), torch.cuda.synchronize()))
        timed("H2D 1 GiB from that array (pageable)", lambda: (x.copy_(ap), torch.cuda.synchronize()))
        timed("H2D again", lambda: (x.copy_(ap), torch.cuda.synchronize()))
        for k in (8,):
            parts = [(i * GiB // k, (i + 1) * GiB // k) for i in range(k)]
            timed("H2D 1 GiB pageable in %d chunks" % k, lambda: ([x[lo:hi].copy_(ap[lo:hi]) for lo, hi in parts], torch.cuda.synchronize()))
    del a
