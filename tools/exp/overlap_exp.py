#!/usr/bin/env python3
"""Experiment (round 3): does a one-shot store stream (what would write the far-constant tiles of a dense field) overlap with
the latency-bound work of a pass -- the tile-sparse field kernel and the marching-cubes chain -- when it runs on a second
HIP stream?  Serial sum vs forked."""
import ctypes, os, subprocess, sys, time
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from tomography_3d_reconstructor_amd import pipeline
so = os.path.join(here, "exp_write2.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_write2.hip")])
L = ctypes.CDLL(so)
L.exp_row.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
n = 1024
dev = torch.device("cuda:0")
mask = pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8)
vol = pipeline.smooth(pipeline.pack_closed(mask), 3, True)
depths = np.full(n, 1.0)
far_bytes = int(0.65 * 1026 * 1026 * 1056 * 4)            # ~65 % of the field lies in blocks no surface comes near
scratch = torch.empty(far_bytes // 4, dtype=torch.float32, device=dev)
side = torch.cuda.Stream(device=dev)

def store(stream):
    L.exp_row(scratch.data_ptr(), far_bytes, 0, None, 1, ctypes.c_void_p(stream.cuda_stream))

def pass_sparse():
    pipeline.FIELD_SPARSE = True
    return pipeline.extract_surface(vol, depths, 1.0, 1.0)

def pass_dense():
    pipeline.FIELD_SPARSE = False
    return pipeline.extract_surface(vol, depths, 1.0, 1.0)

def serial():
    r = pass_sparse()
    store(torch.cuda.current_stream())
    return r

def forked():
    main = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(main)
    side.wait_event(ev)
    store(side)
    r = pass_sparse()
    ev2 = torch.cuda.Event(); ev2.record(side)
    main.wait_event(ev2)
    return r

def T(label, fn, k=12):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    print("%-64s %.3f ms" % (label, (time.perf_counter() - t0) / k * 1e3), flush=True)

T("store stream alone (%.2f GB)" % (far_bytes / 1e9), lambda: store(torch.cuda.current_stream()))
T("dense field + MC chain (today)", pass_dense)
T("sparse field + MC chain", pass_sparse)
T("sparse field + MC chain, then the store stream (serial)", serial)
T("sparse field + MC chain || store stream on a second HIP stream", forked)
T("dense field + MC chain (again)", pass_dense)
# the same with stream priorities: the pass on a HIGH priority stream, the store stream on a LOW priority one
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)
hi_s = torch.cuda.Stream(device=dev, priority=-1)
lo_s = torch.cuda.Stream(device=dev, priority=0)
def forked_prio(first_store=True):
    with torch.cuda.stream(hi_s):
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        lo_s.wait_event(ev)
        if first_store:
            store(lo_s)
            r = pass_sparse()
        else:
            r = pass_sparse()
            store(lo_s)
        ev2 = torch.cuda.Event(); ev2.record(lo_s)
        main.wait_event(ev2)
    return r
def on_hi(fn):
    with torch.cuda.stream(hi_s):
        return fn()
T("sparse + MC on a high-priority stream (alone)", lambda: on_hi(pass_sparse))
T("sparse + MC (high priority) || store stream (low priority)", forked_prio)
T("same, store stream enqueued AFTER the pass's launches", lambda: forked_prio(False))
for k in (2, 4):
    def chunks(k=k):
        with torch.cuda.stream(hi_s):
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(main); lo_s.wait_event(ev)
            per = far_bytes // k // 4096 * 4096
            for i in range(k):
                L.exp_row(scratch.data_ptr() + i * per, per, 0, None, 1, ctypes.c_void_p(lo_s.cuda_stream))
            r = pass_sparse()
            ev2 = torch.cuda.Event(); ev2.record(lo_s); main.wait_event(ev2)
        return r
    T("  ... store stream in %d launches" % k, chunks)
pipeline.FIELD_SPARSE = False
