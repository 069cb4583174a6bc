#!/usr/bin/env python3
"""D2H speed into pageable memory as a function of WHO brought its pages in (threads, NUMA placement)."""
import os, sys, time, glob
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib
L = _lib.lib()
GiB = 1 << 30
dev = torch.device("cuda:0")
x = torch.empty(GiB, dtype=torch.uint8, device=dev); x.fill_(1); torch.cuda.synchronize()
def timed(label, fn):
    t0 = time.perf_counter(); fn(); dt = (time.perf_counter() - t0) * 1e3
    print("%-64s %7.1f ms" % (label, dt), flush=True)
nodes = {}
for p in glob.glob("/sys/devices/system/node/node*/cpulist"):
    nodes[int(p.split("node")[-1].split("/")[0])] = open(p).read().strip()
print("numa nodes:", nodes, "| cpu now:", os.sched_getaffinity(0).__len__(), "cpus allowed")
try:
    import subprocess
    print(subprocess.run("cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -8 | tr '\\n' ' '; cat /sys/kernel/mm/transparent_hugepage/enabled", shell=True, capture_output=True, text=True).stdout)
except Exception as e:
    print(e)
def cpus_of(s):
    out = set()
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-"); out |= set(range(int(a), int(b) + 1))
        elif part:
            out.add(int(part))
    return out
warm = torch.from_numpy(np.ones(1 << 20, np.uint8)); warm.copy_(x[:1 << 20]); torch.cuda.synchronize()
full = os.sched_getaffinity(0)
for label, aff in [("all allowed cpus", full)] + [("node %d cpus" % n, cpus_of(s) & full) for n, s in sorted(nodes.items())]:
    if not aff:
        continue
    os.sched_setaffinity(0, aff)
    for nt in (1, 16):
        a = np.empty(GiB, dtype=np.uint8)
        timed("[%s] touch %2d threads" % (label, nt), lambda: L.tomo_host_touch(a.ctypes.data, GiB, nt))
        ap = torch.from_numpy(a)
        timed("[%s]   D2H 1 GiB into it" % label, lambda: (ap.copy_(x), torch.cuda.synchronize()))
        timed("[%s]   D2H again" % label, lambda: (ap.copy_(x), torch.cuda.synchronize()))
        del a, ap
    os.sched_setaffinity(0, full)
b = np.ones(GiB, np.uint8); bp = torch.from_numpy(b)
timed("np.ones then D2H", lambda: (bp.copy_(x), torch.cuda.synchronize()))
c = np.empty(GiB, np.uint8); cp = torch.from_numpy(c)
timed("np.empty (untouched) then D2H", lambda: (cp.copy_(x), torch.cuda.synchronize()))
timed("  again", lambda: (cp.copy_(x), torch.cuda.synchronize()))
