"""Host checksum rate on this box: AVX2 against the portable loop, 1 / 16 / 32 threads, 1 GiB."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib
L = _lib.lib()
a = np.ones(1 << 30, dtype=np.uint8)
for impl in (0, 1):
    for nt in (1, 8, 16, 32):
        o = (ctypes.c_uint64 * 2)()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); L.tomo_host_checksum_impl(a.ctypes.data, a.nbytes, nt, impl, o); best = min(best, time.perf_counter() - t)
        print("impl %d threads %2d: %.1f ms per GiB" % (impl, nt, best * 1e3))
