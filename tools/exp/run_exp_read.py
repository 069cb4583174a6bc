import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
so = os.path.join(here, "exp_read.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_read.hip")])
L = ctypes.CDLL(so)
L.exp_read.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
nbytes = 1 << 30
buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
out = torch.zeros(16, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
L.exp_march_pack.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p]
bits = torch.zeros(1024 * 1024 * 16, dtype=torch.int64, device=dev)
for rep in range(2):
    for ZR in (16, 32):
        for U in (2, 4, 8):
            for mode in (0, 1, 2, 3):
                t = timeit(lambda: L.exp_march_pack(buf.data_ptr(), bits.data_ptr(), 1024, 1024, ZR, U, mode, out.data_ptr(), st))
                print("march_pack ZR%-2d U%d mode%d (%s, %s): %.1f us" % (ZR, U, mode, "store" if mode & 1 else "no store", "no shuffle" if mode & 2 else "shuffle", t * 1e3), flush=True)
from tomography_3d_reconstructor_amd import pipeline
mask = buf.view(1024, 1024, 1024)
t = timeit(lambda: pipeline.pack(mask))
print("pipeline.pack: %.1f us %.2f TB/s" % (t * 1e3, nbytes / t / 1e9))
t = timeit(lambda: pipeline.pack_closed(mask))
print("pipeline.pack_closed (all launches): %.1f us" % (t * 1e3))
