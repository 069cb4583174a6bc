import ctypes, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exp_store.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_store.hip")])
L = ctypes.CDLL(so)
L.exp_store.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
n = 1024; dev = torch.device("cuda:0")
Nz = Ny = n + 2; pitch = 1056; EY = n + 6; EWX32 = 34
ext = torch.zeros((n + 6) * EY * EWX32, dtype=torch.int32, device=dev)
field = torch.empty(Nz * Ny * pitch, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("memset %.3f ms" % timeit(lambda: field.zero_()))
pitch = 1088
field = torch.empty(Nz * Ny * pitch + 64, dtype=torch.float32, device=dev)
sig = torch.zeros(Nz * Ny * 5 * 4 + 64, dtype=torch.int64, device=dev)
for mode in (0, 3, 8, 11):
    t = timeit(lambda: L.exp_store(ext.data_ptr(), field.data_ptr(), Nz, Ny, pitch, EY, EWX32, mode, 32, 32, 24480, sig.data_ptr(), st))
    print("aligned mode %d: %.3f ms" % (mode, t))
