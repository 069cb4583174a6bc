import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
dev = torch.device("cuda:0")
Nz = Ny = 1026; pitch = 1088
field = torch.empty(Nz * Ny * pitch, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("memset %.3f ms" % timeit(lambda: field.zero_()), flush=True)
print("fill(1.0) %.3f ms" % timeit(lambda: field.fill_(1.0)), flush=True)
so = os.path.join(here, "exp_store2_0.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-DEXPNAME=exp_store2", "-o", so, os.path.join(here, "exp_store2.hip")])
L = ctypes.CDLL(so)
L.exp_store2.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
import ctypes as C
hip = C.CDLL("libamdhip64.so")
for (zgn, rows) in ((4, 16), (1, 16), (1, 4), (1, 1)):
    for mode in (0, 3):
        for lds in (0, 20000, 30000, 40000, 52000, 65000, 80000, 160000):
            t = timeit(lambda: L.exp_store2(field.data_ptr(), Nz, Ny, pitch, zgn, rows, 0.0, 0, lds, mode, st))
            print("zg %d rows %2d mode %d lds %6d : %.3f ms" % (zgn, rows, mode, lds, t), flush=True)
