import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exp_store3.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_store3.hip")])
L = ctypes.CDLL(so)
L.exp_fill.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
n = 1026 * 1026 * 1088
field = torch.empty(n, dtype=torch.float32, device=dev)
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
L.exp_persist.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L.exp_page_dep.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
cls = torch.zeros(n // 32 + 64, dtype=torch.uint8, device=dev)
print("linear T256 U1: %.3f ms" % timeit(lambda: L.exp_fill(field.data_ptr(), n, 256, 1, 0, st)), flush=True)
print("page + dependent class load: %.3f ms" % timeit(lambda: L.exp_page_dep(field.data_ptr(), n, cls.data_ptr(), st)), flush=True)
L.exp_page_sdep.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
for rep in range(3):
    print("page + SCALAR class load: %.3f ms" % timeit(lambda: L.exp_page_sdep(field.data_ptr(), n, cls.data_ptr(), st)), flush=True)
    print("linear T256 U1: %.3f ms" % timeit(lambda: L.exp_fill(field.data_ptr(), n, 256, 1, 0, st)), flush=True)
