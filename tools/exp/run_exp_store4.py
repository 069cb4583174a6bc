import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exp_store3.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_store3.hip")])
L = ctypes.CDLL(so)
L.exp_persist.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
n = 1026 * 1026 * 1088
field = torch.empty(n, dtype=torch.float32, device=dev)
cls = torch.zeros(n // 32 + 64, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for blocks in (128, 256, 512, 768, 1024, 2048):
    for mode in (0, 1, 2):
        print("persist blocks %4d mode %d (%s): %.3f ms" % (blocks, mode, ["no load", "dependent load", "prefetched load"][mode],
              timeit(lambda: L.exp_persist(field.data_ptr(), n, cls.data_ptr(), blocks, mode, st))), flush=True)
