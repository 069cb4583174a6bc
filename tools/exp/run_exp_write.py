import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exp_write.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_write.hip")])
L = ctypes.CDLL(so)
L.exp_write.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
nbytes = 1026 * 1026 * 1088 * 4
buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def show(name, t): print("%-34s %.3f ms  %.2f TB/s" % (name, t, nbytes / t / 1e9), flush=True)
show("memset (zero_)", timeit(lambda: buf.zero_()))
show("torch fill_(1.0)", timeit(lambda: buf.fill_(1.0)))
for nt in (0, 1):
    for U in (1, 2, 4, 8):
        show("chunk U%d nt%d" % (U, nt), timeit(lambda: L.exp_write(buf.data_ptr(), nbytes, 0, U, nt, 0, st)))
    for blocks in (256, 512, 1024, 2048, 4096, 16384):
        show("persist blocks %d nt%d" % (blocks, nt), timeit(lambda: L.exp_write(buf.data_ptr(), nbytes, 1, 0, nt, blocks, st)))
    show("wave pages nt%d" % nt, timeit(lambda: L.exp_write(buf.data_ptr(), nbytes, 2, 0, nt, 0, st)))
