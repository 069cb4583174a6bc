import ctypes, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "exp_write2.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(here, "exp_write2.hip")])
L = ctypes.CDLL(so)
L.exp_row.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
nbytes = 1026 * 1026 * 1056 * 4
buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
nwg = (nbytes // 16 + 255) // 256
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def show(name, t, frac=1.0): print("%-44s %.3f ms  %.2f TB/s" % (name, t, frac * nbytes / t / 1e9), flush=True)
show("memset", timeit(lambda: buf.zero_()))
for per in (1, 64):
    for frac_on in (1.0, 0.8):
        n_ent = (nwg + per - 1) // per
        m8 = (torch.rand(n_ent * 32, device=dev) < frac_on).to(torch.uint8) * 2        # 0 = skip, 2 = ones
        m8w = m8.view(n_ent, 32)[:, :1].expand(n_ent, 32).contiguous()                 # per-workgroup-uniform variant
        w = torch.zeros((n_ent, 2), dtype=torch.int64, device=dev)
        w[:, 0] = torch.where(m8w[:, 0] != 0, torch.tensor(-1, dtype=torch.int64, device=dev), torch.tensor(0, dtype=torch.int64, device=dev))
        w[:, 1] = -1
        on = float((m8w[:, 0] != 0).float().mean().item())
        show("mode0 no load", timeit(lambda: L.exp_row(buf.data_ptr(), nbytes, 0, None, per, st)))
        show("mode1 scalar byte  per=%d on=%.2f" % (per, on), timeit(lambda: L.exp_row(buf.data_ptr(), nbytes, 1, m8w[:, 0].contiguous().data_ptr(), per, st)), on)
        show("mode2 vector byte  per=%d on=%.2f" % (per, on), timeit(lambda: L.exp_row(buf.data_ptr(), nbytes, 2, m8w.data_ptr(), per, st)), on)
        show("mode3 scalar 2xu64 per=%d on=%.2f" % (per, on), timeit(lambda: L.exp_row(buf.data_ptr(), nbytes, 3, w.data_ptr(), per, st)), on)
