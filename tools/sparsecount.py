import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline, _lib
n = 1024; dev = torch.device("cuda:0"); L = _lib.lib()
vol = pipeline.pack(pipeline.ellipsoid_mask(n, n, n, dev).view(torch.uint8))
f = pipeline.make_field(vol, True, True)
span = torch.zeros(L.tomo_field_span_bytes(n, n, n, 1), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
data = torch.full_like(f.data, float("nan"))
assert L.tomo_field_fill_bits_sparse(vol.bits.data_ptr(), data.data_ptr(), n, n, n, 1, f.signs.data_ptr(), f.gcls.data_ptr(), span.data_ptr(), st) == 0
torch.cuda.synchronize()
Nz = n + 2; ntr = (n + 2 + 15) // 16; NT = L.tomo_field_pitch(n, 1) // 32
tiles = Nz * ntr * NT; tal = (tiles + 63) // 64 * 64
comb = span[tal:tal + tiles].view(Nz, ntr, NT)
cnt = int(span[2 * tal:2 * tal + 4].view(torch.int32)[0])
nxc = (NT + 7 + 7) // 8; nzg = (Nz + 3) // 4
print("tiles", tiles, "comb==3: %.2f%%" % (100 * float((comb == 3).float().mean())), "comb==1 %.2f%% comb==2 %.2f%% comb==0 %.2f%%" % tuple(100 * float((comb == k).float().mean()) for k in (1, 2, 0)))
print("blocks", nxc * ntr * nzg, "listed", cnt, "%.1f%%" % (100.0 * cnt / (nxc * ntr * nzg)))
print("written floats: %.2f%%" % (100 * float((~torch.isnan(data)).float().mean())))
