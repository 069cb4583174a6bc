import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mask = pipeline.ellipsoid_mask(64, n, n, dev)
rng = torch.Generator(device="cpu"); 
mask ^= (torch.rand(mask.shape, device=dev) < 0.01)
vol = pipeline.pack(mask.view(torch.uint8))
f = pipeline.make_field(vol)
assert f.signs is not None
fused = f.signs.clone()
f2 = pipeline.Field(f.data, f.Nz, f.Ny, f.Nx, f.pitch, f.xorg)
gen = pipeline.field_signs(f2, 0.5).clone()
torch.cuda.synchronize()
L = _lib.lib()
S = fused.shape[1]
# compare only bits of valid voxels
fz = fused.cpu().numpy().view(np.uint64); gz = gen.cpu().numpy().view(np.uint64)
Nz, S, NyP, _ = fz.shape
bad = 0
for s in range(S):
    for k in range(4):
        Ls = np.arange(64)
        X = 256 * s - 224 + 4 * Ls + k - f.xorg
        valid = (X >= 0) & (X < f.Nx)
        m = np.uint64(0)
        for L_ in Ls[valid]: m |= np.uint64(1) << np.uint64(L_)
        d = (fz[:, s, :f.Ny, k] ^ gz[:, s, :f.Ny, k]) & m
        nb = int(np.count_nonzero(d))
        if nb:
            zz, yy = np.nonzero(d)
            print("seg", s, "k", k, "mismatching records", nb, "first (Z,Y):", list(zip(zz[:5], yy[:5])), "xor", [hex(int(x)) for x in d[zz[:3], yy[:3]]])
            bad += nb
print("total mismatching records", bad)
