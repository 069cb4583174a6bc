"""Ad-hoc fuzz: tile-sparse field fill vs dense on random shapes (NaN-poisoned buffers; meshes must be bitwise equal)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
for it in range(n):
    nz, ny, nx = int(rng.integers(1, 40)), int(rng.integers(1, 90)), int(rng.integers(1, 700))
    kind = it % 4
    if kind == 0:
        v = rng.random((nz, ny, nx)) < rng.random()
    elif kind == 1:
        v = np.zeros((nz, ny, nx), bool)
        z0, y0, x0 = rng.integers(0, nz), rng.integers(0, ny), rng.integers(0, nx)
        v[z0:z0 + rng.integers(1, nz + 1), y0:y0 + rng.integers(1, ny + 1), x0:x0 + rng.integers(1, nx + 1)] = True
    elif kind == 2:
        v = np.ones((nz, ny, nx), bool)
        v[rng.integers(0, nz), rng.integers(0, ny), rng.integers(0, nx)] = False
    else:
        zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        v = ((zz - nz / 2) / (nz * 0.45 + 1)) ** 2 + ((yy - ny / 2) / (ny * 0.4 + 1)) ** 2 + ((xx - nx / 2) / (nx * 0.42 + 1)) ** 2 <= 1
    pad = int(rng.integers(0, 2))
    vol = pipeline.pack(torch.from_numpy(np.ascontiguousarray(v).view(np.uint8)).to(dev))
    fd = pipeline.make_field(vol, True, bool(pad))
    fs = pipeline.make_field(vol, True, bool(pad), sparse=True)
    poisoned = torch.full_like(fs.data, float("nan"))
    span = torch.empty(L.tomo_field_span_bytes(nz, ny, nx, pad), dtype=torch.uint8, device=dev)
    sb = torch.empty_like(fs.signs); gc = torch.empty_like(fs.gcls)
    st = torch.cuda.current_stream().cuda_stream
    assert L.tomo_field_fill_bits_sparse(vol.bits.data_ptr(), poisoned.data_ptr(), nz, ny, nx, pad, sb.data_ptr(), gc.data_ptr(), span.data_ptr(), st) == 0
    # group classes: identical, or constant (0 / 1, no records) where the dense kernel stored records -- it classifies a tile
    # by a window that is not clipped to the array, so a tile at the array's edge can be "mixed" there and constant here
    ok = bool(((gc == fd.gcls) | ((gc < 2) & (fd.gcls == 2))).all())
    w = ~torch.isnan(poisoned)
    ok &= bool(torch.equal(poisoned[w], fd.data[w]))
    fs.data, fs.signs, fs.gcls = poisoned, sb, gc
    md, ms = pipeline.marching_cubes(fd, 0.5), pipeline.marching_cubes(fs, 0.5)
    if (md is None) != (ms is None):
        ok = False
    elif md is not None:
        ok &= bool(torch.equal(md.vkey, ms.vkey) and torch.equal(md.faces32, ms.faces32) and torch.equal(md.vpos.view(torch.int32), ms.vpos.view(torch.int32)))
    if not ok:
        bad += 1
        print("MISMATCH", (nz, ny, nx), "kind", kind, "pad", pad)
print("sparse fuzz: %d cases, %d mismatches" % (n, bad))
