#!/usr/bin/env python3
"""Run only the field ("SDF") kernel (and optionally mc_classify) on one input, for rocprofv3 --pmc passes.
usage: fieldbench.py CASE N [reps] ; CASE in zeros|ones|ellipsoid|noise50"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline  # noqa: E402

case, n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
L = _lib.lib()
if case == "zeros":
    mask = torch.zeros((n, n, n), dtype=torch.bool, device=dev)
elif case == "ones":
    mask = torch.ones((n, n, n), dtype=torch.bool, device=dev)
elif case == "ellipsoid":
    mask = pipeline.ellipsoid_mask(n, n, n, dev)
else:
    mask = torch.rand((n, n, n), device=dev) < 0.5
vol = pipeline.pack(mask.view(torch.uint8))
del mask
pad = 1
ext = torch.empty((L.tomo_ext_slices(n, pad), L.tomo_ext_rows(n, pad), L.tomo_ext_words_per_row(n, pad)), dtype=torch.int64,
                  device=dev)
st = torch.cuda.current_stream().cuda_stream
L.tomo_extend_bits(vol.bits.data_ptr(), ext.data_ptr(), n, n, n, pad, st)
pitch = L.tomo_field_pitch(n, pad)
data = torch.empty((n + 2, n + 2, pitch), dtype=torch.float32, device=dev)
spr = L.tomo_mc_segments_per_row(n + 2, L.tomo_field_xorg(pad))
nseg = (n + 2) * (n + 2) * spr
signs = torch.zeros(L.tomo_sign_buffer_words(n + 2, n + 2, n + 2, L.tomo_field_xorg(pad)), dtype=torch.int64, device=dev)
gcls = torch.empty(((n + 2) * L.tomo_sign_rows(n + 2) // 16 * spr,), dtype=torch.uint8, device=dev)
seg_act = torch.empty(nseg * 4, dtype=torch.int64, device=dev)
seg_cnt = torch.empty(nseg, dtype=torch.int32, device=dev)
for _ in range(reps):
    L.tomo_field_fill(ext.data_ptr(), data.data_ptr(), n, n, n, pad, 1, signs.data_ptr(), gcls.data_ptr(), st)
    L.tomo_mc_classify(signs.data_ptr(), gcls.data_ptr(), n + 2, n + 2, n + 2, L.tomo_field_xorg(pad), seg_act.data_ptr(), seg_cnt.data_ptr(), st)
torch.cuda.synchronize()
print("done", case, n)
