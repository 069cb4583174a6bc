import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline
case, n = sys.argv[1], 1024
dev = torch.device("cuda:0"); L = _lib.lib()
mask = {"zeros": lambda: torch.zeros((n, n, n), dtype=torch.bool, device=dev), "ones": lambda: torch.ones((n, n, n), dtype=torch.bool, device=dev),
        "ellipsoid": lambda: pipeline.ellipsoid_mask(n, n, n, dev), "noise50": lambda: torch.rand((n, n, n), device=dev) < 0.5}[case]()
vol = pipeline.pack(mask.view(torch.uint8)); del mask
pad = 1
ext = torch.empty((L.tomo_ext_slices(n, pad), L.tomo_ext_rows(n, pad), L.tomo_ext_words_per_row(n, pad)), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.tomo_extend_bits(vol.bits.data_ptr(), ext.data_ptr(), n, n, n, pad, st)
data = torch.empty((n + 2, n + 2, L.tomo_field_pitch(n, pad)), dtype=torch.float32, device=dev)
signs = torch.zeros(L.tomo_sign_buffer_words(n + 2, n + 2, n + 2, L.tomo_field_xorg(pad)), dtype=torch.int64, device=dev)
gcls = torch.empty(((n + 2) * L.tomo_sign_rows(n + 2) // 16 * L.tomo_mc_segments_per_row(n + 2, L.tomo_field_xorg(pad)),), dtype=torch.uint8, device=dev)
for use in (0, 1):
    sp = signs.data_ptr() if (use & 1) else None
    gp = gcls.data_ptr() if (use & 1) else None
    for _ in range(3): L.tomo_field_fill(ext.data_ptr(), data.data_ptr(), n, n, n, pad, 1, sp, gp, st)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): L.tomo_field_fill(ext.data_ptr(), data.data_ptr(), n, n, n, pad, 1, sp, gp, st)
    b.record(); torch.cuda.synchronize()
    print(case, os.path.basename(os.environ.get("TOMO_LIB", "default")), "signs" if use else "nosigns", "%.3f ms" % (a.elapsed_time(b) / 20))
