#!/usr/bin/env python3
"""Per-kernel timing (HIP events on the launch stream) for tuning: python tools/kbench.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import _lib, pipeline  # noqa: E402


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda:0")
    L = _lib.lib()
    Np = (n + 2) ** 3
    cases = {
        "zeros": torch.zeros((n, n, n), dtype=torch.bool, device=dev),
        "ones": torch.ones((n, n, n), dtype=torch.bool, device=dev),
        "ellipsoid": pipeline.ellipsoid_mask(n, n, n, dev),
        "noise50": torch.rand((n, n, n), device=dev) < 0.5,
    }
    for name, mask in cases.items():
        m8 = mask.view(torch.uint8)
        t_pack = timeit(lambda: pipeline.pack(m8))
        vol = pipeline.pack(m8)
        del mask
        nz, ny, nx = vol.shape
        pad = 1
        ez, ey, ewx = L.tomo_ext_slices(nz, pad), L.tomo_ext_rows(ny, pad), L.tomo_ext_words_per_row(nx, pad)
        ext = torch.empty((ez, ey, ewx), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        t_ext = timeit(lambda: L.tomo_extend_bits(vol.bits.data_ptr(), ext.data_ptr(), nz, ny, nx, pad, st))
        pitch = L.tomo_field_pitch(nx, pad)
        data = torch.empty((nz + 2, ny + 2, pitch), dtype=torch.float32, device=dev)
        t_zero = timeit(lambda: data.zero_())
        signs = torch.zeros(L.tomo_sign_buffer_words(nz + 2, ny + 2, nx + 2, L.tomo_field_xorg(pad)), dtype=torch.int64, device=dev)
        gcls = torch.empty(((nz + 2) * L.tomo_sign_rows(ny + 2) // 16 * L.tomo_mc_segments_per_row(nx + 2, L.tomo_field_xorg(pad)),), dtype=torch.uint8, device=dev)
        t_field = timeit(lambda: L.tomo_field_fill(ext.data_ptr(), data.data_ptr(), nz, ny, nx, pad, 1, signs.data_ptr(), gcls.data_ptr(), st))
        t_fin = 0.0
        f = pipeline.Field(data, nz + 2, ny + 2, nx + 2, pitch, L.tomo_field_xorg(pad), signs[: (nz + 2) * L.tomo_mc_segments_per_row(nx + 2, L.tomo_field_xorg(pad)) * L.tomo_sign_rows(ny + 2) * 4].view(nz + 2, -1, L.tomo_sign_rows(ny + 2), 4), 0.5, gcls.view(nz + 2, L.tomo_sign_rows(ny + 2) // 16, -1))
        spr = L.tomo_mc_segments_per_row(f.Nx, f.xorg)
        nseg = f.Nz * f.Ny * spr
        seg_act = torch.empty(nseg * 4, dtype=torch.int64, device=dev)
        seg_cnt = torch.empty(nseg, dtype=torch.int32, device=dev)
        t_cls = timeit(lambda: L.tomo_mc_classify(signs.data_ptr(), gcls.data_ptr(), f.Nz, f.Ny, f.Nx, f.xorg, seg_act.data_ptr(), seg_cnt.data_ptr(), st))
        t_morph = timeit(lambda: pipeline.smooth(vol, 3, True))
        t_close = timeit(lambda: pipeline.close_ends(vol))
        if name != "noise50" or n <= 256:
            t_mc = timeit(lambda: pipeline.marching_cubes(f, 0.5), n=3, warm=1)
        else:
            t_mc = float("nan")
        print("%-10s n=%d memset %.3f ms (%.0f GB/s) pack %.3f ms (%.0f GB/s) | ext %.3f | field %.3f ms (%.0f GB/s alg) | classify %.3f ms (%.0f GB/s) | "
              "smooth %.3f | close %.3f | mc_total %.3f" % (
                  name, n, t_zero, data.numel() * 4 / t_zero / 1e6, t_pack, n ** 3 / t_pack / 1e6, t_ext, t_field, 5 * Np / t_field / 1e6, t_cls,
                  4 * Np / t_cls / 1e6, t_morph, t_close, t_mc), flush=True)
        del vol, ext, data, f, seg_act, signs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
