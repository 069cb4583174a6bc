#!/usr/bin/env python3
"""bench.py -- throughput of the hot path (mask stack -> field -> marching cubes -> final mesh).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path over one synthetic ellipsoid mask stack that is already
resident in HBM (uint8, 1 B/voxel): pack -> close ends -> opening + 3 closings -> field ("SDF") fill ->
Lewiner marching cubes (count/scan/emit) -> vertex finalisation -> unique/remap.  Outputs stay in HBM.
N = 1: the 1024^3 ellipsoid of BASELINE.json (configs[2]).  N > 1: Z-slab sharding, 1024 slices per rank
(weak scaling), one-slice field halo and bit-volume halos over RCCL (tomography_3d_reconstructor_amd/slab.py).

Prints ONE JSON line (rank 0) with metric/value/unit..., plus
  roofline:      the field ("SDF") kernel, algorithmic 5 B per padded voxel / measured kernel time (HIP events
                 on the launch stream, inside the timed region) vs the 8 TB/s HBM3E peak;
  cpu_baseline:  the CPU oracle (a C/NumPy restatement of the reference, single thread) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from tomography_3d_reconstructor_amd import _lib, pipeline  # noqa: E402

HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, nargs=3, default=None, metavar=("NZ", "NY", "NX"),
                    help="per-rank volume (default 1024 1024 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1; gloo (halos staged through the host) only to rehearse the "
                         "multi-rank code path with several ranks on ONE GPU -- never a measurement")
    ap.add_argument("--sparse-field", action="store_true",
                    help="opt-in: do not materialise the parts of the float field that marching cubes cannot read "
                         "(same mesh; NOT the headline configuration -- the roofline entry then only carries a note)")
    ap.add_argument("--cpu-sample", type=int, default=768, help="edge of the cube the CPU oracle is timed on")
    return ap.parse_args()


class FieldTimer:
    """HIP events around every field-kernel launch (tomo_field_fill / tomo_field_fill_bits; torch events on the launch stream)."""

    def __init__(self):
        self.pairs = []
        self.padded_voxels = 0
        self.enabled = False
        self._orig = None

    def install(self):
        L = _lib.lib()
        timer = self

        def wrap(orig):
            def wrapped(*a):
                if not timer.enabled:
                    return orig(*a)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = orig(*a)
                e1.record()
                timer.pairs.append((e0, e1))
                nz, ny, nx, pad = a[2], a[3], a[4], a[5]
                timer.padded_voxels = (nz + 2 * pad) * (ny + 2 * pad) * (nx + 2 * pad)
                return rc
            return wrapped

        # ctypes function objects are attributes of the CDLL instance; both entry points launch the field kernel
        L.tomo_field_fill = wrap(L.tomo_field_fill)
        L.tomo_field_fill_bits = wrap(L.tomo_field_fill_bits)
        L.tomo_field_fill_bits_sparse = wrap(L.tomo_field_fill_bits_sparse)

    def mean_ms(self):
        if not self.pairs:
            return None
        return float(np.mean([a.elapsed_time(b) for a, b in self.pairs]))


def one_pass(mask, depths):
    vol = pipeline.pack(mask)
    vol = pipeline.close_ends(vol, inplace=True)       # the packed copy is this pass's own
    vol = pipeline.smooth(vol, 3, True)
    return pipeline.extract_surface(vol, depths, 1.0, 1.0, True, True)


def cpu_baseline(n):
    from oracle import oracle as O
    masks = O.ellipsoid_masks(n, n, n)
    vp, se = O.VoxelProcessor(), O.SurfaceExtractor()
    t0 = time.perf_counter()
    created = vp.create_voxel_data(masks, True, 0, n, 0)
    sm = vp.smooth_voxel_data(created, 3, True)
    res = se.extract_manifold_surface(sm, vp.calculate_slice_depths(float(n)), 1.0, 1.0)
    dt = time.perf_counter() - t0
    assert res is not None
    return {"value": round(n ** 3 / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
            "sample": "%dx%dx%d ellipsoid, whole path (close ends + smooth + field + MC + unique), %.1f s; "
                      "host has %d cores" % (n, n, n, dt, os.cpu_count() or 0),
            "n_vertices": int(len(res[0])), "n_faces": int(len(res[1]))}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    if args.backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)        # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        import torch.distributed as td
        if args.backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group("gloo")
    nz, ny, nx = args.size if args.size else (1024, 1024, 1024)

    timer = FieldTimer()
    timer.install()
    if args.sparse_field:
        pipeline.FIELD_SPARSE = True

    if dist:
        from tomography_3d_reconstructor_amd import slab
        gz = nz * world
        job = slab.SlabJob(gz, ny, nx, slab.TorchDistComm(dev))
        mask = pipeline.ellipsoid_mask(gz, ny, nx, dev, job.z0, job.z1).view(torch.uint8)
        depths = np.full(gz, 1.0)

        def step():
            return job.run(mask, depths, 1.0, 1.0)
        total_voxels = gz * ny * nx
        workload = "%dx%dx%d ellipsoid stack, Z-slabs of %d slices over %d GPUs (halos over RCCL)" % (nx, ny, gz, nz, world)
        parallelism = "zslab%d" % world + ("" if args.backend == "nccl" else " (REHEARSAL over gloo, not a measurement)")
    else:
        mask = pipeline.ellipsoid_mask(nz, ny, nx, dev).view(torch.uint8)
        depths = np.full(nz, 1.0)

        def step():
            return one_pass(mask, depths)
        total_voxels = nz * ny * nx
        workload = "%dx%dx%d ellipsoid stack%s" % (nx, ny, nz, " (BASELINE configs[2])" if (nz, ny, nx) == (1024, 1024, 1024) else "")
        parallelism = "single"

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()

    res = None
    for _ in range(2):          # allocator priming (untimed, like the warm-up): the first passes grow torch's memory pool
        res = step()
    for _ in range(args.warmup):
        res = step()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    rdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the small reductions of this script live
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())

    ms = dt / max(args.steps, 1) * 1e3
    value = total_voxels * args.steps / dt / 1e6
    fms = timer.mean_ms()
    roofline = None
    if fms:
        alg = 5.0 * timer.padded_voxels              # 1 B mask + 4 B field per padded voxel of the launch (rank 0)
        ach = alg / (fms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_field_pmc.json")   # written by tools/profile_round.sh
        if not dist and os.path.exists(pmc) and (nz, ny, nx) == (1024, 1024, 1024):
            traffic = json.load(open(pmc))["hbm_bytes_per_launch"]   # separate rocprofv3 --pmc passes of this command
        roofline = {"bound": "hbm", "kernel": "field_tile_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes": alg, "kernel_ms": round(fms, 4)}
        if args.sparse_field and not dist:
            # the dense figure does not describe this run: most of the field is never written
            roofline = {"bound": "hbm", "kernel": "field_span/comb/worklist/tile kernels (tile-sparse fill)", "achieved": None,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "algorithmic_bytes": alg, "kernel_ms": round(fms, 4),
                        "note": "opt-in sparse field: ~7 % of the tiles are written; the dense 5 B/voxel figure does not apply"}
    nverts = int(res[0].shape[0]) if res else 0
    nfaces = int(res[1].shape[0]) if res else 0
    if dist:
        cnt = torch.tensor([nverts, nfaces], dtype=torch.int64, device=rdev)
        td.all_reduce(cnt)
        nverts, nfaces = [int(x) for x in cnt.cpu()]
    out = {
        "metric": "Mvoxels/s (SDF+MC) on 1024^3 ellipsoid stack; achieved HBM GB/s vs peak",
        "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "parallelism": parallelism, "inputs": "uint8 mask stack resident in HBM",
                   "field": "tile-sparse (opt-in)" if pipeline.FIELD_SPARSE else "dense",
                   "n_vertices": nverts, "n_faces": nfaces},
        "roofline": roofline,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(out), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
