#!/usr/bin/env python3
"""bench.py -- throughput of the hot path (mask stack -> field -> marching cubes -> final mesh).

    python bench.py --gpus N --steps K --warmup W [--workload default|cfg3|cfg4|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts N rank processes itself (a fresh
`python -m torch.distributed.run` child, BEFORE this process touches the GPU; the parent only relays the child's
output and exit code).  Under torch.distributed.run `--gpus` must equal WORLD_SIZE or the run refuses to start.

One "step" = one pass of the whole hot path over one synthetic ellipsoid mask stack that is already resident in HBM
(uint8, 1 B/voxel): pack -> close ends -> opening + 3 closings -> field ("SDF") fill -> Lewiner marching cubes
(classify/scan/list/eval/emit) -> vertex finalisation -> unique/remap.  Outputs stay in HBM.

Workloads (BASELINE.json configs; sizes are width x height x slices there, (nz, ny, nx) here):
  default  N = 1: cfg3 = the 1024^3 ellipsoid.  N > 1: Z-slabs of 1024 slices of 1024^2 per rank (WEAK scaling:
           the N = 1 point is the BENCH line), halos over RCCL (tomography_3d_reconstructor_amd/slab.py) -- followed by a SECOND
           timed block on BASELINE configs[4] (2048x2048x4096, STRONG: the stack north_star quotes "near-linear to 8 GPUs" on),
           reported as `north_star_scaling` with the single-GPU time of the same stack measured on rank 0 in the same run
           (`per_rank_efficiency_vs_n1`); at N = 1 that block is the single-GPU time itself (`cfg5_single_ms_per_step`).
  cfg3     (1024, 1024, 1024) total      cfg4  (2048, 1024, 1024) total      cfg5  (4096, 2048, 2048) total
           -- a FIXED total volume cut into N Z-slabs (STRONG scaling; N = 1 runs the whole stack on one GPU).

Every run checks its own result (`parity_in_run`): SHA-256 of the WHOLE vertex and face arrays -- a Z-slab job forms them rank
after rank, nothing is gathered (SlabJob.mesh_sha256) -- against the committed fixture of the stack (tests/golden/
ellipsoid_hashes.json: the REFERENCE's own output up to 1024^3; ellipsoid_hashes_oracle.json: the pinned oracle's for configs[3]
and [4]); a distributed run on a stack without fixture is compared with a single-GPU pass of the same stack on rank 0.  A
mismatch is printed (`parity_in_run: false`) and the run exits with code 5.

Prints ONE JSON line (rank 0) with metric/value/unit..., plus
  roofline:      the field ("SDF") kernel: `achieved`/`frac` = the bytes the kernel MOVES (rocprofv3 counter bytes of this
                 command where a committed file matches this build of field.hip, else 4 B x padded-pitch voxels written + the bit
                 volume read) / measured kernel time (HIP events on the launch stream, inside the timed region) vs the 8 TB/s
                 HBM3E peak; `frac_survey` = the same with SURVEY 8(d)'s 5 B per padded voxel;
  pass_floor:    the whole pass against its own HBM floor (1 B mask in + 4 B field out per voxel);
  cold_pass_ms, host_to_host_cold_ms / host_to_host_ms: one pass without the size hints of earlier passes; the three
                 drop-in class methods host list in -> host arrays out (PCIe inclusive), FIRST call of the process and warm
                 -- reported beside `value`, never as `value`;
  cpu_baseline:  the CPU oracle (a C/NumPy restatement of the reference) on the SAME 1024^3 stack, one thread, and
                 (`all_cores`) on up to 16 host cores; timed before this process touches the GPU;
  comm (N > 1):  ranks / distinct devices seen by the process group, halo bytes and time per pass.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
WORKLOADS = {"cfg3": (1024, 1024, 1024), "cfg4": (2048, 1024, 1024), "cfg5": (4096, 2048, 2048)}
CONFIG_INDEX = {"cfg3": 2, "cfg4": 3, "cfg5": 4}
# written by tools/profile_round.sh (rocprofv3 --pmc passes of this command); the newest one that exists is used
FIELD_PMCS = [os.path.join("profiles", "r%02d_field_pmc.json" % r) for r in (4, 3, 2, 1)]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="default", choices=["default"] + sorted(WORKLOADS),
                    help="default: 1024^3 per rank (weak); cfg3/cfg4/cfg5: BASELINE configs[2..4], total volume fixed (strong)")
    ap.add_argument("--size", type=int, nargs=3, default=None, metavar=("NZ", "NY", "NX"),
                    help="per-rank volume instead of 1024 1024 1024 (weak), or the TOTAL volume with --strong")
    ap.add_argument("--strong", action="store_true", help="--size is the total volume, cut into --gpus Z-slabs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip cold_pass_ms / host_to_host_ms (N = 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1; gloo (halos staged through the host) only to rehearse the "
                         "multi-rank code path with several ranks on ONE GPU -- never a measurement")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="with --gpus 1: run the N > 1 code path (process group, RCCL communicator, preflight, Z-slab job, comm "
                         "statistics) with ONE rank -- a check of the plumbing on a one-GPU box, never a measurement")
    ap.add_argument("--sparse-field", action="store_true",
                    help="opt-in: do not materialise the parts of the float field that marching cubes cannot read "
                         "(same mesh; NOT the headline configuration -- the roofline entry then only carries a note)")
    ap.add_argument("--one-stream", action="store_true",
                    help="A/B: every kernel of a pass on ONE stream (rounds 1-3); default: the front of a pass (pack + close + smoothing) on "
                         "a second stream, so that it runs under the marching-cubes chain of the pass before")
    ap.add_argument("--equal-slices", action="store_true",
                    help="A/B (N > 1): Z slabs of equal slice counts (rounds 1-3); default: slabs of equal work, from one pass's vertex counts")
    ap.add_argument("--read-every-pass", action="store_true",
                    help="A/B: the host reads the counters of a pass before it enqueues the next one (the GPU idles meanwhile)")
    ap.add_argument("--cpu-sample", type=int, default=1024, help="edge of the cube the CPU oracle is timed on (default: the workload itself)")
    ap.add_argument("--cpu-workers", type=int, default=-1, help="worker processes of the all-core CPU leg (-1: min(16, cores); 0 / 1: skip it)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) time the CPU oracle, print its JSON object, touch no GPU")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run SHA-256 check of the mesh against the golden fixtures")
    ap.add_argument("--no-north-star", action="store_true",
                    help="skip the second block on BASELINE configs[4] (2048x2048x4096, strong scaling; at N = 1 its single-GPU time)")
    ap.add_argument("--north-star-size", type=int, nargs=3, default=None, metavar=("NZ", "NY", "NX"),
                    help="total volume of that block instead of 4096 2048 2048 (rehearsals)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of a self-launched run (0: pick a free one)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launch decision (no GPU, no torch)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_plan(args, env, argv):
    """What this process has to do, decided before anything touches the GPU:
      ("run", world)        -- be a rank (world = 1: the plain single-GPU run)
      ("spawn", command)    -- --gpus N > 1 outside torch.distributed.run: start the N ranks as a child process
      ("refuse", message)   -- --gpus disagrees with the WORLD_SIZE this process was launched with"""
    world = env.get("WORLD_SIZE")
    if world is None:
        if args.gpus <= 1:
            return ("run", 1)
        port = args.master_port or _free_port()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        return ("spawn", cmd)
    world = int(world)
    if args.gpus != world:
        return ("refuse", "bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus "
                          "(or run `python bench.py --gpus N` without torch.distributed.run)" % (args.gpus, world))
    return ("run", world)


def _run_worker(cmd, env):
    """Start the worker as a child that cannot outlive this supervisor: it dies with it (PR_SET_PDEATHSIG) and a SIGTERM /
    SIGINT sent to the supervisor -- torch.distributed.run tearing the job down -- is handed on.  -> the worker's exit code."""
    import ctypes
    import signal

    def die_with_parent():
        try:
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)      # PR_SET_PDEATHSIG
        except Exception:                                                         # noqa: BLE001
            pass
    child = subprocess.Popen(cmd, env=env, preexec_fn=die_with_parent)
    old = {}

    def forward(signum, _frame):
        try:
            child.send_signal(signum)
        except Exception:                                                         # noqa: BLE001
            pass
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sig] = signal.signal(sig, forward)
        except Exception:                                                         # noqa: BLE001
            pass
    try:
        return child.wait()
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)


def retry_port(env, wait_s=30.0):
    """The rendezvous port of the ONE retry, agreed by the rank supervisors of a node without a collective: rank 0 probes for
    a free port from MASTER_PORT + 1 on and publishes it in a file named after the launch (the agent's pid -- every rank's
    supervisor is its child -- and the original port); the others wait for that file.  A rank that does not see it within
    `wait_s` falls back to MASTER_PORT + 1 (what every rank did before, unprobed)."""
    import tempfile
    base = int(env.get("MASTER_PORT", "29500"))
    path = os.path.join(tempfile.gettempdir(), "tomo_bench_retry_%d_%d_%s" % (os.getppid(), base, env.get("TORCHELASTIC_RUN_ID", "none")))
    if env.get("RANK", "0") == "0":
        port = base + 1
        for cand in range(base + 1, base + 65):
            s = socket.socket()
            try:
                s.bind(("127.0.0.1", cand))
                port = cand
                break
            except OSError:
                continue
            finally:
                s.close()
        try:
            with open(path + ".tmp", "w") as f:
                f.write(str(port))
            os.replace(path + ".tmp", path)
        except OSError:
            pass
        return port
    t_end = time.monotonic() + wait_s
    while True:
        try:
            return int(open(path).read().strip())
        except (OSError, ValueError):
            pass
        if time.monotonic() >= t_end:
            return base + 1
        time.sleep(0.05)


def supervise(cmd, env, run=_run_worker, port_wait_s=30.0):
    """A rank launched by torch.distributed.run does not touch the GPU itself: it starts the real worker as a CHILD process
    and relays its exit code.  RCCL's C API (rccl.py, the default transport of the Z-slab job) has no timeout of its own; a
    worker whose watchdog finds no progress for two minutes -- communicator creation, preflight, a pass -- leaves with code 4,
    and every rank's supervisor then starts ONE fresh worker that uses torch.distributed's collectives instead
    (TOMO_RCCL_DIRECT=0) on a fresh rendezvous (a port rank 0's supervisor found free: retry_port; store hosted by rank 0's
    worker).  A hang is collective, so all ranks take the same decision.  `run` is injectable for the CPU tests."""
    env = dict(env)
    env["TOMO_BENCH_WORKER"] = "1"
    rc = run(cmd, env=env)
    if rc == 4 and env.get("TOMO_RCCL_DIRECT", "1") not in ("", "0"):
        print("bench.py: rank %s: the worker gave up (no progress) -- one retry over torch.distributed's collectives"
              % env.get("RANK", "?"), file=sys.stderr, flush=True)
        env["TOMO_RCCL_DIRECT"] = "0"
        env["MASTER_PORT"] = str(retry_port(env, port_wait_s))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        rc = run(cmd, env=env)
    return rc


def workload_shape(args, world):
    """-> (total (gz, ny, nx), scaling, name)."""
    if args.workload != "default":
        gz, ny, nx = WORKLOADS[args.workload]
        return (gz, ny, nx), "strong", args.workload
    if args.size and args.strong:
        gz, ny, nx = args.size
        return (gz, ny, nx), "strong", "custom"
    nz, ny, nx = args.size if args.size else (1024, 1024, 1024)
    return (nz * world, ny, nx), "weak", ("custom" if args.size else ("cfg3" if world == 1 else "cfg3-per-rank"))


# ----------------------------------------------------------------------------- the measured run
class FieldTimer:
    """HIP events around every field-kernel launch (tomo_field_fill*; torch events on the launch stream)."""

    def __init__(self, torch, lib):
        self.pairs = []
        self.padded_voxels = 0
        self.field_bytes = self.bit_bytes = 0
        self.enabled = False
        self.torch = torch
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
                timer.field_bytes = 4 * (nz + 2 * pad) * (ny + 2 * pad) * int(lib.tomo_field_pitch(nx, pad))
                timer.bit_bytes = nz * ny * int(lib.tomo_words_per_row(nx)) * 8
                return rc
            return wrapped

        # ctypes function objects are attributes of the CDLL instance; all three entry points launch the field kernel
        lib.tomo_field_fill = wrap(lib.tomo_field_fill)
        lib.tomo_field_fill_bits = wrap(lib.tomo_field_fill_bits)
        lib.tomo_field_fill_bits_sparse = wrap(lib.tomo_field_fill_bits_sparse)

    def mean_ms(self):
        if not self.pairs:
            return None
        return float(sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs))


def one_pass(mask, depths):
    """One pass of the hot path on a device-resident uint8 mask stack (what a bench step does on one GPU; tools/ use it)."""
    from tomography_3d_reconstructor_amd import pipeline
    vol = pipeline.pack_closed(mask)                   # np.stack + _close_volume_ends, one pass over the mask
    vol = pipeline.smooth(vol, 3, True)
    return pipeline.extract_surface(vol, depths, 1.0, 1.0, True, True)


_FRONT = {}          # device -> the second HIP stream of one_pass_submit


def one_pass_submit(mask, depths, overlap_front=True):
    """one_pass whose counters have not been read yet: -> an object with .result() (pipeline.extract_surface_submit).
    overlap_front: the front of the pass -- pack + close ends + smoothing, which depends on the mask alone and streams at HBM
    rate -- is enqueued on a SECOND HIP stream; the field kernel and the marching-cubes chain wait for it (an event) on the main
    stream.  A caller that submits pass k + 1 before it collects pass k (bench.py's timed loop) thereby lets the front of pass
    k + 1 run under the latency-bound marching-cubes chain of pass k.  Every pass is still complete and checked when its
    .result() returns."""
    import torch
    from tomography_3d_reconstructor_amd import pipeline
    if not overlap_front:
        vol = pipeline.pack_closed(mask)
        vol = pipeline.smooth(vol, 3, True)
        return pipeline.extract_surface_submit(vol, depths, 1.0, 1.0, True, True)
    main = torch.cuda.current_stream(mask.device)
    front = _FRONT.get((str(mask.device), main.cuda_stream))
    if front is None:
        front = _FRONT[(str(mask.device), main.cuda_stream)] = torch.cuda.Stream(device=mask.device)
        front.wait_stream(main)                          # whatever produced the mask is done before the first front runs
    gate = _FRONT.get(("gate", str(mask.device), main.cuda_stream))
    with torch.cuda.stream(front):
        if gate is not None:
            front.wait_event(gate)                        # not next to the field kernel of the pass before (both stream at HBM rate): behind it
        vol = pipeline.smooth(pipeline.pack_closed(mask), 3, True)
        done = torch.cuda.Event()
        done.record(front)
    main.wait_event(done)
    vol.bits.record_stream(main)                         # allocated on the front stream, read by the main one: the allocator must know
    if not (pipeline.MC3 and not pipeline.FIELD_SPARSE):
        return pipeline.extract_surface_submit(vol, depths, 1.0, 1.0, True, True)
    f = pipeline.make_field(vol, True, True)
    gate = torch.cuda.Event()
    gate.record(main)
    _FRONT[("gate", str(mask.device), main.cuda_stream)] = gate
    return pipeline.PendingSurface(surface=pipeline.mc3_vertices(f, depths, 1.0, 1.0, True, defer=True))


def _cpu_chunk(job):
    """One Z-chunk of the stack through the oracle's stage functions, with MARGIN spare slices on either side so that what is
    kept is exactly what one pass over the whole stack gives (the recipe of tests/golden/make_oracle_hashes_slabwise.py, which
    is checked against the reference-derived hashes).  -> (rows kept, triangles)."""
    import numpy as np
    from oracle import oracle as O
    gz, ny, nx, z0, z1 = job
    MARGIN = 13          # 2 (end-slice fill of a chunk) + 8 (smoothing passes) + 2 (Gaussian) + 1 (field slice beyond the cells)
    first, last = z0 == 0, z1 == gz
    a, b = max(0, z0 - MARGIN), min(gz, z1 + MARGIN)
    cx, cy, cz = (nx - 1) / 2.0, (ny - 1) / 2.0, (gz - 1) / 2.0
    ax, ay, az = 0.42 * nx, 0.40 * ny, 0.45 * gz
    exy = ((np.arange(nx, dtype=np.float64)[None, :] - cx) / ax) ** 2 + ((np.arange(ny, dtype=np.float64)[:, None] - cy) / ay) ** 2
    m = np.stack([exy + ((float(z) - cz) / az) ** 2 <= 1.0 for z in range(a, b)])
    sm = O.smooth(O.close_ends(m), 3, True)
    del m
    lo, hi = (0 if first else z0 - 2), (gz if last else z1 + 3)
    f = O.field(sm[lo - a:hi - a], True, True)
    del sm
    Za, Zb = (0 if first else z0 + 1), (gz + 1 if last else z1 + 1)
    try:
        v, fc = O.marching_cubes(f[Za - lo:Zb - lo + 1], 0.5, Za)
    except (ValueError, RuntimeError):
        return 0, 0
    del f
    depths = np.full(gz, 1.0)
    rows = O.finalize_vertices(v, depths, 1.0, 1.0, True, True)
    uniq, inv = np.unique(rows, axis=0, return_inverse=True)
    faces = np.asarray(inv).reshape(-1)[fc]
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    n_top = 0
    if not last:
        ztop = float(O.finalize_vertices(np.array([[Zb, 1.0, 1.0]], np.float32), depths, 1.0, 1.0, True, True)[0, 0])
        n_top = int((uniq[:, 0] == ztop).sum())          # rows on the plane shared with the next chunk belong to that chunk
    return len(uniq) - n_top, int(keep.sum())


def cpu_baseline(n, workers=0):
    """The CPU oracle (C / NumPy restatement of the reference's path) on the n^3 ellipsoid: one thread on the whole stack, and --
    `workers` > 1 -- the same code on `workers` host cores (Z-chunks with spare slices; a process pool).  Runs BEFORE this
    process touches the GPU (bench.main): a process pool is forked, and forking belongs in front of any HIP call."""
    from oracle import oracle as O
    masks = O.ellipsoid_masks(n, n, n)
    vp, se = O.VoxelProcessor(), O.SurfaceExtractor()
    t0 = time.perf_counter()
    created = vp.create_voxel_data(masks, True, 0, n, 0)
    sm = vp.smooth_voxel_data(created, 3, True)
    res = se.extract_manifold_surface(sm, vp.calculate_slice_depths(float(n)), 1.0, 1.0)
    dt = time.perf_counter() - t0
    assert res is not None
    del masks, created, sm
    out = {"value": round(n ** 3 / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
           "sample": "%dx%dx%d ellipsoid, whole path (close ends + smooth + field + MC + unique), %.1f s; "
                     "host has %d cores" % (n, n, n, dt, os.cpu_count() or 0),
           "n_vertices": int(len(res[0])), "n_faces": int(len(res[1]))}
    nv1, nf1 = len(res[0]), len(res[1])
    del res
    if workers > 1:
        import multiprocessing as mp
        thick = max(16, -(-n // (2 * workers)))             # two chunks per worker: the last ones to finish are short
        cuts = list(range(0, n, thick)) + [n]
        if len(cuts) > 2 and cuts[-1] - cuts[-2] < 14:
            cuts.pop(-2)
        jobs = [(n, n, n, cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_cpu_chunk, jobs, chunksize=1)
        dtp = time.perf_counter() - t0
        nvp, nfp = sum(p[0] for p in parts), sum(p[1] for p in parts)
        out["all_cores"] = {"value": round(n ** 3 / dtp / 1e6, 3), "unit": "Mvoxels/s", "cores": workers, "kind": "port",
                            "sample": "the same stack cut into %d Z-chunks of %d slices (+13 spare slices either side), %d worker "
                                      "processes, %.1f s; counts equal to the one-thread run: %s"
                                      % (len(jobs), thick, workers, dtp, (nvp, nfp) == (nv1, nf1)),
                            "n_vertices": nvp, "n_faces": nfp}
    try:
        # the REFERENCE itself (NumPy / SciPy 1.7.1 / scikit-image 0.18.3, one thread), timed once in the build container when
        # the golden hashes were made -- not on this box, quoted for scale: the C port above is ~10x faster than it
        ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ellipsoid_hashes.json")))["1024x1024x1024"]
        out["reference_itself"] = {"value": round(1024 ** 3 / ref["reference_seconds"] / 1e6, 2), "unit": "Mvoxels/s",
                                   "where": "build container (8-core Xeon, 1 thread used), 1024^3, %.0f s; tests/golden/ellipsoid_hashes.json"
                                            % ref["reference_seconds"]}
    except Exception:       # noqa: BLE001
        pass
    return out


def host_to_host(mask_dev, nz, runs=3):
    """SURVEY 8(d)'s metric as written: the three drop-in class methods, host list of 2-D bool masks in -> host
    (vertices, faces) out (upload, the 1 B/voxel volume downloads and the mesh download included)."""
    import contextlib
    import io
    from tomography_3d_reconstructor_amd import SurfaceExtractor, VoxelProcessor
    stack = mask_dev.cpu().numpy().view(bool)
    masks = [stack[i].copy() for i in range(nz)]          # separate pageable arrays, as an image loader produces them
    del stack
    times = []
    shape = None
    for _ in range(runs):                                 # the first run of a process pages its host buffers in
        vp, se = VoxelProcessor(), SurfaceExtractor()
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            vol = vp.create_voxel_data(masks, True, 0, nz, 0)
            depths = vp.calculate_slice_depths(float(nz))
            sm = vp.smooth_voxel_data(vol, iterations=3, create_manifold=True)
            res = se.extract_manifold_surface(sm, depths, 1.0, 1.0, smooth=True, manifold=True, add_padding=True)
            times.append((time.perf_counter() - t0) * 1e3)
        shape = (len(res[0]), len(res[1])) if res is not None else None
        del vol, sm, res
    return times, shape


def two_in_flight(mask, depths, passes):
    """Side measurement (never `value`): two stacks in flight on ONE GPU -- two host threads, each with its own HIP stream,
    each running whole passes on the same resident stack; the latency-bound marching-cubes chain of one overlaps the
    bandwidth-bound stages of the other, and neither waits for the other's host round trip."""
    import threading
    import torch
    dev = mask.device
    bar = threading.Barrier(3)
    errs = []

    def worker():
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                for _ in range(2):
                    one_pass(mask, depths)
                s.synchronize()
                bar.wait()
                for _ in range(passes):
                    one_pass(mask, depths)
                s.synchronize()
            bar.wait()
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            bar.abort()
    th = [threading.Thread(target=worker) for _ in range(2)]
    for t in th:
        t.start()
    try:
        bar.wait()
        t0 = time.perf_counter()
        bar.wait()
        dt = time.perf_counter() - t0
    except threading.BrokenBarrierError:
        dt = None
    for t in th:
        t.join()
    return None if (errs or dt is None) else dt / (2 * passes) * 1e3


GOLDEN_FILES = (("tests/golden/ellipsoid_hashes.json", "reference"), ("tests/golden/ellipsoid_hashes_oracle.json", "oracle-derived"))


def expected_hashes(shape):
    """The committed fixture for the ellipsoid stack of this (gz, ny, nx), if there is one: counts and SHA-256 of the vertex /
    face arrays the REFERENCE returned (<= 1024^3: tests/golden/make_golden.py) or the pinned oracle (BASELINE configs[3], [4])."""
    key = "%dx%dx%d" % tuple(shape)
    files = GOLDEN_FILES
    if os.environ.get("TOMO_BENCH_GOLDEN_OVERRIDE"):          # test hook: a fixture file that replaces the committed ones
        files = ((os.environ["TOMO_BENCH_GOLDEN_OVERRIDE"], "override"),)
    for path, kind in files:
        try:
            ent = json.load(open(os.path.join(ROOT, path))).get(key)
        except Exception:       # noqa: BLE001
            ent = None
        if ent and "vertices_f32_sha256" in ent and "faces_i64_sha256" in ent:
            return {"v": ent["vertices_f32_sha256"], "f": ent["faces_i64_sha256"], "nv": ent["n_vertices"], "nf": ent["n_faces"],
                    "source": "%s (%s)" % (path, kind)}
    return None


def sha_mesh(v, f):
    """(vertices hex, faces hex, n_vertices, n_faces) of a single-GPU mesh (device tensors): hashlib over the host bytes."""
    import hashlib
    import numpy as np
    hv, hf = hashlib.sha256(), hashlib.sha256()
    hv.update(np.ascontiguousarray(v.cpu().numpy(), dtype=np.float32).tobytes())
    hf.update(np.ascontiguousarray(f.cpu().numpy(), dtype=np.int64).tobytes())
    return hv.hexdigest(), hf.hexdigest(), int(v.shape[0]), int(f.shape[0])


def field_pmc_for_this_build():
    """The committed counter bytes of the field kernel (separate rocprofv3 --pmc passes) -- usable only for the kernel they were
    measured on: the file names the SHA-256 of csrc/field.hip it was made with (a kernel change must not leave a
    stale figure behind).  -> (path, bytes per launch) or (None, None)."""
    import hashlib
    try:
        cur = hashlib.sha256(open(os.path.join(ROOT, "tomography_3d_reconstructor_amd", "csrc", "field.hip"), "rb").read()).hexdigest()
    except OSError:
        return None, None
    for p in FIELD_PMCS:
        try:
            d = json.load(open(os.path.join(ROOT, p)))
        except Exception:       # noqa: BLE001
            continue
        if d.get("field_hip_sha256") == cur:
            return p, float(d["hbm_bytes_per_launch"])
    return None, None


def run(args, world, cpu=None):
    import datetime
    import numpy as np
    import torch
    from tomography_3d_reconstructor_amd import _lib, pipeline

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1 or args.rehearse_dist
    if (os.environ.get("TOMO_BENCH_FAKE_HANG") and os.environ.get("TOMO_BENCH_WORKER")
            and os.environ.get("TOMO_RCCL_DIRECT", "1") not in ("", "0")):
        os._exit(4)         # test hook: a worker on the direct transport "hangs" (what the watchdog's exit looks like to the supervisor)
    if args.rehearse_dist and world == 1:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", str(args.master_port or _free_port())), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    json_fd = None
    if dist:
        # RCCL prints a version banner on STDOUT when a communicator is created; the contract is ONE JSON line there.
        # Everything any library writes to fd 1 from here on goes to stderr; the JSON line is written to the saved fd.
        sys.stdout.flush()
        json_fd = os.dup(1)
        os.dup2(2, 1)
    if args.backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)        # rehearsal: ranks may share a GPU
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        print("bench.py: rank %d needs GPU %d, %d visible" % (rank, local, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    td = None
    comm_info = None
    if dist:
        import torch.distributed as td
        tmo = datetime.timedelta(seconds=120)          # a mismatched exchange fails after two minutes instead of hanging
        if args.backend == "nccl":
            td.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            td.init_process_group("gloo", timeout=tmo)
    rdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the small reductions of this script live
    beat = [time.monotonic()]
    if dist:
        # RCCL's C API has no timeout of its own (the process group's watchdog only covers its collectives): a rank that
        # makes no progress for two minutes -- communicator, preflight, a pass -- says so and leaves instead of hanging the node
        def watchdog():
            while True:
                time.sleep(5.0)
                if time.monotonic() - beat[0] > 120.0:
                    print("bench.py: rank %d: no progress for 120 s -- giving up" % rank, file=sys.stderr, flush=True)
                    os._exit(4)
        threading.Thread(target=watchdog, daemon=True).start()

    (gz, ny, nx), scaling, wname = workload_shape(args, world)
    timer = FieldTimer(torch, _lib.lib())
    if args.sparse_field:
        pipeline.FIELD_SPARSE = True

    comm = None
    if dist:
        from tomography_3d_reconstructor_amd import slab
        if args.backend == "nccl" and os.environ.get("TOMO_RCCL_DIRECT", "1") not in ("", "0"):
            # RCCL through its C API on the compute stream (rccl.py): no cross-stream joins, ~0.2 ms less per pass than the
            # process group's calls in the one-GPU rehearsal.  Creating the communicator is collective, so the ranks first
            # agree (all-reduce over the process group) that everybody could load the library.
            from tomography_3d_reconstructor_amd import rccl
            ok = torch.ones(1, dtype=torch.int32, device=dev)
            try:
                rccl.lib()
            except Exception as e:                     # noqa: BLE001
                print("bench.py: rank %d: librccl not usable directly (%r)" % (rank, e), file=sys.stderr, flush=True)
                ok.zero_()
            td.all_reduce(ok, op=td.ReduceOp.MIN)
            if int(ok.item()):
                try:
                    comm = rccl.RcclComm(dev)
                except Exception as e:                 # noqa: BLE001
                    print("bench.py: rank %d: RCCL communicator not created (%r)" % (rank, e), file=sys.stderr, flush=True)
                    ok.zero_()
                td.all_reduce(ok, op=td.ReduceOp.MIN)  # everybody or nobody
                if not int(ok.item()) and comm is not None:
                    comm.close()
                    comm = None
            if comm is None and rank == 0:
                print("bench.py: falling back to torch.distributed collectives", file=sys.stderr, flush=True)
        if comm is None:
            comm = slab.TorchDistComm(dev)
        comm_info = None
        for attempt in range(2):
            # one neighbour exchange + one all-gather, content-checked; who is on which device.  A failure on ANY rank (the
            # verdicts are all-reduced over the process group) sends everybody from the direct communicator to the process
            # group's collectives once; a second failure ends the run with a message instead of a hang.
            good = torch.ones(1, dtype=torch.int32, device=rdev)
            try:
                comm_info = slab.preflight(comm)
            except Exception as e:                     # noqa: BLE001
                print("bench.py: rank %d: preflight over %s failed: %r" % (rank, type(comm).__name__, e), file=sys.stderr, flush=True)
                good.zero_()
            td.all_reduce(good, op=td.ReduceOp.MIN)
            if int(good.item()):
                break
            if attempt == 0 and hasattr(comm, "close"):
                comm.close()
                comm = slab.TorchDistComm(dev)
                continue
            sys.exit(3)
        if args.backend == "nccl" and comm_info["distinct_devices"] != world:
            print("bench.py: %d ranks on %d distinct devices" % (world, comm_info["distinct_devices"]), file=sys.stderr)
            sys.exit(3)
        beat[0] = time.monotonic()

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()

    def free_memory():
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()

    def single_gpu_block(shape, steps, warmup):
        """The whole stack of `shape` on THIS GPU alone: (ms per step, last result) -- the N = 1 point of a strong-scaling
        curve measured in the same run, and the mesh a distributed run must reproduce where no fixture exists."""
        z, y, x = shape
        m = pipeline.ellipsoid_mask(z, y, x, dev).view(torch.uint8)
        d = np.full(z, 1.0)
        r = None
        for _ in range(2):
            r = one_pass_submit(m, d, not args.one_stream).result()
            beat[0] = time.monotonic()
        pend = None
        for _ in range(warmup):
            nxt = one_pass_submit(m, d, not args.one_stream)
            if pend is not None:
                r = pend.result()
            pend = nxt
        if pend is not None:
            r = pend.result()
        best = None
        for _rep in range(2):                               # the better of two timed runs (a denominator must not carry a hiccup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend = None
            for _ in range(steps):
                nxt = one_pass_submit(m, d, not args.one_stream)
                if pend is not None:
                    r = pend.result()
                pend = nxt
                beat[0] = time.monotonic()
            if pend is not None:
                r = pend.result()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t0
            best = dt1 if best is None else min(best, dt1)
        if os.environ.get("TOMO_BENCH_DEBUG"):
            print("single_gpu_block %r: %d steps in %.1f ms; allocator: %s" % (
                shape, steps, best * 1e3, {k: v for k, v in torch.cuda.memory_stats(dev).items()
                                           if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.current")}),
                  file=sys.stderr, flush=True)
        return best / max(steps, 1) * 1e3, r

    def fits_one_gpu(shape):
        free, _total = torch.cuda.mem_get_info(dev)
        return shape[0] * shape[1] * shape[2] * 9.0 < free * 0.9       # mask 1 B + field 4 B + bit volumes, records, mc3 buffers

    def timed_block(shape, steps, warmup, use_timer):
        """Warm-up + EXACTLY `steps` timed passes of the hot path on the ellipsoid stack of `shape` (a Z-slab job when `dist`),
        bracketed by barrier + synchronize, max over ranks.  -> dict(ms, dt, res, job, workload, parallelism, n0)."""
        z, y, x = shape
        job = None
        if dist:
            job = slab.SlabJob(z, y, x, comm)
            mask = pipeline.ellipsoid_mask(z, y, x, dev, job.z0, job.z1).view(torch.uint8)
            depths = np.full(z, 1.0)

            def submit():
                return job.submit(mask, depths, 1.0, 1.0)
            collect = job.result
            parallelism = "zslab%d" % world + ("" if args.backend == "nccl" else " (REHEARSAL over gloo, not a measurement)") + (
                " (REHEARSAL of the multi-rank plumbing with one rank)" if world == 1 else "")
            workload = "%dx%dx%d ellipsoid stack, %d Z-slabs of %d slices (halos over %s)" % (
                x, y, z, world, job.z1 - job.z0, ("RCCL" + (" C API, compute stream" if hasattr(comm, "close") else "")) if args.backend == "nccl" else "gloo")
        else:
            mask = pipeline.ellipsoid_mask(z, y, x, dev).view(torch.uint8)
            depths = np.full(z, 1.0)

            def submit():
                return one_pass_submit(mask, depths, not args.one_stream)

            def collect(p):
                return p.result()
            parallelism = "single"
            workload = "%dx%dx%d ellipsoid stack" % (x, y, z)

        def step():
            out = collect(submit())
            beat[0] = time.monotonic()
            return out

        def steps_in_order(n):
            """n passes, one after the other on one stream.  The host enqueues pass k + 1 BEFORE it reads the counters of pass k
            (one small download per pass): the GPU does not idle while the host reads, checks and slices the outputs of the
            pass before.  Every pass is complete -- results read and checked -- when this returns."""
            out = pend = None
            for _ in range(n):
                nxt = submit()
                if pend is not None:
                    out = collect(pend)
                pend = nxt
                beat[0] = time.monotonic()
            if pend is not None:
                out = collect(pend)
            return out
        res = None
        for _ in range(2):          # allocator priming (untimed, like the warm-up): the first passes grow torch's memory pool
            res = step()
        cuts_note = None
        if job is not None and world > 1 and not args.equal_slices:
            # Z slabs of equal WORK instead of equal slice counts: the per-slice vertex counts of the pass just made (all-gathered)
            # weigh the marching-cubes chain's share of a pass against the volume-sized kernels'; every rank computes the same cuts
            share = min(0.5, 0.29 * 1024.0 / float(y * x) ** 0.5)          # chain / pass on one GPU: 0.29 at 1024^2 slices, 0.15 at 2048^2
            cuts = job.work_balanced_cuts(depths, share, align=4)
            cuts_note = {"z_cuts": cuts, "equal_slices_would_be": list(job.cuts), "surface_share_assumed": round(share, 3)}
            if cuts != job.cuts:
                res = None
                job = slab.SlabJob(z, y, x, comm, z_cuts=cuts)
                mask = None
                free_memory()
                mask = pipeline.ellipsoid_mask(z, y, x, dev, job.z0, job.z1).view(torch.uint8)
                collect = job.result
                workload = "%dx%dx%d ellipsoid stack, %d Z-slabs of %d..%d slices (equal work; halos over %s)" % (
                    x, y, z, world, job.thinnest, max(b - a for a, b in zip(cuts, cuts[1:])),
                    ("RCCL" + (" C API, compute stream" if hasattr(comm, "close") else "")) if args.backend == "nccl" else "gloo")
                for _ in range(2):
                    res = step()
        if warmup:
            res = steps_in_order(warmup)
        if comm is not None:
            comm.reset_stats()
        n0 = (job.deferred_passes, job.deferred_redone) if job is not None else (0, 0)
        barrier()
        timer.enabled = use_timer
        t0 = time.perf_counter()
        res = steps_in_order(steps) if not args.read_every_pass else [step() for _ in range(steps)][-1]
        barrier()
        dt = time.perf_counter() - t0
        timer.enabled = False
        if dist:
            t = torch.tensor([dt], dtype=torch.float64, device=rdev)
            td.all_reduce(t, op=td.ReduceOp.MAX)
            dt = float(t.item())
        return {"ms": dt / max(steps, 1) * 1e3, "dt": dt, "res": res, "job": job, "mask": mask, "depths": depths, "step": step,
                "workload": workload, "parallelism": parallelism, "n0": n0, "cuts": cuts_note}

    def verify(shape, blk, keep_single=False):
        """Is the mesh of this run the reference's?  SHA-256 of the WHOLE vertex / face arrays -- a Z-slab job forms them rank
        after rank (SlabJob.mesh_sha256: nothing is gathered) -- against the committed fixture of this stack
        (reference-derived up to 1024^3, oracle-derived for BASELINE configs[3] / [4]); where no fixture exists a distributed run
        is compared with a single-GPU pass of the same stack on rank 0 (when it fits).  The reference's semantics: ONE globally
        np.unique-numbered vertex list, faces in cell order (surface_extractor.py:115-126).
        -> dict(parity_in_run: True | False | "no fixture", ...); also returns rank 0's single-GPU time when it was measured."""
        res, job = blk["res"], blk["job"]
        if res is None:
            got = ("", "", 0, 0)
        elif job is not None:
            got = job.mesh_sha256((res[0], res[1]))
        else:
            got = sha_mesh(res[0], res[1])
        beat[0] = time.monotonic()
        out = {"vertices_f32_sha256": got[0], "faces_i64_sha256": got[1], "n_vertices": got[2], "n_faces": got[3]}
        exp = expected_hashes(shape)
        if exp is not None:
            out["parity_in_run"] = bool((got[0], got[1], got[2], got[3]) == (exp["v"], exp["f"], exp["nv"], exp["nf"]))
            out["parity_against"] = exp["source"]
        single_ms = None
        need_single = (exp is None or keep_single) and dist and world > 1
        if need_single:
            # free this rank's share of the job first: rank 0 needs the room (and in a gloo rehearsal the ranks share a GPU)
            blk["res"] = blk["job"] = blk["mask"] = blk["step"] = None
            res = job = None
            free_memory()
            barrier()
            verdict = torch.zeros(3, dtype=torch.float64, device=rdev)       # [measured?, equal?, ms]
            if rank == 0 and fits_one_gpu(shape):
                ms1, r1 = single_gpu_block(shape, max(3, min(args.steps, 10)), 1)
                ref = sha_mesh(r1[0], r1[1]) if r1 is not None else ("", "", 0, 0)
                verdict = torch.tensor([1.0, 1.0 if tuple(ref) == tuple(got) else 0.0, ms1], dtype=torch.float64, device=rdev)
                del r1
                free_memory()
            td.all_reduce(verdict, op=td.ReduceOp.MAX)
            beat[0] = time.monotonic()
            if verdict[0].item():
                single_ms = float(verdict[2].item())
                if exp is None:
                    out["parity_in_run"] = bool(verdict[1].item())
                    out["parity_against"] = "a single-GPU pass of the same stack on rank 0, in this run"
                else:
                    out["equal_to_single_gpu_pass_in_this_run"] = bool(verdict[1].item())
        if "parity_in_run" not in out:
            out["parity_in_run"] = "no fixture"
            out["parity_against"] = None
        return out, single_ms

    # ------------------------------------------------------------------ the headline block
    blk = timed_block((gz, ny, nx), args.steps, args.warmup, True)
    workload, parallelism, job, res = blk["workload"], blk["parallelism"], blk["job"], blk["res"]
    if wname in CONFIG_INDEX and (gz, ny, nx) == WORKLOADS[wname]:
        workload += " (BASELINE configs[%d])" % CONFIG_INDEX[wname]
    total_voxels = gz * ny * nx
    steps = max(args.steps, 1)
    ms, dt = blk["ms"], blk["dt"]
    value = total_voxels * args.steps / dt / 1e6
    fms = timer.mean_ms()
    roofline = None
    if fms:
        # What the kernel MOVES: it writes the float32 field in whole 128-byte lines (4 B x Nz x Ny x pitch) and reads the
        # bit-packed volume (1/8 B per voxel; the 1 B/voxel mask is read by pack_close_kernel, not here) -- `touched`, from the
        # geometry of the launch; for the 1024^3 workload the rocprofv3 counter bytes of the same command are used instead
        # (`traffic`, separate --pmc passes of THIS build of the kernel).  SURVEY 8(d) prices the kernel at 5 B per padded voxel
        # (1 B mask + 4 B f32): kept as achieved_survey / frac_survey.
        alg = 5.0 * timer.padded_voxels
        touched = float(timer.field_bytes + timer.bit_bytes)
        pmc, traffic = (None, None)
        if not dist and (gz, ny, nx) == (1024, 1024, 1024) and not args.sparse_field:
            pmc, traffic = field_pmc_for_this_build()
        moved = traffic if traffic is not None else touched
        ach = moved / (fms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "field_tile_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": moved, "kernel_ms": round(fms, 4),
                    "bytes_source": ("%s (HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on "
                                     "this build of field.hip, gfx950 corrections applied; not measured in this run)" % pmc) if traffic is not None else
                                    "geometry of the launch: 4 B x Nz x Ny x pitch written + the bit-packed volume read once "
                                    "(no committed counter file matches this build of field.hip)",
                    "touched_bytes": touched,
                    "algorithmic_bytes_survey": alg, "achieved_survey": round(alg / (fms * 1e-3) / 1e9, 1),
                    "frac_survey": round(alg / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "note": "achieved / frac: the bytes the kernel moves / kernel time (HIP events on the launch stream, inside the timed "
                            "region) vs the 8 TB/s HBM3E peak.  *_survey: SURVEY 8(d)'s 5 B per padded voxel (1 B mask + 4 B f32), which "
                            "credits the kernel with the mask bytes that pack_close_kernel reads."}
        if args.sparse_field and not dist:
            # the dense figure does not describe this run: most of the field is never written
            roofline = {"bound": "hbm", "kernel": "field_span/comb/worklist/tile kernels (tile-sparse fill)", "achieved": None,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                        "algorithmic_bytes_survey": alg, "kernel_ms": round(fms, 4),
                        "note": "opt-in sparse field: ~7 % of the tiles are written; the dense 5 B/voxel figure does not apply"}
    comm_out = None
    if dist:
        st = comm.stats
        agg = torch.tensor([st["bytes_sent"], st["calls"]], dtype=torch.float64, device=rdev)
        td.all_reduce(agg)
        mx = torch.tensor([st["seconds"], st["bytes_sent"]], dtype=torch.float64, device=rdev)
        td.all_reduce(mx, op=td.ReduceOp.MAX)
        comm_out = dict(comm_info)
        if blk.get("cuts"):
            comm_out["slabs"] = blk["cuts"]
        comm_out.update({
            "halo_bytes_per_pass_all_ranks": int(agg[0].item() / steps), "halo_bytes_per_pass_max_rank": int(mx[1].item() / steps),
            "comm_calls_per_pass_per_rank": round(agg[1].item() / steps / world, 1),
            "comm_ms_per_pass_max_rank": round(mx[0].item() / steps * 1e3, 3),
            "comm_ms_note": "host wall time inside exchange/all_gather calls, includes waiting for the kernels that produce the halos",
            "numbering": {"deferred_passes": job.deferred_passes, "redone": job.deferred_redone,
                          "note": "rank 0, warm-up included: passes that ran with ONE download at the end (chain from size hints, "
                                  "shared-plane rows in fixed-capacity messages, global indices from device-side counts); the "
                                  "first pass of a job is always the exact one (four host round trips)"}})
    # ------------------------------------------------------------------ is it the reference's mesh?  (every run says so itself)
    job = None                                        # (verify may have to free this rank's share of the job: no second reference to it)
    if args.no_parity:
        parity = {"parity_in_run": "not checked (--no-parity)"}
        nverts = nfaces = 0
        if res is not None:
            cnt = torch.tensor([int(res[0].shape[0]), int(res[1].shape[0])], dtype=torch.int64, device=rdev)
            if dist:
                td.all_reduce(cnt)
            nverts, nfaces = [int(c) for c in cnt.cpu()]
        res = None
    else:
        res = None
        parity, _ = verify((gz, ny, nx), blk)
        nverts, nfaces = parity.get("n_vertices", 0), parity.get("n_faces", 0)
    failed = parity["parity_in_run"] is False
    pass_floor_bytes = 5.0 * total_voxels            # 1 B mask in + 4 B f32 field out per voxel: what a pass cannot avoid moving
    out = {
        "metric": "Mvoxels/s (SDF+MC) on 1024^3 ellipsoid stack; achieved HBM GB/s vs peak",
        "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "value_note": ("the host reads every pass before it enqueues the next" if args.read_every_pass else
                       "PIPELINED submission: the host enqueues pass k+1 before it reads the counters of pass k (rounds 1-2 read every "
                       "pass: `read_every_pass_ms_per_step`)" + ("" if (args.one_stream or dist) else
                       "; and the FRONT of pass k+1 (pack + close ends + smoothing: needs the mask alone) is enqueued on a second HIP "
                       "stream, so it runs under the marching-cubes chain of pass k -- field kernel and chain of consecutive passes "
                       "still run strictly in order on the main stream (`one_stream_ms_per_step`: everything on one stream, round 3's "
                       "`ms_per_step`)")),
        "config": {"workload": workload, "workload_id": wname, "parallelism": parallelism,
                   "inputs": "uint8 mask stack resident in HBM", "outputs": "final (vertices, faces) resident in HBM",
                   "field": "tile-sparse (opt-in)" if pipeline.FIELD_SPARSE else "dense",
                   "submission": ("host reads every pass before enqueueing the next" if args.read_every_pass else
                                  ("passes strictly one after the other on one stream; the host enqueues pass k+1 before it reads the counters of pass k"
                                   if (args.one_stream or dist) else
                                   "the host enqueues pass k+1 before it reads the counters of pass k; pack + close + smoothing of a pass on a second "
                                   "stream (may run under the marching-cubes chain of the pass before), field + chain in order on the main stream")),
                   "n_vertices": nverts, "n_faces": nfaces},
        "parity_in_run": parity["parity_in_run"], "parity": parity,
        "roofline": roofline,
        "pass_floor": {"bytes": pass_floor_bytes, "frac": round(pass_floor_bytes / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "whole pass vs its own HBM floor (1 B mask in + 4 B field out per voxel) per GPU at 8 TB/s"},
    }
    if comm_out is not None:
        out["comm"] = comm_out
    mask, depths, step = blk["mask"], blk["depths"], blk["step"]
    if not dist and not args.no_extras:
        res = blk["res"] = None
        pipeline._NA_HINT.clear()
        pipeline._MC3_HINT.clear()
        for k in pipeline.COUNTERS:
            pipeline.COUNTERS[k] = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = step()
        torch.cuda.synchronize()
        out["cold_pass_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        out["cold_pass_note"] = "one pass with the marching-cubes size hints and path counters cleared (allocator warm)"
        del r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(4, min(args.steps, 10))):
            r = step()
        torch.cuda.synchronize()
        out["read_every_pass_ms_per_step"] = round((time.perf_counter() - t0) / max(4, min(args.steps, 10)) * 1e3, 3)
        out["read_every_pass_note"] = "side measurement: the host reads a pass's counters before it enqueues the next pass (round 1 / 2's `ms_per_step`)"
        del r
        if not args.one_stream:
            kk = max(4, min(args.steps, 10))
            pend = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(kk):
                nxt = one_pass_submit(mask, depths, False)
                if pend is not None:
                    pend.result()
                pend = nxt
            pend.result()
            torch.cuda.synchronize()
            out["one_stream_ms_per_step"] = round((time.perf_counter() - t0) / kk * 1e3, 3)
            out["one_stream_note"] = "side measurement: every kernel of a pass on ONE stream, pass after pass (round 3's `ms_per_step`)"
        tif = two_in_flight(mask, depths, max(4, min(args.steps, 10)))
        if tif is not None:
            out["two_in_flight_ms_per_pass"] = round(tif, 3)
            out["two_in_flight_note"] = ("side measurement, never `value`: two host threads / HIP streams, whole passes on the same "
                                         "resident stack (`value` runs the passes one after the other on one stream)")
        if total_voxels <= 2 ** 31:
            from tomography_3d_reconstructor_amd import _devcache
            times, shape = host_to_host(mask, gz)
            out["host_to_host_cold_ms"] = round(times[0], 2)
            out["host_to_host_ms"] = round(min(times[1:]), 2)
            out["host_to_host_runs_ms"] = [round(x, 2) for x in times]
            out["host_to_host_mvoxels_s"] = round(total_voxels / (min(times[1:]) * 1e-3) / 1e6, 1)
            out["host_to_host_cold_mvoxels_s"] = round(total_voxels / (times[0] * 1e-3) / 1e6, 1)
            out["host_to_host_results"] = "writeable arrays, verified by checksum (the reference's semantics; default)" if _devcache.WRITEABLE_RESULTS \
                else "write-protected arrays (TOMO_READONLY_RESULTS=1)"
            # the other mode of the result cache, same process (its buffers are warm by now): what the choice of default costs
            was = _devcache.WRITEABLE_RESULTS
            _devcache.WRITEABLE_RESULTS = not was
            try:
                t2, _ = host_to_host(mask, gz, runs=2)
                out["host_to_host_ms_readonly_results" if was else "host_to_host_ms_writeable_results"] = round(min(t2), 2)
            finally:
                _devcache.WRITEABLE_RESULTS = was
            out["host_to_host_note"] = ("create_voxel_data + smooth_voxel_data + extract_manifold_surface of the drop-in classes, "
                                        "host list of masks in -> host arrays out (PCIe inclusive), same volume; SURVEY 8(d)'s "
                                        "end-to-end metric; never `value`; *_cold_*: the FIRST such call of the process "
                                        "(what one run of the reference's main() sees), host_to_host_ms: the best later one; "
                                        "*_readonly_results: the opt-in mode that hands out write-protected volumes and skips the "
                                        "checksums; mesh %s" % (shape,))
    # ------------------------------------------------------------------ the north-star scaling workload (BASELINE configs[4], STRONG)
    ns_shape = tuple(args.north_star_size) if args.north_star_size else WORKLOADS["cfg5"]
    want_ns = args.workload == "default" and not args.size and not args.no_north_star and not args.sparse_field
    if args.north_star_size and not args.no_north_star:
        want_ns = True
    if want_ns and not dist and not args.no_extras:
        # N = 1: the denominator of the strong-scaling curve, so that it is a driver record whatever N the driver can reach
        blk.clear()
        mask = step = None
        free_memory()
        if fits_one_gpu(ns_shape):
            ms1, r1 = single_gpu_block(ns_shape, max(3, min(args.steps, 10)), 1)
            ent = {"workload": "%dx%dx%d ellipsoid stack%s" % (ns_shape[2], ns_shape[1], ns_shape[0],
                                                                " (BASELINE configs[4])" if ns_shape == WORKLOADS["cfg5"] else ""),
                   "scaling": "strong", "n_gpus": 1, "ms_per_step": round(ms1, 3),
                   "value": round(ns_shape[0] * ns_shape[1] * ns_shape[2] / (ms1 * 1e-3) / 1e6, 1), "unit": "Mvoxels/s"}
            if not args.no_parity and r1 is not None:
                got = sha_mesh(r1[0], r1[1])
                exp = expected_hashes(ns_shape)
                ent["n_vertices"], ent["n_faces"] = got[2], got[3]
                ent["parity_in_run"] = "no fixture" if exp is None else bool(tuple(got) == (exp["v"], exp["f"], exp["nv"], exp["nf"]))
                ent["parity_against"] = None if exp is None else exp["source"]
                failed = failed or ent["parity_in_run"] is False
            del r1
            out["north_star_scaling"] = ent
            out["cfg5_single_ms_per_step"] = ent["ms_per_step"] if ns_shape == WORKLOADS["cfg5"] else None
    elif want_ns and dist and world > 1:
        blk.clear()
        mask = step = job = res = None
        free_memory()
        barrier()
        k_ns = max(3, min(args.steps, 10))
        nb = timed_block(ns_shape, k_ns, max(1, min(args.warmup, 3)), False)
        ent = {"workload": "%s%s" % (nb["workload"], " (BASELINE configs[4])" if ns_shape == WORKLOADS["cfg5"] else ""),
               "scaling": "strong", "n_gpus": world, "steps": k_ns, "ms_per_step": round(nb["ms"], 3),
               "value": round(ns_shape[0] * ns_shape[1] * ns_shape[2] / (nb["ms"] * 1e-3) / 1e6, 1), "unit": "Mvoxels/s",
               "numbering": {"deferred_passes": nb["job"].deferred_passes, "redone": nb["job"].deferred_redone}, "slabs": nb.get("cuts")}
        if not args.no_parity:
            par, single_ms = verify(ns_shape, nb, keep_single=True)
            ent.update(par)
            failed = failed or par["parity_in_run"] is False or par.get("equal_to_single_gpu_pass_in_this_run") is False
            if single_ms is not None:
                ent["n1_ms_per_step_same_run"] = round(single_ms, 3)
                ent["per_rank_efficiency_vs_n1"] = round(single_ms / (world * nb["ms"]), 4)
                ent["efficiency_note"] = ("(time of the whole stack on rank 0's GPU alone, measured in this run) / (N x time of the "
                                          "Z-slab job): 1.0 = linear")
        nb.clear()
        out["north_star_scaling"] = ent
    if rank == 0:
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)
        if failed:
            print("bench.py: PARITY FAILED: the mesh of this run is not the reference's (see `parity` / `north_star_scaling`)",
                  file=sys.stderr, flush=True)
    if dist:
        if hasattr(comm, "close"):
            comm.close()
        td.destroy_process_group()
    return 5 if failed else 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    plan = launch_plan(args, os.environ, argv)
    if plan[0] == "refuse":
        print(plan[1], file=sys.stderr)
        return 2
    if plan[0] == "spawn":
        # this process has not touched the GPU (no torch import yet): start the ranks as a child and hand its code on
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "4")
        return subprocess.call(plan[1], env=env)
    supervised = (plan[1] > 1 and args.backend == "nccl") or os.environ.get("TOMO_BENCH_SUPERVISE") == "1"    # (the latter: test hook)
    if supervised and not os.environ.get("TOMO_BENCH_WORKER"):
        # a rank of a multi-GPU run: supervise a worker child (this process never initialises the GPU, so this is a child
        # process, not an exec of a GPU process)
        return supervise([sys.executable, os.path.abspath(__file__)] + list(argv), os.environ)
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.cpu_sample, args.cpu_workers if args.cpu_workers >= 0 else min(16, os.cpu_count() or 1))), flush=True)
        return 0
    cpu = None
    if plan[1] == 1 and not args.rehearse_dist and not args.no_cpu_baseline:
        # In a CHILD process, started before this one touches the GPU: the oracle's 1024^3 run holds ~25 GB for a while and its
        # all-core leg forks a pool -- neither belongs in the address space the host-to-host measurements below page their arrays
        # into (in-process, the first create -> smooth -> extract of the classes took 170 ms instead of 137)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-sample", str(args.cpu_sample),
                            "--cpu-workers", str(args.cpu_workers)], stdout=subprocess.PIPE, text=True)
        try:
            cpu = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
        except Exception:       # noqa: BLE001
            cpu = {"error": "the CPU baseline child left with code %d" % r.returncode}
    return run(args, plan[1], cpu)


if __name__ == "__main__":
    sys.exit(main())
