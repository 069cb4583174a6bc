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
           the N = 1 point is the BENCH line), halos over RCCL (tomography_3d_reconstructor_amd/slab.py).
  cfg3     (1024, 1024, 1024) total      cfg4  (2048, 1024, 1024) total      cfg5  (4096, 2048, 2048) total
           -- a FIXED total volume cut into N Z-slabs (STRONG scaling; N = 1 runs the whole stack on one GPU).

Prints ONE JSON line (rank 0) with metric/value/unit..., plus
  roofline:      the field ("SDF") kernel: `achieved`/`frac` = the bytes the kernel MOVES (rocprofv3 counter bytes of this
                 command where committed, else 4 B x padded-pitch voxels written + the bit volume read) / measured kernel
                 time (HIP events on the launch stream, inside the timed region) vs the 8 TB/s HBM3E peak; `frac_survey` =
                 the same with SURVEY 8(d)'s 5 B per padded voxel (which credits it with the mask bytes pack_close_kernel reads);
  pass_floor:    the whole pass against its own HBM floor (1 B mask in + 4 B field out per voxel);
  cold_pass_ms, host_to_host_cold_ms / host_to_host_ms: one pass without the size hints of earlier passes; the three
                 drop-in class methods host list in -> host arrays out (PCIe inclusive), FIRST call of the process and warm
                 -- reported beside `value`, never as `value`;
  cpu_baseline:  the CPU oracle (a C/NumPy restatement of the reference, single thread) on a bounded sample;
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
FIELD_PMCS = [os.path.join("profiles", "r%02d_field_pmc.json" % r) for r in (3, 2, 1)]


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
    ap.add_argument("--read-every-pass", action="store_true",
                    help="A/B: the host reads the counters of a pass before it enqueues the next one (the GPU idles meanwhile)")
    ap.add_argument("--cpu-sample", type=int, default=768, help="edge of the cube the CPU oracle is timed on")
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


def supervise(cmd, env, run=_run_worker):
    """A rank launched by torch.distributed.run does not touch the GPU itself: it starts the real worker as a CHILD process
    and relays its exit code.  RCCL's C API (rccl.py, the default transport of the Z-slab job) has no timeout of its own; a
    worker whose watchdog finds no progress for two minutes -- communicator creation, preflight, a pass -- leaves with code 4,
    and every rank's supervisor then starts ONE fresh worker that uses torch.distributed's collectives instead
    (TOMO_RCCL_DIRECT=0) on a fresh rendezvous (next port, store hosted by rank 0's worker).  A hang is collective, so all
    ranks take the same decision.  `run` is injectable for the CPU tests."""
    env = dict(env)
    env["TOMO_BENCH_WORKER"] = "1"
    rc = run(cmd, env=env)
    if rc == 4 and env.get("TOMO_RCCL_DIRECT", "1") not in ("", "0"):
        print("bench.py: rank %s: the worker gave up (no progress) -- one retry over torch.distributed's collectives"
              % env.get("RANK", "?"), file=sys.stderr, flush=True)
        env["TOMO_RCCL_DIRECT"] = "0"
        env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + 1)
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


def one_pass_submit(mask, depths):
    """one_pass whose counters have not been read yet: -> an object with .result() (pipeline.extract_surface_submit)."""
    from tomography_3d_reconstructor_amd import pipeline
    vol = pipeline.pack_closed(mask)
    vol = pipeline.smooth(vol, 3, True)
    return pipeline.extract_surface_submit(vol, depths, 1.0, 1.0, True, True)


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
    out = {"value": round(n ** 3 / dt / 1e6, 3), "unit": "Mvoxels/s", "cores": 1, "kind": "port",
           "sample": "%dx%dx%d ellipsoid, whole path (close ends + smooth + field + MC + unique), %.1f s; "
                     "host has %d cores" % (n, n, n, dt, os.cpu_count() or 0),
           "n_vertices": int(len(res[0])), "n_faces": int(len(res[1]))}
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


def host_to_host(mask_dev, nz):
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
    for _ in range(3):                                    # the first run page-locks its host buffers (once per process)
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


def run(args, world):
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

    job = comm = None
    if dist:
        from tomography_3d_reconstructor_amd import slab
        comm = None
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
        job = slab.SlabJob(gz, ny, nx, comm)
        mask = pipeline.ellipsoid_mask(gz, ny, nx, dev, job.z0, job.z1).view(torch.uint8)
        depths = np.full(gz, 1.0)

        def submit():
            return job.submit(mask, depths, 1.0, 1.0)
        collect = job.result
        parallelism = "zslab%d" % world + ("" if args.backend == "nccl" else " (REHEARSAL over gloo, not a measurement)") + (
            " (REHEARSAL of the multi-rank plumbing with one rank)" if world == 1 else "")
        workload = "%dx%dx%d ellipsoid stack, %d Z-slabs of %d slices (halos over %s)" % (
            nx, ny, gz, world, job.z1 - job.z0, ("RCCL" + (" C API, compute stream" if hasattr(comm, "close") else "")) if args.backend == "nccl" else "gloo")
    else:
        mask = pipeline.ellipsoid_mask(gz, ny, nx, dev).view(torch.uint8)
        depths = np.full(gz, 1.0)

        def submit():
            return one_pass_submit(mask, depths)

        def collect(p):
            return p.result()
        parallelism = "single"
        workload = "%dx%dx%d ellipsoid stack" % (nx, ny, gz)
    if wname in CONFIG_INDEX and (gz, ny, nx) == WORKLOADS[wname]:
        workload += " (BASELINE configs[%d])" % CONFIG_INDEX[wname]
    total_voxels = gz * ny * nx

    def barrier():
        torch.cuda.synchronize()
        if dist:
            td.barrier()
            torch.cuda.synchronize()

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
    if args.warmup:
        res = steps_in_order(args.warmup)
    if comm is not None:
        comm.reset_stats()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    res = steps_in_order(args.steps) if not args.read_every_pass else [step() for _ in range(args.steps)][-1]
    barrier()
    dt = time.perf_counter() - t0
    timer.enabled = False
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=rdev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())

    steps = max(args.steps, 1)
    ms = dt / steps * 1e3
    value = total_voxels * args.steps / dt / 1e6
    fms = timer.mean_ms()
    roofline = None
    if fms:
        # What the kernel MOVES: it writes the float32 field in whole 128-byte lines (4 B x Nz x Ny x pitch) and reads the
        # bit-packed volume (1/8 B per voxel; the 1 B/voxel mask is read by pack_close_kernel, not here) -- `touched`, from the
        # geometry of the launch; for the 1024^3 workload the rocprofv3 counter bytes of the same command are used instead
        # (`traffic`, separate --pmc passes).  SURVEY 8(d) prices the kernel at 5 B per padded voxel (1 B mask + 4 B f32): kept
        # as achieved_survey / frac_survey.
        alg = 5.0 * timer.padded_voxels
        touched = float(timer.field_bytes + timer.bit_bytes)
        pmc = next((p for p in FIELD_PMCS if os.path.exists(os.path.join(ROOT, p))), None)
        traffic = None
        if not dist and pmc and (gz, ny, nx) == (1024, 1024, 1024) and not args.sparse_field:
            traffic = float(json.load(open(os.path.join(ROOT, pmc)))["hbm_bytes_per_launch"])
        moved = traffic if traffic is not None else touched
        ach = moved / (fms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "field_tile_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "bytes_per_launch": moved, "kernel_ms": round(fms, 4),
                    "bytes_source": ("%s (HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                     "gfx950 corrections applied; not measured in this run)" % pmc) if traffic is not None else
                                    "geometry of the launch: 4 B x Nz x Ny x pitch written + the bit-packed volume read once",
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
    nverts = int(res[0].shape[0]) if res else 0
    nfaces = int(res[1].shape[0]) if res else 0
    comm_out = None
    if dist:
        cnt = torch.tensor([nverts, nfaces], dtype=torch.int64, device=rdev)
        td.all_reduce(cnt)
        nverts, nfaces = [int(x) for x in cnt.cpu()]
        st = comm.stats
        agg = torch.tensor([st["bytes_sent"], st["calls"]], dtype=torch.float64, device=rdev)
        td.all_reduce(agg)
        mx = torch.tensor([st["seconds"], st["bytes_sent"]], dtype=torch.float64, device=rdev)
        td.all_reduce(mx, op=td.ReduceOp.MAX)
        comm_out = dict(comm_info)
        comm_out.update({
            "halo_bytes_per_pass_all_ranks": int(agg[0].item() / steps), "halo_bytes_per_pass_max_rank": int(mx[1].item() / steps),
            "comm_calls_per_pass_per_rank": round(agg[1].item() / steps / world, 1),
            "comm_ms_per_pass_max_rank": round(mx[0].item() / steps * 1e3, 3),
            "comm_ms_note": "host wall time inside exchange/all_gather calls, includes waiting for the kernels that produce the halos",
            "numbering": {"deferred_passes": job.deferred_passes, "redone": job.deferred_redone,
                          "note": "rank 0, warm-up included: passes that ran with ONE download at the end (chain from size hints, "
                                  "shared-plane rows in fixed-capacity messages, global indices from device-side counts); the "
                                  "first pass of a job is always the exact one (four host round trips)"}})
    pass_floor_bytes = 5.0 * total_voxels            # 1 B mask in + 4 B f32 field out per voxel: what a pass cannot avoid moving
    out = {
        "metric": "Mvoxels/s (SDF+MC) on 1024^3 ellipsoid stack; achieved HBM GB/s vs peak",
        "value": round(value, 1), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "workload_id": wname, "parallelism": parallelism,
                   "inputs": "uint8 mask stack resident in HBM", "outputs": "final (vertices, faces) resident in HBM",
                   "field": "tile-sparse (opt-in)" if pipeline.FIELD_SPARSE else "dense",
                   "submission": ("host reads every pass before enqueueing the next" if args.read_every_pass else
                                  "passes strictly one after the other on one stream; the host enqueues pass k+1 before it reads the counters of pass k"),
                   "n_vertices": nverts, "n_faces": nfaces},
        "roofline": roofline,
        "pass_floor": {"bytes": pass_floor_bytes, "frac": round(pass_floor_bytes / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "whole pass vs its own HBM floor (1 B mask in + 4 B field out per voxel) per GPU at 8 TB/s"},
    }
    if comm_out is not None:
        out["comm"] = comm_out
    if not dist and not args.no_extras:
        del res
        pipeline._NA_HINT.clear()
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
        tif = two_in_flight(mask, depths, max(4, min(args.steps, 10)))
        if tif is not None:
            out["two_in_flight_ms_per_pass"] = round(tif, 3)
            out["two_in_flight_note"] = ("side measurement, never `value`: two host threads / HIP streams, whole passes on the same "
                                         "resident stack (`value` runs the passes one after the other on one stream)")
        if total_voxels <= 2 ** 31:
            times, shape = host_to_host(mask, gz)
            out["host_to_host_cold_ms"] = round(times[0], 2)
            out["host_to_host_ms"] = round(min(times[1:]), 2)
            out["host_to_host_runs_ms"] = [round(x, 2) for x in times]
            out["host_to_host_mvoxels_s"] = round(total_voxels / (min(times[1:]) * 1e-3) / 1e6, 1)
            out["host_to_host_cold_mvoxels_s"] = round(total_voxels / (times[0] * 1e-3) / 1e6, 1)
            out["host_to_host_note"] = ("create_voxel_data + smooth_voxel_data + extract_manifold_surface of the drop-in classes, "
                                        "host list of masks in -> host arrays out (PCIe inclusive), same volume; SURVEY 8(d)'s "
                                        "end-to-end metric; never `value`; *_cold_*: the FIRST such call of the process (page-locks its "
                                        "host buffers; what one run of the reference's main() sees), host_to_host_ms: the best later one; "
                                        "mesh %s" % (shape,))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        if json_fd is not None:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        else:
            print(json.dumps(out), flush=True)
    if dist:
        if hasattr(comm, "close"):
            comm.close()
        td.destroy_process_group()


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
    run(args, plan[1])
    return 0


if __name__ == "__main__":
    sys.exit(main())
