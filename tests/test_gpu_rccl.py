"""GPU tier: what of the RCCL transport can run on ONE GPU -- a world-size-1 `nccl` process group and, next to it, the
communicator the bench uses (rccl.RcclComm: RCCL's C API on the compute stream) with a rank that is its own neighbour.
Runs tools/rccl_selfloop.py in a child process (a process group is per process; pytest keeps none).  The exchange between
DISTINCT ranks needs a multi-GPU node and is not covered here."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_self_loop_process_group_and_direct_communicator():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_selfloop.py")], env=env, capture_output=True, text=True,
                       timeout=300)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert "self-loop batch_isend_irecv on uint8 views: ok" in out
    assert "RcclComm (C API, compute stream): self-loop exchange, all-gather and a 1-rank SlabJob ok" in out
    assert "rccl selfloop ok" in out


def test_bench_multi_rank_plumbing_with_one_rank():
    """bench.py --rehearse-dist: process group, the direct RCCL communicator (agreed by all-reduce), preflight, SlabJob,
    comm statistics and the teardown, with ONE rank -- the code a multi-GPU run goes through before its first exchange."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-dist", "--size", "160", "128", "192", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout: RCCL's banner and the like go to stderr
    d = json.loads(lines[0])
    assert d["comm"]["backend"] == "rccl-direct" and d["comm"]["ranks"] == 1
    assert d["config"]["n_vertices"] > 0 and d["config"]["n_faces"] > 0 and "REHEARSAL" in d["config"]["parallelism"]
    # and with the process group's own collectives
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MASTER_ADDR="127.0.0.1", TOMO_RCCL_DIRECT="0"))
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d2 = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert d2["comm"]["backend"] == "nccl"
    assert (d2["config"]["n_vertices"], d2["config"]["n_faces"]) == (d["config"]["n_vertices"], d["config"]["n_faces"])


def test_rank_supervisor_relays_the_worker_and_retries_after_a_hang():
    """bench.py under a multi-rank launch is a supervisor per rank around a worker child (bench.supervise).  Forced here for the
    one-rank rehearsal (TOMO_BENCH_SUPERVISE=1): the worker's ONE JSON line and exit code come through; and a worker that leaves
    with the watchdog's code 4 (TOMO_BENCH_FAKE_HANG: only while it is on the direct RCCL transport) is replaced once by one on
    torch.distributed's collectives, on the next rendezvous port."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse-dist", "--size", "160", "128", "192", "--steps", "2",
           "--warmup", "1", "--no-cpu-baseline"]
    base = {k: v for k, v in os.environ.items() if k not in ("TOMO_BENCH_WORKER", "TOMO_RCCL_DIRECT")}
    env = dict(base, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", TOMO_BENCH_SUPERVISE="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["comm"]["backend"] == "rccl-direct", lines
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(env, TOMO_BENCH_FAKE_HANG="1"))
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "one retry over torch.distributed's collectives" in p.stderr
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["comm"]["backend"] == "nccl", lines


@pytest.mark.parametrize("world,supervised", [(2, True), (3, False)])
def test_rank_processes_over_gloo_give_the_single_gpu_mesh(world, supervised):
    """bench.py --gpus N --backend gloo: N rank PROCESSES sharing this GPU (three: a rank with BOTH neighbours), halos staged through the host -- the whole
    multi-rank code path of the bench (self-launch, process group, TorchDistComm, preflight, one-exchange front, deferred
    numbering) except the transport; the mesh they report must have the size of the single-GPU mesh of the same stack."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import numpy as np
    from tomography_3d_reconstructor_amd import pipeline
    nzr, ny, nx = 160, 96, 128
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--size", str(nzr), str(ny), str(nx),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TOMO_BENCH_WORKER")}
    # TOMO_BENCH_SUPERVISE=1: every rank process is a supervisor around a worker child, as under an `nccl` launch -- the real
    # process tree (self-launch -> torch.distributed.run -> rank supervisors -> workers), environment and output relayed
    if supervised:
        env["TOMO_BENCH_SUPERVISE"] = "1"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == world and d["comm"]["ranks"] == world and d["comm"]["backend"] == "gloo"
    from tomography_3d_reconstructor_amd import slab
    deferred = slab.DEFERRED_NUMBERING and pipeline.MC3 and pipeline.NA_HINTS            # (A/B switches of the environment)
    assert d["comm"]["numbering"]["redone"] == 0 and (d["comm"]["numbering"]["deferred_passes"] >= 3) == bool(deferred)
    if deferred and pipeline.PACK_CLOSE_FUSED:
        assert d["comm"]["comm_calls_per_pass_per_rank"] == (4.0 if slab.SPLIT_PACK else 5.0)
    dev = torch.device("cuda:0")
    mask = pipeline.ellipsoid_mask(world * nzr, ny, nx, dev).view(torch.uint8)
    v, f = pipeline.extract_surface(pipeline.smooth(pipeline.pack_closed(mask), 3, True), np.full(world * nzr, 1.0), 1.0, 1.0)
    assert (d["config"]["n_vertices"], d["config"]["n_faces"]) == (v.shape[0], f.shape[0])
