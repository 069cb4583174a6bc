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


def _sha(t):
    import hashlib
    import numpy as np
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


@pytest.mark.parametrize("world,supervised,size,ns_size", [(2, True, (128, 256, 256), (64, 128, 128)), (3, False, (160, 96, 128), (96, 80, 112))])
def test_rank_processes_over_gloo_give_the_single_gpu_mesh(world, supervised, size, ns_size):
    """bench.py --gpus N --backend gloo: N rank PROCESSES sharing this GPU (three: a rank with BOTH neighbours), halos staged
    through the host -- the whole multi-rank code path of the bench (self-launch, process group, TorchDistComm, preflight,
    one-exchange front, deferred numbering, the SHA-256 chain over the ranks, the second block on the strong-scaling stack)
    except the transport.  The run checks ITSELF (`parity_in_run`): against the reference-derived fixture where the total stack
    has one (world 2: 256^3 and 64x128x128; world 3: 96x80x112), else against a single-GPU pass on rank 0; and here the BYTES
    are compared once more: the digests the ranks formed together equal the SHA-256 of this process's single-GPU tensors."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import numpy as np
    from tomography_3d_reconstructor_amd import pipeline
    nzr, ny, nx = size
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--size", str(nzr), str(ny), str(nx),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--north-star-size"] + [str(x) for x in ns_size]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "TOMO_BENCH_WORKER")}
    # TOMO_BENCH_SUPERVISE=1: every rank process is a supervisor around a worker child, as under an `nccl` launch -- the real
    # process tree (self-launch -> torch.distributed.run -> rank supervisors -> workers), environment and output relayed
    if supervised:
        env["TOMO_BENCH_SUPERVISE"] = "1"
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == world and d["comm"]["ranks"] == world and d["comm"]["backend"] == "gloo"
    from tomography_3d_reconstructor_amd import slab
    deferred = slab.DEFERRED_NUMBERING and pipeline.MC3 and pipeline.NA_HINTS            # (A/B switches of the environment)
    assert d["comm"]["numbering"]["redone"] == 0 and (d["comm"]["numbering"]["deferred_passes"] >= 3) == bool(deferred)
    if deferred and pipeline.PACK_CLOSE_FUSED and nzr >= slab.MERGED_MIN:
        assert d["comm"]["comm_calls_per_pass_per_rank"] == (4.0 if slab.SPLIT_PACK else 5.0)
    dev = torch.device("cuda:0")

    def single(shape):
        z, y, x = shape
        mask = pipeline.ellipsoid_mask(z, y, x, dev).view(torch.uint8)
        return pipeline.extract_surface(pipeline.smooth(pipeline.pack_closed(mask), 3, True), np.full(z, 1.0), 1.0, 1.0)
    # the headline block
    v, f = single((world * nzr, ny, nx))
    assert d["parity_in_run"] is True, d["parity"]
    assert (d["config"]["n_vertices"], d["config"]["n_faces"]) == (v.shape[0], f.shape[0])
    assert d["parity"]["vertices_f32_sha256"] == _sha(v) and d["parity"]["faces_i64_sha256"] == _sha(f)
    if world == 2:
        assert "ellipsoid_hashes.json (reference)" in d["parity"]["parity_against"]          # 256^3: the reference's own run
    else:
        assert "single-GPU pass" in d["parity"]["parity_against"]
    # the strong-scaling block: a fixture AND the single-GPU pass of the same run (the N = 1 denominator)
    ns = d["north_star_scaling"]
    v2, f2 = single(ns_size)
    assert ns["scaling"] == "strong" and ns["n_gpus"] == world and ns["parity_in_run"] is True, ns
    assert "ellipsoid_hashes.json (reference)" in ns["parity_against"] and ns["equal_to_single_gpu_pass_in_this_run"] is True
    assert ns["vertices_f32_sha256"] == _sha(v2) and ns["faces_i64_sha256"] == _sha(f2)
    assert ns["n1_ms_per_step_same_run"] > 0 and ns["per_rank_efficiency_vs_n1"] > 0


def test_a_run_whose_mesh_is_wrong_says_so_and_fails():
    """parity_in_run is a CHECK: with a fixture that does not describe the stack (TOMO_BENCH_GOLDEN_OVERRIDE: a copy whose
    face digest is wrong) the run prints parity_in_run false and leaves with a non-zero code."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import json
    import tempfile
    good = json.load(open(os.path.join(ROOT, "tests", "golden", "ellipsoid_hashes.json")))
    bad = {"96x80x112": dict(good["96x80x112"], faces_i64_sha256="0" * 64)}
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fh:
        json.dump(bad, fh)
    try:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--size", "96", "80", "112", "--steps", "2", "--warmup", "1",
               "--no-cpu-baseline", "--no-extras"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, TOMO_BENCH_GOLDEN_OVERRIDE=fh.name))
        d = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
        assert p.returncode == 5 and d["parity_in_run"] is False and "PARITY FAILED" in p.stderr
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=os.environ)
        d = json.loads([x for x in p.stdout.splitlines() if x.startswith("{")][-1])
        assert p.returncode == 0 and d["parity_in_run"] is True and "reference" in d["parity"]["parity_against"]
    finally:
        os.unlink(fh.name)
