"""CPU tier: bench.py's launch decision (taken before anything touches the GPU) and its workload table."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (imports neither torch nor the HIP library at module level)


def test_bench_module_import_is_gpu_free():
    code = "import sys; sys.path.insert(0, %r); import bench; assert 'torch' not in sys.modules, 'bench imported torch'" % ROOT
    assert subprocess.run([sys.executable, "-c", code]).returncode == 0


def test_single_gpu_runs_in_process():
    argv = ["--steps", "5"]
    assert bench.launch_plan(bench.parse(argv), {}, argv) == ("run", 1)


def test_gpus_n_without_world_size_spawns_n_ranks():
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2", "--workload", "cfg5"]
    kind, cmd = bench.launch_plan(bench.parse(argv), {}, argv)
    assert kind == "spawn"
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == argv                      # the ranks get this command line unchanged


def test_under_torchrun_be_a_rank_or_refuse():
    argv = ["--gpus", "4"]
    assert bench.launch_plan(bench.parse(argv), {"WORLD_SIZE": "4", "RANK": "1"}, argv) == ("run", 4)
    kind, msg = bench.launch_plan(bench.parse(argv), {"WORLD_SIZE": "2"}, argv)
    assert kind == "refuse" and "--gpus 4" in msg and "WORLD_SIZE=2" in msg
    # --gpus 1 (the default) under a multi-rank launch is a mistake too, not a silent single-GPU run
    assert bench.launch_plan(bench.parse([]), {"WORLD_SIZE": "8"}, [])[0] == "refuse"


def test_refusal_exit_code(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.main(["--gpus", "8"]) == 2


@pytest.mark.parametrize("argv,world,total,scaling", [
    ([], 1, (1024, 1024, 1024), "weak"),
    ([], 8, (8192, 1024, 1024), "weak"),
    (["--workload", "cfg4"], 2, (2048, 1024, 1024), "strong"),
    (["--workload", "cfg4"], 4, (2048, 1024, 1024), "strong"),
    (["--workload", "cfg5"], 8, (4096, 2048, 2048), "strong"),
    (["--size", "256", "512", "512"], 2, (512, 512, 512), "weak"),
    (["--size", "256", "512", "512", "--strong"], 2, (256, 512, 512), "strong"),
])
def test_workloads(argv, world, total, scaling):
    shape, sc, _ = bench.workload_shape(bench.parse(argv), world)
    assert shape == total and sc == scaling


def test_rank_supervisor_retries_once_over_the_process_group():
    """bench.supervise: a worker that gives up with code 4 (the watchdog's: no progress for two minutes) is replaced ONCE by a
    worker on torch.distributed's collectives and a fresh rendezvous; any other exit code is handed on unchanged."""
    calls = []

    def fake(codes):
        def run(cmd, env):
            calls.append((list(cmd), dict(env)))
            return codes.pop(0)
        return run
    env = {"RANK": "3", "WORLD_SIZE": "8", "MASTER_PORT": "29511", "TORCHELASTIC_USE_AGENT_STORE": "True"}
    assert bench.supervise(["x"], env, fake([4, 0]), port_wait_s=0.1) == 0       # (rank 0 never published: the old + 1)
    assert len(calls) == 2 and calls[0][1]["TOMO_BENCH_WORKER"] == "1" and "TOMO_RCCL_DIRECT" not in calls[0][1]
    assert calls[1][1]["TOMO_RCCL_DIRECT"] == "0" and calls[1][1]["MASTER_PORT"] == "29512"
    assert "TORCHELASTIC_USE_AGENT_STORE" not in calls[1][1] and calls[1][1]["TOMO_BENCH_WORKER"] == "1"
    del calls[:]
    assert bench.supervise(["x"], env, fake([4, 4]), port_wait_s=0.1) == 4 and len(calls) == 2          # one retry, not a loop
    del calls[:]
    assert bench.supervise(["x"], env, fake([1])) == 1 and len(calls) == 1             # a crash is not a hang
    del calls[:]
    assert bench.supervise(["x"], dict(env, TOMO_RCCL_DIRECT="0"), fake([4])) == 4 and len(calls) == 1   # nothing left to fall back to
    assert env == {"RANK": "3", "WORLD_SIZE": "8", "MASTER_PORT": "29511", "TORCHELASTIC_USE_AGENT_STORE": "True"}


def test_retry_port_is_probed_by_rank_0_and_shared():
    """the retry's rendezvous port is one rank 0's supervisor found FREE, and every other rank's supervisor (same
    agent, same node) reads that very port instead of assuming MASTER_PORT + 1."""
    import socket
    busy = socket.socket()
    busy.bind(("127.0.0.1", 0))
    busy.listen(1)
    base = busy.getsockname()[1] - 1                       # MASTER_PORT + 1 is taken
    env0 = {"RANK": "0", "MASTER_PORT": str(base), "TORCHELASTIC_RUN_ID": "t%d" % os.getpid()}
    try:
        p0 = bench.retry_port(env0)
        assert p0 > base + 1                               # skipped the busy one
        s = socket.socket()
        s.bind(("127.0.0.1", p0))                          # and it is free
        s.close()
        assert bench.retry_port(dict(env0, RANK="5"), wait_s=5.0) == p0
    finally:
        busy.close()


def test_golden_lookup_and_the_counter_file_tied_to_the_kernel_source(tmp_path, monkeypatch):
    """bench.expected_hashes: the fixture a run's parity_in_run is checked against (reference-derived up to 1024^3, oracle-derived
    for BASELINE configs[3] / [4], none for other stacks); bench.field_pmc_for_this_build: counter bytes are used only for the
    build of csrc/field.hip they were measured on."""
    import hashlib
    import json
    e = bench.expected_hashes((1024, 1024, 1024))
    assert e and "ellipsoid_hashes.json (reference)" in e["source"] and (e["nv"], e["nf"]) == (3538048, 7076092) and len(e["v"]) == 64
    e4, e5 = bench.expected_hashes((2048, 1024, 1024)), bench.expected_hashes((4096, 2048, 2048))
    assert e4 and e5 and "oracle-derived" in e4["source"] and "oracle-derived" in e5["source"] and e5["nv"] == 23876144
    assert bench.expected_hashes((8192, 1024, 1024)) is None and bench.expected_hashes((100, 100, 100)) is None
    p, b = bench.field_pmc_for_this_build()                       # the committed r04 file matches the committed kernel
    src = os.path.join(ROOT, "tomography_3d_reconstructor_amd", "csrc", "field.hip")
    assert p is not None and b > 4.5e9
    assert json.load(open(os.path.join(ROOT, p)))["field_hip_sha256"] == hashlib.sha256(open(src, "rb").read()).hexdigest()
    stale = tmp_path / "pmc.json"
    stale.write_text(json.dumps({"field_hip_sha256": "0" * 64, "hbm_bytes_per_launch": 1.0}))
    monkeypatch.setattr(bench, "FIELD_PMCS", [str(stale)])
    assert bench.field_pmc_for_this_build() == (None, None)       # a file of another build is not used


def test_cpu_baseline_all_core_leg_counts_like_the_one_thread_run():
    """bench.cpu_baseline: the all-core leg (Z-chunks with 13 spare slices either side, a process pool) finds exactly the
    vertices and triangles of the one-thread run on the whole stack."""
    r = bench.cpu_baseline(96, workers=3)
    assert r["cores"] == 1 and r["kind"] == "port" and r["all_cores"]["cores"] == 3
    assert (r["all_cores"]["n_vertices"], r["all_cores"]["n_faces"]) == (r["n_vertices"], r["n_faces"]) and r["n_vertices"] > 0
    assert "counts equal to the one-thread run: True" in r["all_cores"]["sample"]
