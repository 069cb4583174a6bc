"""GPU tier: the Z-slab pass WITHOUT host round trips (slab.SlabJob._numbering_deferred) -- from the second pass of a job
on, the chain runs from size hints, the shared-plane rows travel in fixed-capacity messages, the global index table is
built from device-side counts and ONE download ends the pass.  Rank threads share this GPU (in-process communicator);
every pass of every scenario must give the single-GPU mesh byte for byte, whichever way it went (deferred, redone, exact)."""
import threading

import numpy as np
import pytest
import torch

from oracle import oracle as O
from tomography_3d_reconstructor_amd import pipeline, slab

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def single_gpu(v, depths, mm_y, mm_x, dev):
    mask = torch.from_numpy(np.ascontiguousarray(v).view(np.uint8)).to(dev)
    vol = pipeline.smooth(pipeline.close_ends(pipeline.pack(mask)), 3, True)
    got = pipeline.extract_surface(vol, depths, mm_y, mm_x)
    if got is None:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
    return got[0].cpu().numpy(), got[1].cpu().numpy()


def run_passes(world, volumes, depths, mm_y, mm_x, dev, z_cuts=None):
    """One SlabJob per rank thread, one pass per entry of `volumes` -> ([(verts, faces, n_global) per pass], [job stats])."""
    nz, ny, nx = volumes[0].shape
    out = [[None] * world for _ in volumes]
    stats, errs = [None] * world, []
    bar = threading.Barrier(world)

    def target(c):
        try:
            job = slab.SlabJob(nz, ny, nx, c, z_cuts=z_cuts)
            with torch.cuda.stream(torch.cuda.Stream()):
                for p, v in enumerate(volumes):
                    mask = torch.from_numpy(np.ascontiguousarray(v[job.z0:job.z1]).view(np.uint8)).to(dev)
                    verts, faces = job.run(mask, depths, mm_y, mm_x)
                    torch.cuda.current_stream().synchronize()
                    out[p][c.rank] = (verts.cpu().numpy(), faces.cpu().numpy(), job.vertex_offset, job.n_vertices_global)
                    bar.wait()
            stats[c.rank] = (job.deferred_passes, job.deferred_redone)
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            bar.abort()
            raise

    ts = [threading.Thread(target=target, args=(c,)) for c in slab.ThreadComm.make(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    return out, stats


def check_pass(per_rank, ref):
    rv, rf = ref
    verts = np.concatenate([o[0] for o in per_rank])
    faces = np.concatenate([o[1] for o in per_rank])
    assert per_rank[0][3] == rv.shape[0]
    offs = np.cumsum([0] + [o[0].shape[0] for o in per_rank])
    assert [o[2] for o in per_rank] == [int(x) for x in offs[:-1]]
    assert verts.shape == rv.shape and verts.tobytes() == rv.tobytes()
    assert faces.shape == rf.shape and np.array_equal(faces, rf)


def blob(nz, ny, nx, scale, seed):
    rng = np.random.default_rng(seed)
    zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    v = ((zz - nz / 2) / (nz * 0.48 * scale)) ** 2 + ((yy - ny / 2) / (ny * 0.42 * scale)) ** 2 + ((xx - nx / 2) / (nx * 0.45 * scale)) ** 2 <= 1
    v ^= rng.random(v.shape) < 0.004
    return v


@pytest.mark.parametrize("world", [2, 3, 4])
def test_deferred_pass_matches_single_gpu(dev, world):
    """Same stack three times: pass 1 is the exact pass (it sets the hints and the message capacities), passes 2 and 3 run
    deferred on every rank -- all three byte-identical to the single-GPU mesh (variable depths, anisotropic pixels)."""
    if not (pipeline.MC3 and pipeline.NA_HINTS and slab.DEFERRED_NUMBERING):
        pytest.skip("needs the mc3 chain with size hints")
    nz, ny, nx = 128, 96, 144
    v = blob(nz, ny, nx, 1.0, 5)
    depths = np.concatenate([np.full(40, 0.5), np.full(48, 0.25), np.full(40, 0.75)])
    ref = single_gpu(v, depths, 0.7, 0.9, dev)
    out, stats = run_passes(world, [v, v, v], depths, 0.7, 0.9, dev)
    for per_rank in out:
        check_pass(per_rank, ref)
    assert stats == [(2, 0)] * world, stats


def test_deferred_pass_redone_when_the_surface_outgrows_its_hints(dev):
    """A job whose stack changes between passes: the small body sets the hints, the large one overflows them (chain buffers
    and the shared-plane message) -- every rank goes through the exact pass again, together -- and the pass after that is
    deferred again; so is a pass whose upper slab holds no surface at all (zero counts are counts)."""
    if not (pipeline.MC3 and pipeline.NA_HINTS and slab.DEFERRED_NUMBERING):
        pytest.skip("needs the mc3 chain with size hints")
    nz, ny, nx = 120, 112, 128
    small, large = blob(nz, ny, nx, 0.45, 1), blob(nz, ny, nx, 1.0, 2)
    low = np.zeros_like(large)                            # a body in the lower slab alone (no flat cut: a cut face of ~8 000 vertices
    low[:nz // 2 - 8] = blob(nz // 2 - 8, ny, nx, 1.0, 3)   # between two planes would send the sort to its library path: tested below)
    depths = np.full(nz, 0.4)
    volumes = [small, small, large, large, low, low]
    refs = [single_gpu(v, depths, 1.0, 1.0, dev) for v in volumes]
    out, stats = run_passes(2, volumes, depths, 1.0, 1.0, dev)
    for per_rank, ref in zip(out, refs):
        check_pass(per_rank, ref)
    # passes: exact | deferred | deferred attempt, redone | deferred | deferred (rank 1: no surface) | deferred
    assert stats == [(4, 1), (4, 1)], stats


def test_job_with_an_empty_slab_from_the_start_stays_on_the_exact_pass(dev):
    """A rank that has no surface in the exact pass has no size hints: the ranks agree (all-gather) not to defer."""
    nz, ny, nx = 120, 112, 128
    low = blob(nz, ny, nx, 1.0, 2)
    low[nz // 2 - 8:] = False
    depths = np.full(nz, 0.4)
    ref = single_gpu(low, depths, 1.0, 1.0, dev)
    out, stats = run_passes(2, [low, low, low], depths, 1.0, 1.0, dev)
    for per_rank in out:
        check_pass(per_rank, ref)
    assert stats == [(0, 0), (0, 0)], stats


def test_deferred_pass_with_coinciding_rows(dev):
    """Zero slice depths make vertex rows coincide: the rows do not ascend strictly, the general sort decides (np.unique
    semantics) and every deferred attempt is redone -- the mesh is still the single-GPU one."""
    if not (pipeline.MC3 and pipeline.NA_HINTS and slab.DEFERRED_NUMBERING):
        pytest.skip("needs the mc3 chain with size hints")
    nz, ny, nx = 96, 64, 80
    v = np.stack(O.ellipsoid_masks(nz, ny, nx))
    depths = np.full(nz, 0.3)
    depths[30:34] = 0.0
    depths[70] = 0.0
    ref = single_gpu(v, depths, 1.0, 1.0, dev)
    out, stats = run_passes(3, [v, v, v], depths, 1.0, 1.0, dev)
    for per_rank in out:
        check_pass(per_rank, ref)
    assert all(s == stats[0] for s in stats) and stats[0][0] + stats[0][1] == 2, stats


def test_slab_kernels_take_counts_from_device_memory(dev):
    """tomo_slab_top_rows / lookup / summary on hand-made inputs against NumPy (the mapping to global indices inside the triangle
    kernel is covered by every deferred pass of this file)."""
    e = slab.HipEngine()
    rng = np.random.default_rng(4)
    nv, nt, cap_v, cap = 500, 37, 640, 64
    rows = np.unique(rng.integers(0, 40, (4000, 3)).astype(np.float32), axis=0)[:nv]
    rows[nv - nt:, 0] = 99.0                                             # the top plane closes the sorted list
    uniq = torch.zeros((cap_v, 3), dtype=torch.float32, device=dev)
    uniq[:nv] = torch.from_numpy(rows).to(dev)
    tot = torch.tensor([123, nv, 77, 0, 0, 0, 0, nt], dtype=torch.int64, device=dev)
    msg = e.slab_top_rows(uniq, tot, cap_v, cap).cpu().numpy()
    assert msg[:1].view(np.uint32)[0] == nt and msg[1] == 0 and msg[2] == 0
    assert np.array_equal(msg[3:3 + 3 * nt].reshape(-1, 3), rows[nv - nt:]) and not msg[3 + 3 * nt:].any()
    # the rank above: the same rows sit somewhere in ITS list, one of them is unknown to it
    up_rows = np.unique(np.concatenate([rows[nv - nt:], rng.integers(100, 140, (300, 3)).astype(np.float32)]), axis=0)
    nu2 = len(up_rows)
    uniq2 = torch.zeros((nu2 + 50, 3), dtype=torch.float32, device=dev)
    uniq2[:nu2] = torch.from_numpy(up_rows).to(dev)
    tot2 = torch.tensor([1, nu2, 1, 0, 0, 0, 0, 0], dtype=torch.int64, device=dev)
    m2 = msg.copy()
    m2[3 + 3 * 5] = 98.5                                                 # row 5 is not there
    idx, miss = e.slab_lookup(uniq2, tot2, nu2 + 50, torch.from_numpy(m2).to(dev), cap)
    idx = idx.cpu().numpy()
    want = np.array([np.flatnonzero((up_rows == r).all(1))[0] for r in rows[nv - nt:]])
    assert int(miss.item()) == 1 and idx[5] == -1 and np.array_equal(np.delete(idx[:nt], 5), np.delete(want, 5))
    assert (idx[nt:] == -1).all()
    s = e.slab_summary(tot, cap_v, None, None, cap, 0).cpu().numpy()
    assert list(s) == [nv - nt, 0, 0, nv, nt, 0, 123, 77]
    s2 = e.slab_summary(tot2, nu2 + 50, torch.from_numpy(m2).to(dev), miss, 0, 1).cpu().numpy()
    assert list(s2) == [nu2, 1, 8, nu2, 0, nt, 1, 1]
    assert int(miss.item()) == 0                                           # read once, cleared for the next pass
    tot_small = tot.clone()
    assert int(e.slab_summary(tot_small, cap_v, None, None, nt - 1, 0)[2].item()) == 4        # message capacity too small
    tot_small[4] = 3
    assert int(e.slab_summary(tot_small, cap_v, None, None, cap, 0)[2].item()) == 2           # rows not strictly ascending
    tot_small[3] = 1
    assert int(e.slab_summary(tot_small, cap_v, None, None, cap, 0)[2].item()) & 1            # a chain buffer overflowed


@pytest.mark.parametrize("world,nz", [(2, 300), (3, 391)])
def test_merged_edge_exchange_matches_single_gpu(dev, world, nz):
    """Slabs of at least MERGED_MIN slices take the one-exchange front: ORIGINAL edge slices travel while the middle of the
    slab is packed + closed, every rank closes its neighbours' halo slices itself.  Noisy stack with holes in both end
    slices and structure right at the slab boundaries; two passes (exact, deferred), meshes and closed / smoothed volumes
    byte-identical to the single-GPU path."""
    ny, nx = 40, 80
    assert nz // world >= slab.MERGED_MIN
    rng = np.random.default_rng(nz)
    v = blob(nz, ny, nx, 1.0, nz)
    v ^= rng.random(v.shape) < 0.01
    v[0, 4:30, 6:70] = True
    v[0, 10:20, 20:40] = False
    v[nz - 1, 8:36, 10:60] = True
    v[nz - 1, 15:25, 25:45] = False
    for r in range(1, world):                                            # alternating slices across every slab boundary
        zb = slab.slab_range(nz, r, world)[0]
        v[zb - 14:zb + 14:2, 5:35, 5:75] = True
        v[zb - 13:zb + 14:2, 5:35, 5:75] = rng.random((14, 30, 70)) < 0.5
    depths = np.full(nz, 0.5)
    ref = single_gpu(v, depths, 1.0, 1.0, dev)
    mask = torch.from_numpy(v.view(np.uint8)).to(dev)
    created = pipeline.close_ends(pipeline.pack(mask))
    smoothed = pipeline.smooth(created, 3, True)
    ref_counts = (pipeline.slice_counts(created).cpu().numpy(), pipeline.slice_counts(smoothed).cpu().numpy())
    counts = [None] * world
    orig_run = slab.SlabJob.run

    def run_and_count(self, *a, **k):
        out = orig_run(self, *a, **k)
        counts[self.rank] = (self.slice_counts("created"), self.slice_counts("smoothed"))
        return out

    slab.SlabJob.run = run_and_count
    try:
        out, stats = run_passes(world, [v, v], depths, 1.0, 1.0, dev)
    finally:
        slab.SlabJob.run = orig_run
    for per_rank in out:
        check_pass(per_rank, ref)
    for c in counts:
        assert np.array_equal(c[0], ref_counts[0]) and np.array_equal(c[1], ref_counts[1])
    if pipeline.MC3 and pipeline.NA_HINTS and slab.DEFERRED_NUMBERING:
        assert stats == [(1, 0)] * world, stats


def test_merged_front_with_a_misaligned_mask_on_one_rank(dev):
    """The one-exchange front is decided jointly; a rank whose mask sits at an odd address (a view) copies it instead of
    falling back to a path that exchanges differently."""
    world, nz, ny, nx = 2, 264, 24, 48
    v = blob(nz, ny, nx, 1.0, 11)
    depths = np.full(nz, 1.0)
    ref = single_gpu(v, depths, 1.0, 1.0, dev)
    out, errs = [None] * world, []

    def target(c):
        try:
            job = slab.SlabJob(nz, ny, nx, c)
            with torch.cuda.stream(torch.cuda.Stream()):
                part = np.ascontiguousarray(v[job.z0:job.z1]).view(np.uint8)
                if c.rank == 1:                                   # 3 bytes into a larger buffer: not 16-byte aligned
                    flat = torch.zeros(part.size + 64, dtype=torch.uint8, device=dev)
                    mask = flat[3:3 + part.size].view(part.shape)
                    mask.copy_(torch.from_numpy(part).to(dev))
                    assert mask.data_ptr() % 16 != 0
                else:
                    mask = torch.from_numpy(part).to(dev)
                verts, faces = job.run(mask, depths, 1.0, 1.0)
                torch.cuda.current_stream().synchronize()
            out[c.rank] = (verts.cpu().numpy(), faces.cpu().numpy(), job.vertex_offset, job.n_vertices_global)
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            raise

    ts = [threading.Thread(target=target, args=(c,)) for c in slab.ThreadComm.make(world)]
    [t.start() for t in ts]
    [t.join(120) for t in ts]
    assert not errs, errs
    check_pass(out, ref)


def run_pipelined(world, volumes, depths, mm_y, mm_x, dev):
    """run_passes with the host reading every pass ONE PASS LATE: submit(k + 1) before result(k) (what bench.py does)."""
    nz, ny, nx = volumes[0].shape
    out = [[None] * world for _ in volumes]
    stats, errs = [None] * world, []

    def target(c):
        try:
            job = slab.SlabJob(nz, ny, nx, c)
            with torch.cuda.stream(torch.cuda.Stream()):
                masks = [torch.from_numpy(np.ascontiguousarray(v[job.z0:job.z1]).view(np.uint8)).to(dev) for v in volumes]
                pend = None
                for p in range(len(volumes) + 1):
                    nxt = job.submit(masks[p], depths, mm_y, mm_x) if p < len(volumes) else None
                    if pend is not None:
                        verts, faces = job.result(pend)
                        assert job.mesh[0] is verts and job.smoothed is not None
                        out[p - 1][c.rank] = (verts.cpu().numpy(), faces.cpu().numpy(), job.vertex_offset, job.n_vertices_global)
                    pend = nxt
                torch.cuda.current_stream().synchronize()
            stats[c.rank] = (job.deferred_passes, job.deferred_redone)
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            raise

    ts = [threading.Thread(target=target, args=(c,)) for c in slab.ThreadComm.make(world)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    return out, stats


@pytest.mark.parametrize("world", [2, 3])
def test_submit_before_result_gives_the_same_meshes(dev, world):
    """SlabJob.submit / result: pass k + 1 is enqueued BEFORE the host reads the counters of pass k.  Stacks that differ from
    pass to pass (so a stale buffer or a result published under the wrong pass would show), incl. one that outgrows the
    hints while the next pass is already in flight: every pass equals the single-GPU mesh of ITS stack, the numbering
    (vertex_offset / n_vertices_global) belongs to the pass that result() returned."""
    if not (pipeline.MC3 and pipeline.NA_HINTS and slab.DEFERRED_NUMBERING):
        pytest.skip("needs the mc3 chain with size hints")
    nz, ny, nx = 128, 96, 144
    a, b, big = blob(nz, ny, nx, 0.8, 1), blob(nz, ny, nx, 0.8, 2), blob(nz, ny, nx, 1.0, 3)
    volumes = [a, a, b, a, big, b, b]
    depths = np.concatenate([np.full(40, 0.5), np.full(48, 0.25), np.full(40, 0.75)])
    refs = {id(v): single_gpu(v, depths, 0.7, 0.9, dev) for v in (a, b, big)}
    out, stats = run_pipelined(world, volumes, depths, 0.7, 0.9, dev)
    for per_rank, v in zip(out, volumes):
        check_pass(per_rank, refs[id(v)])
    assert all(s == stats[0] for s in stats) and stats[0][0] >= 3, stats       # deferred passes happened, alike on every rank


def test_extract_surface_submit_reads_one_pass_late(dev):
    """pipeline.extract_surface_submit: the single-GPU counterpart -- two surfaces in flight, results read in order, each
    equal to the plain call on its own volume; None where the reference returns None."""
    nz, ny, nx = 96, 80, 112
    vols = [blob(nz, ny, nx, s, k) for k, s in enumerate((0.7, 1.0, 0.7, 0.9))] + [np.zeros((nz, ny, nx), bool)]
    depths = np.full(nz, 0.6)
    dv = [pipeline.smooth(pipeline.close_ends(pipeline.pack(torch.from_numpy(v.view(np.uint8)).to(dev))), 3, True) for v in vols]
    plain = [pipeline.extract_surface(x, depths, 0.7, 0.9) for x in dv]
    got, pend = [], None
    for x in dv + [None]:
        nxt = pipeline.extract_surface_submit(x, depths, 0.7, 0.9) if x is not None else None
        if pend is not None:
            got.append(pend.result())
        pend = nxt
    assert plain[-1] is None and got[-1] is None
    for p, g in zip(plain[:-1], got[:-1]):
        assert torch.equal(p[0], g[0]) and torch.equal(p[1], g[1])


def test_host_waits_of_a_multi_rank_pass_have_a_deadline():
    """RCCL's C API has no timeout and its calls sit in the compute stream, so every host wait of a multi-rank pass goes
    through pipeline.wait_event: a stream that does not get there in time becomes a TomoError instead of a hang."""
    import time
    from tomography_3d_reconstructor_amd import _lib
    dev = torch.device("cuda:0")
    t = torch.arange(8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    torch.cuda._sleep(int(1.5e9))                      # ~0.6-1 s of a spinning kernel in front of the copy
    pend = pipeline.PendingDownload(t)
    t0 = time.monotonic()
    with pytest.raises(_lib.TomoError, match="did not arrive"):
        pipeline.wait_event(pend._event, 0.05, "the counters of a pass")
    assert time.monotonic() - t0 < 0.5
    assert pend.wait(30.0) == list(range(8))           # with a sane deadline the same download simply arrives
    job = slab.SlabJob(64, 32, 64, type("C", (), {"rank": 0, "world": 2})())
    job._await(dev)                                    # an idle stream: returns at once


def test_a_cut_face_too_long_for_the_sort_kernel_takes_the_library_path(dev):
    """Round 4: the unique stage is one hand-written kernel that keeps a sort segment in LDS (4 096 entries).  A flat cut face --
    thousands of vertices between the same two planes -- does not fit: the kernel says so (bit 8 of tot[3]), the pass is repeated
    with the library's segmented sort and the geometry stays on that path.  Single GPU (first pass, then from the size hints) and
    a 2-rank job whose deferred pass meets the cut: bytes-equal to the oracle's / the single-GPU mesh every time."""
    if not (pipeline.MC3 and pipeline.FUSED_SORT):
        pytest.skip("needs the mc3 chain with the fused sort")
    nz, ny, nx = 120, 112, 128
    large = blob(nz, ny, nx, 1.0, 2)
    cut = large.copy()
    cut[nz // 2 - 8:] = False                              # the lower half of the body: a cut face of ~8 000 vertices
    depths = np.full(nz, 0.4)
    pipeline._MC3_LARGE.clear()
    c0 = dict(pipeline.COUNTERS)
    ref = single_gpu(cut, depths, 1.0, 1.0, dev)
    ose = O.SurfaceExtractor()
    ovp = O.VoxelProcessor()
    osm = ovp.smooth_voxel_data(ovp.create_voxel_data(list(cut), True, 0, nz, 0), 3, True)
    ev, ef = ose.extract_manifold_surface(osm, depths, 1.0, 1.0)
    assert ref[0].tobytes() == ev.tobytes() and np.array_equal(ref[1], ef)
    assert pipeline.COUNTERS.get("mc3_sort_library", 0) > c0.get("mc3_sort_library", 0) and any(pipeline._MC3_LARGE.values())
    again = single_gpu(cut, depths, 1.0, 1.0, dev)          # hinted pass: straight to the library path
    assert again[0].tobytes() == ev.tobytes() and np.array_equal(again[1], ef)
    small = blob(nz, ny, nx, 0.45, 1)
    volumes = [small, small, cut, cut, small]
    refs = [single_gpu(v, depths, 1.0, 1.0, dev) for v in volumes]
    pipeline._MC3_LARGE.clear()
    out, stats = run_passes(2, volumes, depths, 1.0, 1.0, dev)
    for per_rank, r in zip(out, refs):
        check_pass(per_rank, r)
    assert all(s[1] >= 1 for s in stats), stats             # the pass that met the cut was redone, on every rank


def test_slabs_of_unequal_thickness_on_the_hip_engine(dev):
    """SlabJob(z_cuts=...) (round 4: slabs of equal WORK): three ranks with 150 / 21 / 213 slices -- a thin slab between two thick
    ones, on either side of the one-exchange front's 128-slice limit -- exact pass, then two deferred ones: bytes-equal to the
    single-GPU mesh; and the cuts work_balanced_cuts derives from a pass are the same on every rank."""
    if not (pipeline.MC3 and pipeline.NA_HINTS and slab.DEFERRED_NUMBERING):
        pytest.skip("needs the mc3 chain with size hints")
    nz, ny, nx = 384, 96, 144
    v = blob(nz, ny, nx, 1.0, 9)
    depths = np.concatenate([np.full(100, 0.5), np.full(184, 0.25), np.full(100, 0.75)])
    ref = single_gpu(v, depths, 0.7, 0.9, dev)
    for cuts in ([0, 150, 171, 384], [0, 130, 258, 384]):
        out, stats = run_passes(3, [v, v, v], depths, 0.7, 0.9, dev, z_cuts=cuts)
        for per_rank in out:
            check_pass(per_rank, ref)
        assert stats == [(2, 0)] * 3, stats
    got = [None] * 3
    errs = []

    def target(c):
        try:
            job = slab.SlabJob(nz, ny, nx, c)
            with torch.cuda.stream(torch.cuda.Stream()):
                mask = torch.from_numpy(np.ascontiguousarray(v[job.z0:job.z1]).view(np.uint8)).to(dev)
                job.run(mask, depths, 0.7, 0.9)
                got[c.rank] = (job.work_balanced_cuts(depths, 0.3, align=4), job.slice_vertex_counts(depths))
        except BaseException as e:   # noqa: BLE001
            errs.append(e)
            raise
    ts = [threading.Thread(target=target, args=(c,)) for c in slab.ThreadComm.make(3)]
    [t.start() for t in ts]
    [t.join(300) for t in ts]
    assert not errs, errs
    assert got[0][0] == got[1][0] == got[2][0] and got[0][0][0] == 0 and got[0][0][-1] == nz
    assert np.array_equal(got[0][1], got[2][1]) and int(got[0][1].sum()) == len(ref[0])
    thick = np.diff(got[0][0])
    assert thick[1] < thick[0] and thick[1] < thick[2]              # the middle of the body holds the most surface per slice
