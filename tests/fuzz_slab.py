"""Ad-hoc fuzz of the Z-slab job against the single-GPU path on random stacks: rank threads on this GPU (in-process
communicator), random world sizes, slab thicknesses on both sides of the one-exchange front's limit, noise / blobs / smoothed
noise, slice depths with zeros, three passes per job (exact, deferred, deferred or redone).  run: python tests/fuzz_slab.py
[seed] [cases]"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tomography_3d_reconstructor_amd import pipeline, slab  # noqa: E402

dev = torch.device("cuda:0")


def run(seed=0, cases=20):
    """-> number of mismatches (0 expected); prints one summary line."""
    rng = np.random.default_rng(seed)
    bad = 0
    stats_seen = {}


    def single(v, depths, my, mx):
        mask = torch.from_numpy(np.ascontiguousarray(v).view(np.uint8)).to(dev)
        vol = pipeline.smooth(pipeline.close_ends(pipeline.pack(mask)), 3, True)
        got = pipeline.extract_surface(vol, depths, my, mx)
        if got is None:
            return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int64)
        return got[0].cpu().numpy(), got[1].cpu().numpy()


    for it in range(cases):
        world = int(rng.integers(2, 5))
        thick = int(rng.integers(11, 40)) if it % 2 else int(rng.integers(slab.MERGED_MIN, slab.MERGED_MIN + 24))
        nz = world * thick + int(rng.integers(0, world))
        ny = int(rng.integers(8, 48))
        nx = 16 * int(rng.integers(1, 6)) if it % 3 else int(rng.integers(8, 90))          # nx % 16 != 0: the unfused front
        kind = it % 3
        if kind == 0:
            v = rng.random((nz, ny, nx)) < 0.25 + 0.5 * rng.random()
        elif kind == 1:
            zz, yy, xx = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
            v = ((zz - nz / 2) / (nz * 0.47)) ** 2 + ((yy - ny / 2) / (ny * 0.4 + 1)) ** 2 + ((xx - nx / 2) / (nx * 0.42 + 1)) ** 2 <= 1
            v ^= rng.random(v.shape) < 0.02
        else:
            v = rng.random((nz, ny, nx)) < 0.55
            v[:, : ny // 3] = False
        volumes = [v, v, v if it % 4 else (v ^ (rng.random(v.shape) < 0.3))]
        depths = rng.random(nz) * 0.9 + 0.1
        if it % 5 == 0:
            depths[rng.integers(0, nz, 3)] = 0.0
        my, mx = float(rng.random() + 0.5), float(rng.random() + 0.5)
        refs = [single(w, depths, my, mx) for w in (volumes[0], volumes[2])]
        refs = [refs[0], refs[0], refs[1]]
        out = [[None] * world for _ in volumes]
        stats, errs = [None] * world, []

        def target(c):
            try:
                job = slab.SlabJob(nz, ny, nx, c)
                with torch.cuda.stream(torch.cuda.Stream()):
                    masks = [torch.from_numpy(np.ascontiguousarray(w[job.z0:job.z1]).view(np.uint8)).to(dev) for w in volumes]
                    pend = None
                    for p in range(len(volumes) + 1):
                        # odd cases: pass p + 1 is submitted BEFORE the host reads pass p (SlabJob.submit / result)
                        nxt = job.submit(masks[p], depths, my, mx) if p < len(volumes) else None
                        if not it % 2 and nxt is not None:
                            job.result(nxt)
                            got = (p, job.mesh)
                        elif it % 2 and pend is not None:
                            job.result(pend[1])
                            got = (pend[0], job.mesh)
                        else:
                            got = None
                        pend = (p, nxt) if nxt is not None else None
                        if got is not None:
                            verts, faces = got[1]
                            torch.cuda.current_stream().synchronize()
                            out[got[0]][c.rank] = (verts.cpu().numpy(), faces.cpu().numpy(), job.n_vertices_global)
                stats[c.rank] = (job.deferred_passes, job.deferred_redone)
            except BaseException as e:   # noqa: BLE001
                errs.append(e)
                raise

        ts = [threading.Thread(target=target, args=(c,)) for c in slab.ThreadComm.make(world)]
        [t.start() for t in ts]
        [t.join(300) for t in ts]
        ok = not errs
        if ok:
            for per_rank, (rv, rf) in zip(out, refs):
                sv = np.concatenate([o[0] for o in per_rank])
                sf = np.concatenate([o[1] for o in per_rank])
                ok &= sv.shape == rv.shape and sv.tobytes() == rv.tobytes() and sf.shape == rf.shape and bool(np.array_equal(sf, rf))
                ok &= per_rank[0][2] == rv.shape[0]
            ok &= all(s == stats[0] for s in stats)
            stats_seen[stats[0]] = stats_seen.get(stats[0], 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH case", it, "world", world, "shape", (nz, ny, nx), "kind", kind, "errors", errs[:1], "stats", stats, flush=True)
    print("slab fuzz: %d cases, %d mismatches; (deferred, redone) per job: %s; counters %s" % (cases, bad, stats_seen, pipeline.COUNTERS))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 20) else 0)
