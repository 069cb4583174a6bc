"""Probe: can two RCCL ranks share the one GPU of a gpurun box?  (If yes, slab.TorchDistComm can be rehearsed
over the real nccl backend there; if RCCL refuses duplicate devices this prints the error and exits 0.)"""
import os
import sys
import torch
import torch.distributed as td

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    try:
        td.init_process_group("nccl", device_id=dev)
        t = torch.full((4,), float(rank), device=dev)
        td.all_reduce(t)
        torch.cuda.synchronize()
        print("rank", rank, "all_reduce ok", t.tolist(), flush=True)
        if rank == 0:
            td.send(torch.arange(8, device=dev), 1)
        else:
            r = torch.zeros(8, dtype=torch.int64, device=dev)
            td.recv(r, 0)
            print("recv ok", r.tolist(), flush=True)
        if len(sys.argv) > 1 and sys.argv[1] == "slab":
            import numpy as np
            from tomography_3d_reconstructor_amd import pipeline, slab
            gz, ny, nx = 96, 80, 144
            job = slab.SlabJob(gz, ny, nx, slab.TorchDistComm(dev))
            mask = pipeline.ellipsoid_mask(gz, ny, nx, dev, job.z0, job.z1).view(torch.uint8)
            depths = np.full(gz, 1.0)
            v, f = job.run(mask, depths, 1.0, 1.0)
            full = pipeline.ellipsoid_mask(gz, ny, nx, dev).view(torch.uint8)
            vol = pipeline.smooth(pipeline.close_ends(pipeline.pack(full)), 3, True)
            V, F = pipeline.extract_surface(vol, depths, 1.0, 1.0)
            vs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
            td.all_gather(vs, torch.tensor([v.shape[0]], device=dev))
            off = job.vertex_offset
            assert torch.equal(V[off:off + v.shape[0]], v), "vertices differ"
            fs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
            td.all_gather(fs, torch.tensor([f.shape[0]], device=dev))
            foff = 0 if rank == 0 else int(fs[0].item())
            assert torch.equal(F[foff:foff + f.shape[0]], f), "faces differ"
            print("rank", rank, "slab over nccl == single GPU:", v.shape[0], "verts", f.shape[0], "faces", flush=True)
        td.destroy_process_group()
    except Exception as ex:  # noqa: BLE001
        print("rank", rank, "nccl on a shared GPU failed:", type(ex).__name__, str(ex)[:400], flush=True)


if __name__ == "__main__":
    main()
