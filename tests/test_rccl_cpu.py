"""CPU tier: the ctypes binding of RCCL's C API (tomography_3d_reconstructor_amd/rccl.py) finds every entry point it calls in
the librccl.so PyTorch-ROCm ships (no communicator is created: that needs a GPU -- tests/test_gpu_rccl.py)."""
import ctypes

from tomography_3d_reconstructor_amd import rccl


def test_librccl_loads_and_exports_what_the_binding_calls():
    L = rccl.lib()
    for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclAllGather",
                 "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
        assert hasattr(L, name), name
    v = ctypes.c_int()
    assert L.ncclGetVersion(ctypes.byref(v)) == 0 and v.value >= 21000          # grouped send / recv exist since 2.7
    assert ctypes.sizeof(rccl._UniqueId) == 128
    assert L.ncclGetErrorString(0) is not None
