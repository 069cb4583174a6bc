"""ctypes binding of libtomo_hip.so (C ABI declared in include/tomo_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` /
``make -C tomography_3d_reconstructor_amd/csrc``.  There is NO fallback: if the shared object is
missing or a call fails, the product raises -- it never routes through the CPU oracle.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TOMO_LIB") or os.path.join(_HERE, "libtomo_hip.so")   # TOMO_LIB: tuning builds (tools/)
CSRC = os.path.join(_HERE, "csrc")

_c_i = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_p = ctypes.c_void_p
_c_d = ctypes.c_double
_c_f = ctypes.c_float

# name -> (restype, argtypes); must list EVERY symbol of include/tomo_hip.h (checked by tests/test_abi.py)
SIGNATURES = {
    "tomo_abi_version": (_c_i, []),
    "tomo_error_string": (ctypes.c_char_p, [_c_i]),
    "tomo_host_mc_cell": (_c_i, [_c_p, _c_d, _c_p, _c_p]),
    "tomo_host_mc_edge_offset": (_c_d, [_c_d, _c_d]),
    "tomo_host_mc_centre_offset": (None, [_c_p, _c_p]),
    "tomo_host_checksum": (_c_i, [_c_p, _c_i64, _c_i, _c_p]),
    "tomo_host_checksum_impl": (_c_i, [_c_p, _c_i64, _c_i, _c_i, _c_p]),
    "tomo_host_checksum_chunk_bytes": (_c_i64, []),
    "tomo_host_checksum_part": (_c_i, [_c_p, _c_i64, _c_i64, _c_i, _c_i, _c_p]),
    "tomo_host_checksum_fold": (_c_i, [_c_p, _c_i64, _c_i64, _c_p]),
    "tomo_host_touch": (_c_i, [_c_p, _c_i64, _c_i]),
    "tomo_host_gather": (_c_i, [_c_p, _c_i64, _c_i64, _c_p, _c_i]),
    "tomo_host_sha256_init": (_c_i, [_c_p]),
    "tomo_host_sha256_update": (_c_i, [_c_p, _c_p, _c_i64, _c_i]),
    "tomo_host_sha256_digest": (_c_i, [_c_p, _c_p]),
    "tomo_words_per_row": (_c_i64, [_c_i]),
    "tomo_ext_words_per_row": (_c_i64, [_c_i, _c_i]),
    "tomo_ext_rows": (_c_i64, [_c_i, _c_i]),
    "tomo_ext_slices": (_c_i64, [_c_i, _c_i]),
    "tomo_field_pitch": (_c_i64, [_c_i, _c_i]),
    "tomo_field_xorg": (_c_i, [_c_i]),
    "tomo_mc_segments_per_row": (_c_i64, [_c_i, _c_i]),
    "tomo_pack_bits": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "tomo_unpack_bits": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "tomo_popcount": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_slice_popcounts": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_bbox": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_pack_threshold": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "tomo_obj_write": (_c_i, [ctypes.c_char_p, _c_p, _c_i, _c_i64, _c_p, _c_i64, _c_i]),
    "tomo_obj_block_format": (_c_i, [_c_i, _c_p, _c_i64, _c_i, ctypes.POINTER(_c_p), ctypes.POINTER(_c_i64)]),
    "tomo_obj_block_pwrite": (_c_i, [ctypes.c_char_p, _c_i64, _c_p]),
    "tomo_obj_block_free": (None, [_c_p]),
    "tomo_fill_holes_slice": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_fill_holes_ends": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_pack_close_ends": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_pack_close_slab": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "tomo_pack_bits_pair": (_c_i, [_c_p, _c_p, _c_i, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    "tomo_slab_edges": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_i, _c_p, _c_p, _c_p, _c_i, _c_p, _c_p, _c_p, _c_p,
                               _c_i, _c_p, _c_p]),
    "tomo_close_stencil": (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_pack_close_range": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_i, _c_p]),
    "tomo_close_ends_workspace_words": (_c_i64, [_c_i, _c_i, _c_i]),
    "tomo_close_ends_scan": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    "tomo_close_ends_gp": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p]),
    "tomo_morph_pass": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "tomo_morph_fused": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, ctypes.c_uint32, _c_i, _c_p]),
    "tomo_extend_bits": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    "tomo_field_fill": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p]),
    "tomo_field_fill_bits": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p]),
    "tomo_field_span_bytes": (_c_i64, [_c_i, _c_i, _c_i, _c_i]),
    "tomo_field_fill_bits_sparse": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p]),
    "tomo_sign_rows": (_c_i64, [_c_i]),
    "tomo_field_signs_fused": (_c_i, [_c_i]),
    "tomo_sign_buffer_words": (_c_i64, [_c_i, _c_i, _c_i, _c_i]),
    "tomo_field_signs": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i64, _c_i, _c_d, _c_i, _c_i, _c_p, _c_p, _c_p]),
    "tomo_mc_classify": (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p]),
    "tomo_mc_scan_segments": (_c_i, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mc_scan_workspace_bytes": (_c_i64, [_c_i64]),
    "tomo_mc_scan": (_c_i, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mc_list": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p]),
    "tomo_mc_eval": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i64, _c_i, _c_d, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "tomo_mc_list_capped": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mc_eval_capped": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i64, _c_i, _c_d, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "tomo_mc_emit": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i64, _c_i, _c_d, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_p,
                            _c_p, _c_p, _c_p, _c_p]),
    "tomo_mc_first_touch": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i64, _c_i, _c_d, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i, _c_p,
                                   _c_p, _c_p, _c_p, _c_p]),
    "tomo_mc3_list": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "tomo_mc3_eval": (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_i64, _c_i, _c_d, _c_p, _c_i64, _c_p, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p,
                             _c_p, _c_p, _c_p]),
    "tomo_mc3_slice_table_words": (_c_i64, [_c_i, _c_i]),
    "tomo_mc3_sort_segments": (_c_i64, [_c_i, _c_i]),
    "tomo_mc3_scan": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p]),
    "tomo_mc3_vertices": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i,
                                 _c_p, _c_i64, _c_p, _c_i64, _c_f, _c_f, _c_p, _c_p, _c_p, _c_p]),
    "tomo_mc3_sort_workspace_bytes": (_c_i64, [_c_i64, _c_i64]),
    "tomo_mc3_sort_rank": (_c_i, [_c_p, _c_p, _c_p, _c_i64, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mc3_sort_rank_top": (_c_i, [_c_p, _c_p, _c_p, _c_i64, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_f, _c_p]),
    "tomo_mc3_sort_rank_fused": (_c_i, [_c_p, _c_p, _c_i64, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p, _c_f, _c_p]),
    "tomo_mc3_faces": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_vertex_finalize": (_c_i, [_c_p, _c_i64, _c_i, _c_p, _c_i64, _c_p, _c_i64, _c_f, _c_f, _c_p]),
    "tomo_mesh_unique_workspace_bytes": (_c_i64, [_c_i64]),
    "tomo_mesh_unique": (_c_i, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mesh_unique_presorted": (_c_i, [_c_p, _c_p, _c_i64, _c_i, _c_i, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mesh_faces_workspace_bytes": (_c_i64, [_c_i64]),
    "tomo_mesh_lookup": (_c_i, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "tomo_slab_top_rows": (_c_i, [_c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "tomo_slab_lookup": (_c_i, [_c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "tomo_slab_summary": (_c_i, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p]),
    "tomo_slab_lookup_summary": (_c_i, [_c_p, _c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "tomo_mc3_faces_slab": (_c_i, [_c_i, _c_i, _c_i, _c_i, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                   _c_p, _c_i, _c_i, _c_p, _c_i64, _c_i64, _c_p]),
    "tomo_mesh_faces": (_c_i, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_i64, _c_p]),
    "tomo_mesh_faces_direct": (_c_i, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "tomo_mesh_volume_area": (_c_i, [_c_p, _c_p, _c_i64, _c_p, _c_p]),
}

_LIB = None


class TomoError(RuntimeError):
    pass


class TomoUnavailable(TomoError):
    """No MI355X visible / libtomo_hip.so not built.  The one failure the drop-in classes never turn into the
    reference's `return None`: there is no CPU fallback to hide behind."""


def build(force=False):
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp", "Makefile"))]
    srcs.append(os.path.join(_HERE, "..", "include", "tomo_hip.h"))
    stale = force or not os.path.exists(SO_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "-j4"])
    return SO_PATH


def lib():
    """Load libtomo_hip.so (raises if it has not been built -- no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise TomoUnavailable("libtomo_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "or `make -C tomography_3d_reconstructor_amd/csrc` (there is no CPU fallback)")
        L = ctypes.CDLL(SO_PATH)
        host_only = os.environ.get("TOMO_HOST_ONLY", "0") not in ("", "0")      # the sanitizer build holds the host entry points only
        for name, (res, args) in SIGNATURES.items():
            if host_only and not hasattr(L, name):
                continue
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(code, what):
    if code != 0:
        raise TomoError("%s failed: %s (%d)" % (what, lib().tomo_error_string(code).decode(), code))
