"""CPU tier: the C-ABI library builds for gfx950, loads, and exports every symbol include/tomo_hip.h declares."""
import ctypes
import os
import re

from tomography_3d_reconstructor_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tomo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tomo_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    so = _lib.build()
    assert os.path.exists(so)
    L = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), "missing export: " + s
        assert s in _lib.SIGNATURES, "no ctypes signature for " + s
    assert sorted(_lib.SIGNATURES) == syms


def test_abi_version_and_error_strings():
    L = _lib.lib()
    assert L.tomo_abi_version() == 6
    assert L.tomo_error_string(0) == b"ok"
    assert b"argument" in L.tomo_error_string(-1)


def test_geometry_helpers():
    L = _lib.lib()
    assert L.tomo_words_per_row(1024) == 16 and L.tomo_words_per_row(1025) == 17 and L.tomo_words_per_row(1) == 1
    assert L.tomo_field_xorg(1) == 0 and L.tomo_field_xorg(0) == 0
    for nx in (1, 7, 64, 150, 1024, 2048):
        for pad in (0, 1):
            pitch = L.tomo_field_pitch(nx, pad)
            assert pitch % 32 == 0 and pitch >= L.tomo_field_xorg(pad) + nx + 2 * pad
            assert L.tomo_ext_words_per_row(nx, pad) * 64 >= nx + pad + 6 + 4
    assert L.tomo_mc_segments_per_row(1026, 0) == 5 and L.tomo_mc_segments_per_row(32, 0) == 1


def test_argument_checks_do_not_need_a_gpu():
    L = _lib.lib()
    assert L.tomo_pack_bits(None, None, 1, 1, 1, None) == -1
    assert L.tomo_morph_pass(None, None, 4, 4, 4, 0, None) == -1
    assert L.tomo_mc_classify(None, None, 4, 4, 4, 0, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    import pytest
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libtomo_hip.so")
    with pytest.raises(_lib.TomoError):
        _lib.lib()


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the product package (or the drop-in shims) may import,
    include, link or load it."""
    bad = re.compile(r"(^\s*(from|import)\s+oracle\b|from\s+\.+\s*oracle|oracle[/\\]|libtomo_oracle|tomo_oracle|lewiner_luts\.h)",
                     re.M)
    for top in ("tomography_3d_reconstructor_amd", "dropin"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f)).read()
                    assert not bad.search(text), "product file references the oracle: " + os.path.join(dirpath, f)


def test_host_checksum_sees_every_byte():
    """tomo_host_checksum: full content, position dependent, the same for any thread count."""
    import numpy as np
    from tomography_3d_reconstructor_amd import _devcache
    rng = np.random.default_rng(5)
    for shape in [(2, 3, 5), (64, 128, 128), (7, 333, 129), (130, 100, 100)]:
        a = rng.random(shape) < 0.5
        h0 = _devcache.checksum(a)
        assert h0 == _devcache.checksum(a, 1) == _devcache.checksum(a, 3) == _devcache.checksum(a.copy(), 8)
        for off in rng.integers(0, a.size, 24):                  # any single voxel, wherever it sits
            a.reshape(-1)[off] ^= True
            assert _devcache.checksum(a) != h0
            a.reshape(-1)[off] ^= True
        assert _devcache.checksum(a) == h0
    a = rng.random((40, 64, 64)) < 0.5
    h0 = _devcache.checksum(a)
    b = a.copy()
    b[0, 0, 0:8], b[0, 0, 8:16] = a[0, 0, 8:16].copy(), a[0, 0, 0:8].copy()      # two words swapped
    assert (b != a).any() and _devcache.checksum(b) != h0
    assert _devcache.checksum(np.ascontiguousarray(a[:, :, ::-1])) != h0             # same bytes, other places
    assert _devcache.checksum(np.ascontiguousarray(np.roll(a, 1))) != h0
    assert _devcache.checksum(np.zeros(0, bool)) == _devcache.checksum(np.zeros((0, 4), bool))


def test_host_checksum_avx2_and_portable_loops_agree():
    """tomo_host_checksum_impl: the AVX2 stripes and the portable loop give ONE digest (ragged sizes, several threads)."""
    import ctypes
    import numpy as np
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for n in (0, 1, 63, 64, 65, 1023, 1024, 1025, (1 << 20) - 1, 1 << 20, (1 << 20) + 77, 3 * (1 << 20) + 12345):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        o0, o1, o2 = (ctypes.c_uint64 * 2)(), (ctypes.c_uint64 * 2)(), (ctypes.c_uint64 * 2)()
        assert L.tomo_host_checksum_impl(a.ctypes.data if n else None, n, 3, 0, o0) == 0
        assert L.tomo_host_checksum_impl(a.ctypes.data if n else None, n, 1, 1, o1) == 0
        assert L.tomo_host_checksum(a.ctypes.data if n else None, n, 2, o2) == 0
        assert tuple(o0) == tuple(o1) == tuple(o2), n


def test_devcache_never_returns_a_stale_volume(monkeypatch):
    """A cached device copy is used only for an array that is write-protected or verified byte for byte."""
    import numpy as np
    from tomography_3d_reconstructor_amd import _devcache
    _devcache.clear()
    v = object()
    assert _devcache._writeable_default({}) is True                              # the reference's semantics by default
    assert _devcache._writeable_default({"TOMO_READONLY_RESULTS": "1"}) is False and _devcache._writeable_default({"TOMO_WRITEABLE_RESULTS": "0"}) is False
    # (1) TOMO_READONLY_RESULTS: hand-outs are write-protected: trusted while protected, dropped once the caller made them writeable
    monkeypatch.setattr(_devcache, "WRITEABLE_RESULTS", False)
    big = np.ones((128, 128, 128), bool)
    _devcache.put(big, v)
    assert not big.flags.writeable and _devcache.get(big) is v
    try:
        big[64, 64, 65] = False
        raise AssertionError("a protected hand-out accepted a write")
    except ValueError:
        pass
    big.flags.writeable = True                                 # (a NumPy-owned array can be unprotected; a page-locked result cannot)
    big[64, 64, 65] = False                                    # the advisor's single-voxel edit
    assert _devcache.get(big) is None
    # (2) a caller-owned (writeable) array: verified in full at every lookup
    mine = np.ones((128, 128, 128), bool)
    _devcache.put(mine, v, protect=False)
    assert mine.flags.writeable and _devcache.get(mine) is v and _devcache.get(mine) is v
    mine[:, :, 1::32] ^= True                                  # the advisor's strided edit (missed by a sampled checksum)
    assert _devcache.get(mine) is None
    mine2 = np.ones((70, 100, 100), bool)
    _devcache.put(mine2, v, protect=False)
    mine2.reshape(-1)[12345] = False
    assert _devcache.get(mine2) is None
    # (3) the default: results stay writeable like the reference's, every lookup verifies
    monkeypatch.setattr(_devcache, "WRITEABLE_RESULTS", True)
    res = np.zeros((128, 128, 128), bool)
    _devcache.put(res, v)
    assert res.flags.writeable and _devcache.get(res) is v
    res[5, 6, 7] = True
    assert _devcache.get(res) is None
    # (4) a different object with the same content is not the cached one
    assert _devcache.get(np.ones((128, 128, 128), bool)) is None
    _devcache.clear()


def test_host_sha256_with_a_relocatable_state_matches_hashlib():
    """tomo_host_sha256_*: FIPS 180-4 with a caller-held 112-byte state that can be copied / sent between updates (the Z-slab
    job hashes its whole mesh rank after rank).  Both implementations (x86 SHA extensions where present, portable) against
    hashlib, fed in ragged pieces, with the state moved to another buffer in between."""
    import hashlib
    import numpy as np
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for impl in (0, 1):
        for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 1000, 70001):
            data = rng.integers(0, 256, n, dtype=np.uint8)
            st = np.zeros(112, np.uint8)
            assert L.tomo_host_sha256_init(st.ctypes.data) == 0
            pos = 0
            while pos < n:
                k = min(int(rng.integers(1, 300)), n - pos)
                assert L.tomo_host_sha256_update(st.ctypes.data, data[pos:pos + k].ctypes.data, k, impl) == 0
                st = np.frombuffer(st.tobytes(), np.uint8).copy()               # the state is plain bytes
                pos += k
            d = np.zeros(32, np.uint8)
            assert L.tomo_host_sha256_digest(st.ctypes.data, d.ctypes.data) == 0
            assert d.tobytes().hex() == hashlib.sha256(data.tobytes()).hexdigest(), (impl, n)
            d2 = np.zeros(32, np.uint8)                                         # digest() leaves the state usable
            L.tomo_host_sha256_update(st.ctypes.data, data.ctypes.data if n else None, n, impl)
            L.tomo_host_sha256_digest(st.ctypes.data, d2.ctypes.data)
            assert d2.tobytes().hex() == hashlib.sha256(data.tobytes() * 2).hexdigest()
    assert L.tomo_host_sha256_init(None) == -1 and L.tomo_host_sha256_update(None, None, 0, 0) == -1
    st = np.zeros(112, np.uint8)
    L.tomo_host_sha256_init(st.ctypes.data)
    assert L.tomo_host_sha256_update(st.ctypes.data, None, 5, 0) == -1 and L.tomo_host_sha256_digest(st.ctypes.data, None) == -1
