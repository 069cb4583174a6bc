#!/bin/bash
# CPU tier under AddressSanitizer + UndefinedBehaviorSanitizer for the native HOST code:
#   oracle/tomo_oracle.c                                  -> oracle/_build/libtomo_oracle_asan.so
#   csrc/host_shim.cpp, obj_writer.cpp, host_hash.cpp     -> tomography_3d_reconstructor_amd/libtomo_host_asan.so
# and the tests that exercise them (oracle vs golden fixtures, OBJ writer incl. its thread pool and the block API through
# the slab job, host MC cell, host checksum, the multi-threaded gather / page-in helpers).  GPU sanitizers do not exist on this pool; device code is covered by the
# parity tests.    usage: bash tools/asan_cpu_tier.sh   (from the repo root; exits non-zero on any sanitizer report)
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
make -s -C tomography_3d_reconstructor_amd/csrc asan
ASAN_LIB=$(gcc -print-file-name=libasan.so)
UBSAN_LIB=$(gcc -print-file-name=libubsan.so)
export LD_PRELOAD="$ASAN_LIB:$UBSAN_LIB"
export ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
export TOMO_ORACLE_LIB="$PWD/oracle/_build/libtomo_oracle_asan.so"
export TOMO_LIB="$PWD/tomography_3d_reconstructor_amd/libtomo_host_asan.so"
export TOMO_HOST_ONLY=1
python -m pytest -q -x -p no:cacheprovider tests/test_oracle_golden.py tests/test_consumers_cpu.py tests/test_host_mc_cell.py \
    tests/test_slab_cpu.py "tests/test_abi.py::test_host_checksum_sees_every_byte" "tests/test_abi.py::test_host_checksum_avx2_and_portable_loops_agree" \
    "tests/test_abi.py::test_devcache_never_returns_a_stale_volume" "tests/test_abi.py::test_host_sha256_with_a_relocatable_state_matches_hashlib" "tests/test_abi.py::test_abi_version_and_error_strings" \
    tests/test_host_helpers.py
