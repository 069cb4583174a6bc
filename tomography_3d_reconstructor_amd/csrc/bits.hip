// bits.hip -- bit-packed binary stages on gfx950: pack/unpack, popcount, 2-D fill-holes,
// the close-ends z recurrence, 6-neighbour erosion/dilation, and the halo-extended bit volume
// the field kernel reads.  All of these are HBM-bound byte/bit work: one bit per voxel, uint64
// words along x, coalesced word loads, no LDS needed except for block reductions.
//
// Reference lines replaced: voxel_processor.py:46 (stack), :51 (np.sum), :56-77 (close ends),
// :79-97 (binary_opening + binary_closing with the 3-D cross, erosion border_value=True).
#include <stdlib.h>
#include <atomic>
#include "tomo_common.h"

// ------------------------------------------------------------------------------------------
// geometry helpers (host)
TOMO_API int64_t tomo_words_per_row(int nx) { return ((int64_t)nx + 63) / 64; }
TOMO_API int64_t tomo_ext_words_per_row(int nx, int pad) { (void)pad; return ((int64_t)nx + 16 + 63) / 64; }
TOMO_API int64_t tomo_ext_rows(int ny, int pad) { return (int64_t)ny + 2 * pad + 4; }
TOMO_API int64_t tomo_ext_slices(int nz, int pad) { return (int64_t)nz + 2 * pad + 4; }
// Field rows: padded column X is float column xorg + X = X; the pitch is a whole number of 128-byte lines (the field
// kernel writes whole lines only; a 1026-column row costs 33 lines).
TOMO_API int tomo_field_xorg(int pad) { (void)pad; return 0; }
TOMO_API int64_t tomo_field_pitch(int nx, int pad)
{
    int64_t cols = (int64_t)nx + 2 * pad;
    return (cols + 31) / 32 * 32;
}
// marching-cubes segments are 256 float COLUMNS of a field row, segment s = columns [256 s - 224, 256 s + 32)
// rows per (slice, segment) block of sign records: Ny rounded up so that 16-row groups are 512-byte aligned
TOMO_API int64_t tomo_sign_rows(int Ny) { return ((int64_t)Ny + 15) / 16 * 16; }
TOMO_API int64_t tomo_mc_segments_per_row(int Nx, int xorg) { return ((int64_t)Nx + xorg + SEG_SHIFT + 255) / 256; }

// ------------------------------------------------------------------------------------------
// pack: one wave per group of 16 words (1024 voxels) of a row; lane L reads the byte of voxel
// 64k + L for k = 0..15 (64 contiguous bytes per wave-load), the wave ballot IS the word.
__global__ __launch_bounds__(256) void pack_kernel(const uint8_t *__restrict__ mask, u64 *__restrict__ bits,
                                                   int64_t rows, int nx, int wx, int groups)
{
    const int lane = threadIdx.x & 63;
    int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= rows * groups) return;
    int64_t row = wid / groups;
    int g = (int)(wid - row * groups);
    const uint8_t *src = mask + row * (int64_t)nx;
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int x = (g * 16 + k) * 64 + lane;
        uint8_t b = x < nx ? src[x] : (uint8_t)0;
        u64 m = __ballot(b != 0);
        if (lane == k) mine = m;
    }
    int w = g * 16 + lane;
    if (lane < 16 && w < wx) bits[row * (int64_t)wx + w] = mine;
}

// fast variant for rows that are 16-byte aligned (nx % 16 == 0): lane L loads the 16 bytes of voxels
// 16L..16L+15 of a 1024-voxel group (1 KiB per wave-load), turns them into 16 mask bits with integer
// arithmetic, and four neighbouring lanes combine their pieces into one 64-bit word.
__device__ static inline u32 nonzero_nibble(u32 w)
{   // bit i of the result = (byte i of w != 0)
    u32 h = (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
    return ((h >> 7) * 0x10204080u) >> 28;
}

#ifndef PACK_U
#define PACK_U 4        // 1024-voxel units per wave, their 16-byte loads in flight together.  Plain loads: 1: 4.8, 2: 5.2, 4: 5.0,
                        // 8: 4.6 TB/s; nontemporal loads (the mask is read exactly once): 1: 4.9, 2: 5.6, 4: 6.1 TB/s
#endif
__device__ __forceinline__ void pack16_body(const uint8_t *__restrict__ mask, u64 *__restrict__ bits, int64_t rows, int nx,
                                            int wx, int groups)
{
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t units = rows * groups, u0 = wid * PACK_U;
    if (u0 >= units) return;
    uint4 v[PACK_U];
#pragma unroll
    for (int i = 0; i < PACK_U; i++) {
        const int64_t u = u0 + i;
        const int64_t row = u / groups;
        const int x = (int)(u - row * groups) * 1024 + lane * 16;
        v[i] = make_uint4(0, 0, 0, 0);
        if (u < units && x < nx) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            const u4 t = __builtin_nontemporal_load((const u4 *)(mask + row * (int64_t)nx + x));
            v[i] = make_uint4(t.x, t.y, t.z, t.w);
        }
    }
#pragma unroll
    for (int i = 0; i < PACK_U; i++) {
        const int64_t u = u0 + i;
        if (u >= units) break;
        const int64_t row = u / groups;
        const int g = (int)(u - row * groups);
        u32 piece = nonzero_nibble(v[i].x) | (nonzero_nibble(v[i].y) << 4) | (nonzero_nibble(v[i].z) << 8) |
                    (nonzero_nibble(v[i].w) << 12);
        u64 w = (u64)piece << (16 * (lane & 3));
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        int word = g * 16 + (lane >> 2);
        if ((lane & 3) == 0 && word < wx) bits[row * (int64_t)wx + word] = w;
    }
}

__global__ __launch_bounds__(256) void pack16_kernel(const uint8_t *__restrict__ mask, u64 *__restrict__ bits,
                                                     int64_t rows, int nx, int wx, int groups)
{
    pack16_body(mask, bits, rows, nx, wx, groups);
}

// The two end slices of a stack in ONE launch (blockIdx.y = 0 / 1: slice 0 / nz - 1), and the 16 control words of the
// fill-holes kernel that runs next are cleared on the way (saves a launch and a 128-byte memset in front of every pass).
__global__ __launch_bounds__(256) void pack16_ends_kernel(const uint8_t *__restrict__ mask, u64 *__restrict__ bits, int64_t rows,
                                                          int nx, int wx, int groups, int64_t mask_step, int64_t bits_step,
                                                          u64 *__restrict__ ctrl)
{
    if (ctrl && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 16) ctrl[threadIdx.x] = 0ull;
    pack16_body(mask + (int64_t)blockIdx.y * mask_step, bits + (int64_t)blockIdx.y * bits_step, rows, nx, wx, groups);
}

// Two independent stacks of slices in ONE launch (blockIdx.y = 0 / 1): the ORIGINAL edge slices a Z-slab rank sends to its
// two neighbours (slab.py) -- two launches of 6 us each with nothing to overlap them otherwise.
__global__ __launch_bounds__(256) void pack16_pair_kernel(const uint8_t *__restrict__ maskA, u64 *__restrict__ bitsA, int64_t rowsA,
                                                          const uint8_t *__restrict__ maskB, u64 *__restrict__ bitsB, int64_t rowsB,
                                                          int nx, int wx, int groups)
{
    if (blockIdx.y == 0) pack16_body(maskA, bitsA, rowsA, nx, wx, groups);
    else pack16_body(maskB, bitsB, rowsB, nx, wx, groups);
}

TOMO_API int tomo_pack_bits_pair(const uint8_t *maskA, uint64_t *bitsA, int nzA, const uint8_t *maskB, uint64_t *bitsB, int nzB,
                                 int ny, int nx, void *stream)
{
    if (!maskA || !bitsA || !maskB || !bitsB || nzA <= 0 || nzB <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    if (nx % 16 != 0 || (((uintptr_t)maskA) & 15) != 0 || (((uintptr_t)maskB) & 15) != 0) return TOMO_E_ARG;
    const int wx = (int)tomo_words_per_row(nx), groups = (wx + 15) / 16;
    const int64_t rowsA = (int64_t)nzA * ny, rowsB = (int64_t)nzB * ny, rows = rowsA > rowsB ? rowsA : rowsB;
    const int64_t blocks = ceil_div64(ceil_div64(rows * groups, PACK_U), 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(pack16_pair_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, (hipStream_t)stream, maskA, (u64 *)bitsA, rowsA,
                       maskB, (u64 *)bitsB, rowsB, nx, wx, groups);
    return tomo_status();
}

TOMO_API int tomo_pack_bits(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, void *stream)
{
    if (!mask || !bits || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    int groups = (wx + 15) / 16;
    int64_t rows = (int64_t)nz * ny;
    int64_t waves = rows * groups;
    int64_t blocks = ceil_div64(waves, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    if (nx % 16 == 0 && (((uintptr_t)mask) & 15) == 0)
        hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)ceil_div64(ceil_div64(waves, PACK_U), 4)), dim3(256), 0,
                           (hipStream_t)stream, mask, (u64 *)bits, rows, nx, wx, groups);
    else
        hipLaunchKernelGGL(pack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, (u64 *)bits, rows,
                           nx, wx, groups);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------
// pack + the z recurrence of _close_volume_ends in ONE pass over the mask (voxel_processor.py:46, :72-75).
// The recurrence c'[z] = c[z] | (c'[z-1] & c[z+1]), z = 1 .. nz-2 (in place, ascending: c'[z-1] already updated, c[z+1] not
// yet) is a LOCAL three-tap stencil: where c[z] = 0 the updated c'[z-1] = c[z-1] | (c'[z-2] & c[z]) is just c[z-1], and
// where c[z] = 1 the result is 1 anyway -- so c'[z] = c[z] | (c[z-1] & c[z+1]) with c[0], c[nz-1] the FILLED end slices
// (checked against the recurrence on random volumes, tests/test_oracle_golden.py).  Hence: pack and fill the two end slices
// first (they are 2 of nz), then one streaming kernel packs every other slice and applies the stencil on the fly -- a wave
// marches a run of PC_ZR slices of its (row, 1024-voxel group) with a three-word window, so the mask is read (PC_ZR + 2) /
// PC_ZR times and the separate close-ends passes over the bit volume (4 launches, 66 us at 1024^3) disappear.
#ifndef PC_ZR
#define PC_ZR 32        // slices per wave run (the mask is read (PC_ZR + 2) / PC_ZR times)
#endif
#ifndef PC_U
#define PC_U 4          // slices whose 16-byte loads are in flight together per lane
#endif
#ifndef PC_PIPELINED
#define PC_PIPELINED 1  // 0: every run through the generic loop (A/B)
#endif
// Z-slab form: the stencil's neighbours of the first / last slice of a slab are slices of the ranks below / above, handed
// in bit-packed (`below`, `above`: (ny, wx) words each, or null); a slab that holds a GLOBAL end slice has it packed and
// filled in `bits` already (lo_fixed / hi_fixed) and does not recompute it.  Outputs slices za .. zb-1.
__device__ __forceinline__ void pack_close_body(const int64_t wid, const uint8_t *__restrict__ mask, u64 *__restrict__ bits, int nz, int ny,
                                                int nx, int wx, int groups, int runs, int za, int zb, int lo_fixed,
                                                int hi_fixed, const u64 *__restrict__ below,
                                                const u64 *__restrict__ above, int zr)
{
    const int lane = threadIdx.x & 63;
    const int64_t per_run = (int64_t)ny * groups;
    if (wid >= per_run * runs) return;
    const int r = (int)(wid / per_run);
    const int64_t rem = wid - (int64_t)r * per_run;
    const int y = (int)(rem / groups), g = (int)(rem - (int64_t)y * groups);
    const int x = g * 1024 + lane * 16, word = g * 16 + (lane >> 2);
    const bool inx = x < nx, inw = word < wx;
    const int z0 = za + r * zr, z1 = (z0 + zr < zb) ? z0 + zr : zb;                          // outputs z0 .. z1-1
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int64_t slice_bytes = (int64_t)ny * nx, slice_words = (int64_t)ny * wx;
    const uint8_t *mp = mask + (int64_t)y * nx + x;
    u64 *bp = bits + (int64_t)y * wx + word;
    // the 64-bit word of slice z for this lane's group of four lanes: an end slice comes from `bits` (packed and filled
    // before this kernel), any other one from the mask
    auto from_raw = [&](u4 t) -> u64 {
        const u32 piece = nonzero_nibble(t.x) | (nonzero_nibble(t.y) << 4) | (nonzero_nibble(t.z) << 8) | (nonzero_nibble(t.w) << 12);
        u64 w = (u64)piece << (16 * (lane & 3));
        w |= __shfl_xor(w, 1, 64);
        w |= __shfl_xor(w, 2, 64);
        return w;
    };
    auto in_bits = [&](int z) -> bool { return (z == 0 && lo_fixed) || (z == nz - 1 && hi_fixed); };
    auto raw_of = [&](int z) -> u4 {
        u4 t = {0u, 0u, 0u, 0u};
        if (z >= 0 && z < nz && !in_bits(z) && inx) t = __builtin_nontemporal_load((const u4 *)(mp + (int64_t)z * slice_bytes));
        return t;
    };
    auto word_of = [&](int z, u4 t) -> u64 {
        if (z < 0) return (below && inw) ? below[(int64_t)y * wx + word] : 0ull;
        if (z >= nz) return (above && inw) ? above[(int64_t)y * wx + word] : 0ull;
        if (in_bits(z)) return inw ? bp[(int64_t)z * slice_words] : 0ull;
        return from_raw(t);
    };
    // Fast path (all runs but the first and the last of a stack): every slice z0-1 .. z1 comes from the mask and the run is a
    // whole number of groups of PC_U slices -- straight-line code, the loads of the NEXT group are issued before the current
    // one is reduced and stored, so no wave ever waits for its own stores before it may load again (the generic loop below
    // keeps its loads under the end-slice conditions and is left with s_waitcnt vmcnt(0) at the top of every iteration).
    // Lanes past the row end read the row's first 16 bytes instead and contribute nothing.
    if (PC_PIPELINED && z0 >= 1 && z1 <= nz - 1 && !in_bits(z0 - 1) && !in_bits(z1) && (z1 - z0) % PC_U == 0 && z1 > z0) {
        const uint8_t *p = mask + (int64_t)y * nx + (inx ? x : 0) + (int64_t)(z0 - 1) * slice_bytes;
        auto word_fast = [&](u4 t) -> u64 {
            const u32 piece = nonzero_nibble(t.x) | (nonzero_nibble(t.y) << 4) | (nonzero_nibble(t.z) << 8) | (nonzero_nibble(t.w) << 12);
            u64 w = (u64)(inx ? piece : 0u) << (16 * (lane & 3));
            w |= __shfl_xor(w, 1, 64);
            w |= __shfl_xor(w, 2, 64);
            return w;
        };
        const u4 t_prev = __builtin_nontemporal_load((const u4 *)p);
        const u4 t_cur = __builtin_nontemporal_load((const u4 *)(p + slice_bytes));
        p += 2 * slice_bytes;
        u4 t[PC_U];
#pragma unroll
        for (int j = 0; j < PC_U; j++) t[j] = __builtin_nontemporal_load((const u4 *)(p + (int64_t)j * slice_bytes));
        p += (int64_t)PC_U * slice_bytes;
        u64 prev = word_fast(t_prev), cur = word_fast(t_cur);
        u64 *q = bp + (int64_t)z0 * slice_words;
        const bool st = (lane & 3) == 0 && inw;
        const int groups_z = (z1 - z0) / PC_U;
        for (int gz = 0; gz + 1 < groups_z; gz++) {
            u4 n[PC_U];
#pragma unroll
            for (int j = 0; j < PC_U; j++) n[j] = __builtin_nontemporal_load((const u4 *)(p + (int64_t)j * slice_bytes));
            p += (int64_t)PC_U * slice_bytes;
#pragma unroll
            for (int j = 0; j < PC_U; j++) {
                const u64 next = word_fast(t[j]);
                const u64 out = cur | (prev & next);
                if (st) q[(int64_t)j * slice_words] = out;
                prev = cur;
                cur = next;
            }
            q += (int64_t)PC_U * slice_words;
#pragma unroll
            for (int j = 0; j < PC_U; j++) t[j] = n[j];
        }
#pragma unroll
        for (int j = 0; j < PC_U; j++) {
            const u64 next = word_fast(t[j]);
            const u64 out = cur | (prev & next);
            if (st) q[(int64_t)j * slice_words] = out;
            prev = cur;
            cur = next;
        }
        return;
    }
    u64 prev = word_of(z0 - 1, raw_of(z0 - 1));
    u64 cur = word_of(z0, raw_of(z0));
    for (int z = z0; z < z1; z += PC_U) {
        u4 t[PC_U];
#pragma unroll
        for (int j = 0; j < PC_U; j++) t[j] = z + j < z1 ? raw_of(z + 1 + j) : (u4){0u, 0u, 0u, 0u};   // PC_U slices in flight
#pragma unroll
        for (int j = 0; j < PC_U; j++) {
            if (z + j >= z1) break;
            const u64 next = word_of(z + 1 + j, t[j]);
            const u64 out = cur | (prev & next);
            if ((lane & 3) == 0 && inw) bp[(int64_t)(z + j) * slice_words] = out;
            prev = cur;
            cur = next;
        }
    }
}

__global__ __launch_bounds__(256) void pack_close_kernel(const uint8_t *__restrict__ mask, u64 *__restrict__ bits, int nz, int ny,
                                                         int nx, int wx, int groups, int runs, int za, int zb, int lo_fixed,
                                                         int hi_fixed, const u64 *__restrict__ below,
                                                         const u64 *__restrict__ above, int zr)
{
    pack_close_body((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), mask, bits, nz, ny, nx, wx, groups, runs, za, zb,
                    lo_fixed, hi_fixed, below, above, zr);
}

// ---- the same pass with the boundary words of a run HANDED OVER inside the workgroup (round 4): the four waves of a workgroup
// take four CONSECUTIVE runs of one (row, 1024-voxel group) column; a wave loads only its own slices, writes the outputs whose
// three taps it holds as it marches, and at the end the first and the last word of every run pass through LDS: the first / last
// output of a run is written after one barrier.  Only the two outer waves of a workgroup read a neighbour's slice, so the mask
// is read (4 zr + 2) / (4 zr) times instead of (zr + 2) / zr (1.016 against 1.0625 at zr = 32).  Outputs za .. zb - 1, zb - za a
// multiple of PC_U, split into `nwg` workgroups per column and four runs each, a whole number of groups of PC_U slices per run
// (the first `extra` runs one group longer).  `head` (< PC_U) more slices in FRONT of za -- what (range) mod PC_U leaves -- are
// written by the first run's wave as well (it knows the word in front of them from the start; a separate launch of the generic
// kernel for them cost 6 us).  End slices as in pack_close_body.
__global__ __launch_bounds__(256) void pack_close_ho_kernel(const uint8_t *__restrict__ mask, u64 *__restrict__ bits, int nz, int ny,
                                                            int nx, int wx, int groups, int nwg, int za, int zb, int lo_fixed,
                                                            int hi_fixed, const u64 *__restrict__ below,
                                                            const u64 *__restrict__ above, int head)
{
    __shared__ u64 s_edge[4][2][16];                               // [wave][first / last word of its run][word of the group]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t cols = (int64_t)ny * groups;
    const int c = (int)(blockIdx.x / cols);                        // consecutive workgroups: consecutive columns of one z range
    const int64_t rem = blockIdx.x - (int64_t)c * cols;
    const int y = (int)(rem / groups), g = (int)(rem - (int64_t)y * groups);
    const int x = g * 1024 + lane * 16, word = g * 16 + (lane >> 2);
    const bool inx = x < nx, inw = word < wx;
    const int G = (zb - za) / PC_U, nruns = 4 * nwg, base = G / nruns, extra = G - base * nruns;
    const int q = 4 * c + w;
    const int z0 = za + PC_U * (q * base + (q < extra ? q : extra)), groups_z = base + (q < extra ? 1 : 0), z1 = z0 + PC_U * groups_z;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const int64_t slice_bytes = (int64_t)ny * nx, slice_words = (int64_t)ny * wx;
    const uint8_t *p = mask + (int64_t)y * nx + (inx ? x : 0) + (int64_t)z0 * slice_bytes;
    auto word_fast = [&](u4 t) -> u64 {
        const u32 piece = nonzero_nibble(t.x) | (nonzero_nibble(t.y) << 4) | (nonzero_nibble(t.z) << 8) | (nonzero_nibble(t.w) << 12);
        u64 v = (u64)(inx ? piece : 0u) << (16 * (lane & 3));
        v |= __shfl_xor(v, 1, 64);
        v |= __shfl_xor(v, 2, 64);
        return v;
    };
    // the slice beyond the workgroup's range, for its first and its last wave only: another slice of the mask, a filled end slice
    // (bits), a neighbour rank's slice (below / above) or nothing
    u64 outer = 0ull;
    u4 t_outer = {0u, 0u, 0u, 0u};
    bool outer_raw = false;
    const bool with_head = head > 0 && q == 0;                     // (uniform per wave)
    if (w == 0 || w == 3) {
        const int zo = w == 0 ? z0 - 1 - (with_head ? head : 0) : z1;
        if (zo < 0) outer = (below && inw) ? below[(int64_t)y * wx + word] : 0ull;
        else if (zo >= nz) outer = (above && inw) ? above[(int64_t)y * wx + word] : 0ull;
        else if ((zo == 0 && lo_fixed) || (zo == nz - 1 && hi_fixed)) outer = inw ? bits[(int64_t)zo * slice_words + (int64_t)y * wx + word] : 0ull;
        else { t_outer = __builtin_nontemporal_load((const u4 *)(mask + (int64_t)y * nx + (inx ? x : 0) + (int64_t)zo * slice_bytes)); outer_raw = true; }
    }
    u4 t_head[PC_U - 1];
#pragma unroll
    for (int j = 0; j < PC_U - 1; j++)                             // the slices z0 - head .. z0 - 1 (the last `head` entries; the others repeat)
        t_head[j] = with_head ? __builtin_nontemporal_load((const u4 *)(p - (int64_t)(j < head ? head - j : 1) * slice_bytes))
                              : (u4){0u, 0u, 0u, 0u};
    u4 t[PC_U];
#pragma unroll
    for (int j = 0; j < PC_U; j++) t[j] = __builtin_nontemporal_load((const u4 *)(p + (int64_t)j * slice_bytes));
    p += (int64_t)PC_U * slice_bytes;
    u64 *qo = bits + (int64_t)y * wx + word + (int64_t)z0 * slice_words;
    const bool st = (lane & 3) == 0 && inw;
    // first group: words 0 .. 3; outputs 1, 2 (output 0 waits for the hand-over, output 3 for word 4)
    u64 w0, w1, prev, cur;
    {
        u4 n[PC_U];
        const bool more = groups_z > 1;
        if (more) {
#pragma unroll
            for (int j = 0; j < PC_U; j++) n[j] = __builtin_nontemporal_load((const u4 *)(p + (int64_t)j * slice_bytes));
            p += (int64_t)PC_U * slice_bytes;
        }
        w0 = word_fast(t[0]); w1 = word_fast(t[1]);
        prev = w0; cur = w1;
#pragma unroll
        for (int j = 2; j < PC_U; j++) {
            const u64 next = word_fast(t[j]);
            if (st) qo[(int64_t)(j - 1) * slice_words] = cur | (prev & next);
            prev = cur; cur = next;
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < PC_U; j++) t[j] = n[j];
        }
    }
    // groups 1 .. groups_z - 1: word k arrives -> output k - 1
    u64 *qq = qo + (int64_t)(PC_U - 1) * slice_words;              // next output to write: index PC_U - 1
    for (int gz = 1; gz < groups_z; gz++) {
        u4 n[PC_U];
        const bool more = gz + 1 < groups_z;
        if (more) {
#pragma unroll
            for (int j = 0; j < PC_U; j++) n[j] = __builtin_nontemporal_load((const u4 *)(p + (int64_t)j * slice_bytes));
            p += (int64_t)PC_U * slice_bytes;
        }
#pragma unroll
        for (int j = 0; j < PC_U; j++) {
            const u64 next = word_fast(t[j]);
            if (st) qq[(int64_t)j * slice_words] = cur | (prev & next);
            prev = cur; cur = next;
        }
        qq += (int64_t)PC_U * slice_words;
        if (more) {
#pragma unroll
            for (int j = 0; j < PC_U; j++) t[j] = n[j];
        }
    }
    // prev, cur = the last two words of the run; qq points at the last output
    if (outer_raw) outer = word_fast(t_outer);
    if (with_head) {
        // slices z0 - head .. z0 - 1: h[0 .. head); in front of them `outer`, behind them w0.  Written here; the run's own first
        // output then has h[head - 1] in front of it.
        u64 h[PC_U - 1];
#pragma unroll
        for (int j = 0; j < PC_U - 1; j++) h[j] = word_fast(t_head[j]);          // h[j] = slice z0 - head + j for j < head
        u64 pv = outer;
#pragma unroll
        for (int j = 0; j < PC_U - 1; j++) {
            if (j < head) {
                const u64 nxw = j + 1 < head ? h[j + 1] : w0;
                if (st) qo[(int64_t)(j - head) * slice_words] = h[j] | (pv & nxw);
                pv = h[j];
            }
        }
        outer = pv;
    }
    if ((lane & 3) == 0) { s_edge[w][0][lane >> 2] = w0; s_edge[w][1][lane >> 2] = cur; }
    __syncthreads();
    const u64 before = w == 0 ? outer : s_edge[w - 1][1][lane >> 2];
    const u64 after = w == 3 ? outer : s_edge[w + 1][0][lane >> 2];
    if (st) {
        qo[0] = w0 | (before & w1);
        qq[0] = cur | (prev & after);
    }
}

// unpack: one thread per 8 voxels (one byte of the word) -> 8 output bytes.
__global__ __launch_bounds__(256) void unpack_kernel(const u64 *__restrict__ bits, uint8_t *__restrict__ mask,
                                                     int64_t rows, int nx, int wx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int bytes_per_row = (nx + 7) / 8;
    if (i >= rows * bytes_per_row) return;
    int64_t row = i / bytes_per_row;
    int g = (int)(i - row * bytes_per_row);
    u64 w = bits[row * (int64_t)wx + (g >> 3)];
    u64 b = (w >> ((g & 7) * 8)) & 0xffull;
    // spread the 8 bits of b into 8 bytes (0/1)
    u64 x = (b * 0x0101010101010101ull) & 0x8040201008040201ull;
    x = ((x + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
    uint8_t *dst = mask + row * (int64_t)nx + (int64_t)g * 8;
    int x0 = g * 8;
    if (x0 + 8 <= nx && (((uintptr_t)dst) & 7) == 0) {
        *(u64 *)dst = x;
    } else {
        for (int k = 0; k < 8 && x0 + k < nx; k++) dst[k] = (uint8_t)((x >> (8 * k)) & 1);
    }
}

TOMO_API int tomo_unpack_bits(const uint64_t *bits, uint8_t *mask, int nz, int ny, int nx, void *stream)
{
    if (!mask || !bits || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    int64_t rows = (int64_t)nz * ny;
    int64_t n = rows * ((nx + 7) / 8);
    int64_t blocks = ceil_div64(n, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(unpack_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u64 *)bits, mask,
                       rows, nx, wx);
    return tomo_status();
}

__global__ __launch_bounds__(256) void popcount_kernel(const u64 *__restrict__ bits, int64_t nwords, u64 *count)
{
    u64 acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x)
        acc += (u64)__popcll(bits[i]);
    acc = wave_sum64(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(count, acc);
}

TOMO_API int tomo_popcount(const uint64_t *bits, int nz, int ny, int nx, unsigned long long *count, void *stream)
{
    if (!bits || !count || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    int64_t nwords = (int64_t)nz * ny * tomo_words_per_row(nx);
    int64_t blocks = ceil_div64(nwords, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(popcount_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u64 *)bits,
                       nwords, count);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------
// 2-D fill holes (ndimage.binary_fill_holes): flood the background from outside the image
// through 4-connected background pixels; pixels not reached are holes -> set.
// One workgroup (1024 threads) iterates until stable.  `reach` lives in scratch (L2 resident).
// Each thread owns a (word column, chunk of rows) strip and sweeps it down then up, so one
// iteration moves the flood by a whole strip vertically and a whole word horizontally.

// complete flood of seed s through free mask m inside one 64-bit word (both directions)
__device__ static inline u64 word_flood(u64 s, u64 m)
{
    u64 g = s & m, p = m;
    // towards higher bits
    u64 gl = g, pl = p;
    gl |= pl & (gl << 1);  pl &= pl << 1;
    gl |= pl & (gl << 2);  pl &= pl << 2;
    gl |= pl & (gl << 4);  pl &= pl << 4;
    gl |= pl & (gl << 8);  pl &= pl << 8;
    gl |= pl & (gl << 16); pl &= pl << 16;
    gl |= pl & (gl << 32);
    // towards lower bits
    u64 gr = g, pr = p;
    gr |= pr & (gr >> 1);  pr &= pr >> 1;
    gr |= pr & (gr >> 2);  pr &= pr >> 2;
    gr |= pr & (gr >> 4);  pr &= pr >> 4;
    gr |= pr & (gr >> 8);  pr &= pr >> 8;
    gr |= pr & (gr >> 16); pr &= pr >> 16;
    gr |= pr & (gr >> 32);
    return gl | gr;
}

#define FH_THREADS 1024
__device__ static inline void fill_holes_one_workgroup(u64 *slice, u64 *reach, int ny, int nx, int wx)
{
    __shared__ int s_any, s_changed;
    const int tid = threadIdx.x;
    const int64_t nwords = (int64_t)ny * wx;
    const u64 tailmask = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
    if (tid == 0) { s_any = 0; s_changed = 0; }
    __syncthreads();
    // np.any(slice) guard of the reference (voxel_processor.py:60,66): an empty slice stays empty
    int any = 0;
    for (int64_t i = tid; i < nwords; i += FH_THREADS) any |= slice[i] != 0;
    if (any) s_any = 1;
    __syncthreads();
    if (!s_any) return;
    // seed: background pixels on the image border
    for (int64_t i = tid; i < nwords; i += FH_THREADS) {
        int y = (int)(i / wx), w = (int)(i - (int64_t)y * wx);
        u64 valid = (w == wx - 1) ? tailmask : ~0ull;
        u64 freem = ~slice[i] & valid;
        u64 seed = 0;
        if (y == 0 || y == ny - 1) seed = freem;
        if (w == 0) seed |= freem & 1ull;
        if (w == (nx - 1) / 64) seed |= freem & (1ull << ((nx - 1) & 63));
        reach[i] = word_flood(seed, freem);
    }
    __syncthreads();
    // strips: thread -> (word column w, row chunk c)
    const int chunks = FH_THREADS / wx > 0 ? FH_THREADS / wx : 1;   // row chunks per word column
    const int rows_per_chunk = (ny + chunks - 1) / chunks;
    const int64_t max_iter = (int64_t)ny * wx * 64 + 2;
    for (int64_t it = 0; it < max_iter; it++) {
        int changed = 0;
        for (int strip = tid; strip < wx * chunks; strip += FH_THREADS) {
            int w = strip % wx, c = strip / wx;
            int ya = c * rows_per_chunk, yb = ya + rows_per_chunk < ny ? ya + rows_per_chunk : ny;
            u64 valid = (w == wx - 1) ? tailmask : ~0ull;
            for (int dir = 0; dir < 2; dir++) {
                for (int k = 0; k < yb - ya; k++) {
                    int y = dir == 0 ? ya + k : yb - 1 - k;
                    int64_t i = (int64_t)y * wx + w;
                    u64 freem = ~slice[i] & valid;
                    u64 r = reach[i];
                    u64 nb = 0;
                    if (y > 0) nb |= reach[i - wx];
                    if (y < ny - 1) nb |= reach[i + wx];
                    if (w > 0) nb |= reach[i - 1] >> 63;
                    if (w < wx - 1) nb |= reach[i + 1] << 63;
                    u64 nr = word_flood(r | (nb & freem), freem);
                    if (nr != r) { reach[i] = nr; changed = 1; }
                }
            }
        }
        if (changed) s_changed = 1;
        __threadfence_block();
        __syncthreads();
        int ch = s_changed;
        __syncthreads();
        if (tid == 0) s_changed = 0;
        __syncthreads();
        if (!ch) break;
    }
    // holes = background not reached -> filled = everything not reached
    for (int64_t i = tid; i < nwords; i += FH_THREADS) {
        int w = (int)(i % wx);
        u64 valid = (w == wx - 1) ? tailmask : ~0ull;
        slice[i] = ~reach[i] & valid;
    }
}

__global__ __launch_bounds__(FH_THREADS) void fill_holes_kernel(u64 *slice, u64 *reach, int ny, int nx, int wx)
{
    fill_holes_one_workgroup(slice, reach, ny, nx, wx);
}

// Safety net behind fill_holes_bands_kernel (below): its grid barrier gives up when not all of its workgroups become
// resident in time (a GPU shared with other streams or processes, a partitioned device) and the slice is then left
// unfilled or partly filled, with ctrl[8 sl + 2] set.  This one-workgroup kernel runs right after it and redoes such a
// slice with the flood above -- filling holes is idempotent and monotone, so a partly filled slice gives the same result
// as the original.  The flags are read before `reach` (which overlaps the control words in scratch) is written.
__global__ __launch_bounds__(FH_THREADS) void fill_holes_redo_kernel(u64 *sliceA, u64 *sliceB, u64 *ctrl, u64 *reach, int ny,
                                                                     int nx, int wx)
{
    __shared__ int s_redo[2];
    if (threadIdx.x < 2) s_redo[threadIdx.x] = ctrl[8 * threadIdx.x + 2] != 0ull;
    __syncthreads();
    const int ra = s_redo[0], rb = s_redo[1] && sliceB != nullptr;
    if (!ra && !rb) return;
    __syncthreads();
    if (ra) fill_holes_one_workgroup(sliceA, reach, ny, nx, wx);
    __syncthreads();
    if (rb) fill_holes_one_workgroup(sliceB, reach, ny, nx, wx);
}

// ------------------------------------------------------------------------------------------
// The same flood on MANY workgroups (the kernel above uses one CU of 256 and took 184 us for a filled 1024^2 ellipse,
// 823 us at 2048^2 -- profiles/r02_fill_holes.md).  The slice is cut into bands of rows; a band (free mask F and reach
// set R) lives in its workgroup's LDS.  Inside a band the flood runs to convergence with COMPLETE sweeps: along rows
// (in-word flood + a carry chain across the words, both directions) and along bit columns (a carry chain down and up the
// rows) -- a sweep moves the flood arbitrarily far along its axis, so convex shapes take one round.  Between bands only
// three words per word column travel: what reaches the band's bottom row (Gd) and top row (Gu), and which bit columns
// are free all the way through (P).  After a grid barrier every band folds the bands above / below it into a carry-in
// for its top / bottom row (c = G | (P & c), the close-ends carry chain again) and, if that adds anything, converges
// again.  Rounds repeat until no band received a new bit; two grid barriers for a slice whose flood needs no detour.
// Both end slices of a volume run in one launch (blockIdx.y).
// The grid barrier needs all workgroups of a slice resident at once: at most 256 per slice, of 256 threads (the launcher falls
// back to the one-workgroup kernel otherwise) on a 256-CU part; its spin is bounded, so every wave terminates.
#define FB_THREADS 256
#define FB_MAX_BANDS 256
#define FB_LDS_WORDS 4096            // RB * wx <= this: F and R of a band in 64 KiB
#define FB_FOLD 8                    // partial folds per word column and direction (carry fold across the bands)

struct FillBands {
    u64 *slice[2];
    u64 *ctrl;        // [2][8]: bar, any, err, -, flag[4]
    u64 *comp;        // [2][3][nb][wx]: Gd, Gu, P
    int ny, nx, wx, RB, nb;
    int spin_limit;   // sleeps a workgroup waits at a grid barrier before it gives up (tests: TOMO_FILL_SPIN_LIMIT=0)
};

__device__ static inline bool fb_barrier(u64 *bar, u64 target, int spin_limit)
{   // all workgroups of this slice; false if the wait was abandoned (never expected: every workgroup is resident)
    __syncthreads();
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd((unsigned long long *)bar, 1ull);
        int spins = 0;
        ok = 1;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > spin_limit) { ok = 0; break; }
        }
        __threadfence();
    }
    __syncthreads();
    return ok != 0;
}

// R <- complete row floods (returns through *chg whether any word changed); all threads of the workgroup
__device__ static inline void fb_sweep_rows(u64 *__restrict__ R, const u64 *__restrict__ F, int rows, int wx, int *chg)
{
    const int tid = threadIdx.x;
    int c = 0;
    for (int i = tid; i < rows * wx; i += FB_THREADS) {
        const u64 r = R[i], nr = word_flood(r, F[i]);
        if (nr != r) { R[i] = nr; c = 1; }
    }
    __syncthreads();
    for (int r = tid; r < rows; r += FB_THREADS) {       // carries across the words of a row: one thread per row
        u64 *Rr = R + r * wx;
        const u64 *Fr = F + r * wx;
        u64 carry = 0;
        for (int w = 0; w < wx; w++) {                    // towards higher x
            u64 v = Rr[w];
            if (carry & Fr[w] & ~v) { v = word_flood(v | 1ull, Fr[w]); Rr[w] = v; c = 1; }
            carry = v >> 63;
        }
        carry = 0;
        for (int w = wx - 1; w >= 0; w--) {               // towards lower x
            u64 v = Rr[w];
            if ((carry << 63) & Fr[w] & ~v) { v = word_flood(v | (1ull << 63), Fr[w]); Rr[w] = v; c = 1; }
            carry = v & 1ull;
        }
    }
    if (c) *chg = 1;
    __syncthreads();
}

// R <- complete floods along the bit columns (down, then up); one thread per word column
__device__ static inline void fb_sweep_columns(u64 *__restrict__ R, const u64 *__restrict__ F, int rows, int wx, int *chg)
{
    int c = 0;
    for (int w = threadIdx.x; w < wx; w += FB_THREADS) {
        u64 carry = R[w];
        for (int r = 1; r < rows; r++) {
            const u64 v = R[r * wx + w], nv = v | (carry & F[r * wx + w]);
            if (nv != v) { R[r * wx + w] = nv; c = 1; }
            carry = nv;
        }
        for (int r = rows - 2; r >= 0; r--) {
            const u64 v = R[r * wx + w], nv = v | (carry & F[r * wx + w]);
            if (nv != v) { R[r * wx + w] = nv; c = 1; }
            carry = nv;
        }
    }
    if (c) *chg = 1;
    __syncthreads();
}

__device__ static inline void fb_converge(u64 *R, const u64 *F, int rows, int wx, int *s_chg)
{
    for (int it = 0;; it++) {
        if (threadIdx.x == 0) *s_chg = 0;
        __syncthreads();
        fb_sweep_rows(R, F, rows, wx, s_chg);
        const int ch = *s_chg;
        __syncthreads();
        if (it > 0 && !ch) break;             // the column sweep before was complete and this row sweep added nothing
        if (threadIdx.x == 0) *s_chg = 0;
        __syncthreads();
        fb_sweep_columns(R, F, rows, wx, s_chg);
        const int cv = *s_chg;
        __syncthreads();
        if (!cv) break;                       // the row sweep before was complete and this column sweep added nothing
    }
}

__global__ __launch_bounds__(FB_THREADS) void fill_holes_bands_kernel(const FillBands p)
{
    extern __shared__ u64 fb_lds[];
    __shared__ int s_chg, s_any;
    const int tid = threadIdx.x, b = blockIdx.x, sl = blockIdx.y;
    const int wx = p.wx, nb = p.nb;
    u64 *F = fb_lds, *R = fb_lds + p.RB * wx;
    u64 *fold_g = R + p.RB * wx, *fold_p = fold_g + 2 * FB_FOLD * wx;
    u64 *slice = p.slice[sl];
    u64 *ctrl = p.ctrl + 8 * sl;
    u64 *Gd = p.comp + (size_t)sl * 3 * nb * wx, *Gu = Gd + (size_t)nb * wx, *P = Gu + (size_t)nb * wx;
    const int y0 = b * p.RB, rows = (p.ny - y0 < p.RB) ? p.ny - y0 : p.RB;
    const u64 tailmask = (p.nx & 63) ? ((1ull << (p.nx & 63)) - 1ull) : ~0ull;
    // np.any(slice) guard of the reference (voxel_processor.py:60,66): an empty slice stays as it is.  Every workgroup finds
    // that out for itself -- the slice is 128 KiB - 512 KiB in L2, sixteen bytes per lane and round -- so an empty end
    // slice (the synthetic ellipsoid's) costs one short kernel and no grid barrier at all.
    if (tid == 0) s_any = 0;
    __syncthreads();
    {
        const int64_t nw = (int64_t)p.ny * wx;
        int any = 0;
        if ((nw & 1) == 0 && (((uintptr_t)slice) & 15) == 0) {
            const ulonglong2 *q = (const ulonglong2 *)slice;
            for (int64_t i = tid; i < nw / 2; i += FB_THREADS) { const ulonglong2 v = q[i]; any |= (v.x | v.y) != 0; }
        } else {
            for (int64_t i = tid; i < nw; i += FB_THREADS) any |= slice[i] != 0;
        }
        if (any) s_any = 1;
    }
    __syncthreads();
    if (!s_any) return;                                             // the same answer in every workgroup of this slice
    for (int i = tid; i < rows * wx; i += FB_THREADS) {
        const int r = i / wx, w = i - r * wx, y = y0 + r;
        const u64 v = slice[(int64_t)y * wx + w];
        const u64 valid = (w == wx - 1) ? tailmask : ~0ull;
        const u64 freem = ~v & valid;
        u64 seed = 0;                                           // background pixels on the image border
        if (y == 0 || y == p.ny - 1) seed = freem;
        if (w == 0) seed |= freem & 1ull;
        if (w == (p.nx - 1) / 64) seed |= freem & (1ull << ((p.nx - 1) & 63));
        F[i] = freem;
        R[i] = seed;
    }
    __syncthreads();
    fb_converge(R, F, rows, wx, &s_chg);
    for (int w = tid; w < wx; w += FB_THREADS) {
        u64 pr = ~0ull;
        for (int r = 0; r < rows; r++) pr &= F[r * wx + w];
        P[(size_t)b * wx + w] = pr;
        Gd[(size_t)b * wx + w] = R[(rows - 1) * wx + w];
        Gu[(size_t)b * wx + w] = R[w];
    }
    u64 k = 0;
    if (!fb_barrier(&ctrl[0], (u64)nb * ++k, p.spin_limit)) { if (tid == 0) ctrl[2] = 1; return; }
    const int64_t max_rounds = (int64_t)p.ny * wx * 64 + 2;
    for (int64_t g = 0; g < max_rounds; g++) {
        if (tid == 0) s_chg = 0;
        __syncthreads();
        int got = 0;
        // carry-in from the bands above (Gd, downwards) and below (Gu, upwards): c = G | (P & c) over up to nb bands.
        // Two levels, so that no thread waits on a long chain of dependent memory round trips: FB_FOLD threads per word
        // column each fold a run of bands into one (g, p) pair (their loads are independent and issued eight at a time),
        // then one thread per column folds the FB_FOLD pairs from LDS.
        const int run = (nb + FB_FOLD - 1) / FB_FOLD;
        for (int it = tid; it < wx * FB_FOLD * 2; it += FB_THREADS) {
            const int w = it % wx, j = (it / wx) % FB_FOLD, up = it / (wx * FB_FOLD);
            // down: bands [j run, (j+1) run) below b, ascending; up: the same run of bands above b, taken descending
            int a0 = j * run, a1 = a0 + run < nb ? a0 + run : nb;
            if (!up) { if (a1 > b) a1 = b; } else { if (a0 < b + 1) a0 = b + 1; }
            const u64 *Gsrc = up ? Gu : Gd;
            u64 g = 0, pp = ~0ull;
            for (int base = 0; base < a1 - a0; base += 8) {
                u64 gv[8], pv[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int n = base + q;
                    const int a = up ? a1 - 1 - n : a0 + n;
                    const bool ok = n < a1 - a0;
                    gv[q] = ok ? __hip_atomic_load(&Gsrc[(size_t)a * wx + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                    pv[q] = ok ? P[(size_t)a * wx + w] : ~0ull;
                }
#pragma unroll
                for (int q = 0; q < 8; q++) { g = gv[q] | (pv[q] & g); pp &= pv[q]; }
            }
            fold_g[(up * FB_FOLD + j) * wx + w] = g;
            fold_p[(up * FB_FOLD + j) * wx + w] = pp;
        }
        __syncthreads();
        for (int w = tid; w < wx; w += FB_THREADS) {
            u64 c = 0;
            for (int j = 0; j < FB_FOLD; j++) c = fold_g[j * wx + w] | (fold_p[j * wx + w] & c);
            u64 add = c & F[w] & ~R[w];
            if (add) { R[w] |= add; got = 1; }
            c = 0;
            for (int j = FB_FOLD - 1; j >= 0; j--) c = fold_g[(FB_FOLD + j) * wx + w] | (fold_p[(FB_FOLD + j) * wx + w] & c);
            add = c & F[(rows - 1) * wx + w] & ~R[(rows - 1) * wx + w];
            if (add) { R[(rows - 1) * wx + w] |= add; got = 1; }
        }
        if (got) s_chg = 1;
        __syncthreads();
        const int fresh = s_chg;
        __syncthreads();
        if (fresh) {
            fb_converge(R, F, rows, wx, &s_chg);
            for (int w = tid; w < wx; w += FB_THREADS) {
                __hip_atomic_store(&Gd[(size_t)b * wx + w], R[(rows - 1) * wx + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&Gu[(size_t)b * wx + w], R[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // round g's flag word (reused four rounds later, when every workgroup is long past reading it)
            if (tid == 0) __hip_atomic_store(&ctrl[4 + (g & 3)], (u64)(g + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!fb_barrier(&ctrl[0], (u64)nb * ++k, p.spin_limit)) { if (tid == 0) ctrl[2] = 1; return; }
        if (__hip_atomic_load(&ctrl[4 + (g & 3)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (u64)(g + 1)) break;
    }
    // holes = background not reached -> filled = everything not reached
    for (int i = tid; i < rows * wx; i += FB_THREADS) {
        const int r = i / wx, w = i - r * wx;
        const u64 valid = (w == wx - 1) ? tailmask : ~0ull;
        slice[(int64_t)(y0 + r) * wx + w] = ~R[i] & valid;
    }
}

// bands for a slice of ny rows x wx words: rows per band and their number (0: use the one-workgroup kernel)
static inline int fill_bands_plan(int ny, int wx, int *RB_out)
{
    if (wx * 8 > FB_LDS_WORDS) return 0;
    int RB = (ny + 63) / 64;                       // aim at 64 bands
    if (RB < 8) RB = 8;
    if (RB * wx > FB_LDS_WORDS) RB = FB_LDS_WORDS / wx;
    int nb = (ny + RB - 1) / RB;
    if (nb > FB_MAX_BANDS) return 0;
    *RB_out = RB;
    return nb;
}

// Can nb * nsl workgroups of the banded kernel be resident at once on the current device?  (Its grid barrier needs that; the
// answer is cached per LDS size.)  Occupancy per CU x CUs -- on a partitioned or smaller part the one-workgroup kernel runs.
static bool fill_bands_fit(int blocks, size_t lds_bytes)
{
    // one slot per device; a slot packs (LDS bytes << 20 | capacity + 1) into ONE atomic word, so concurrent callers (the rank
    // threads of a rehearsed slab job, one device each) either see a complete entry or recompute it -- no torn state
    static std::atomic<uint64_t> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const bool slot = dev >= 0 && dev < 64 && lds_bytes < ((size_t)1 << 40);
    if (slot) {
        const uint64_t e = cache[dev].load(std::memory_order_relaxed);
        if (e != 0 && (e >> 20) == (uint64_t)lds_bytes) return blocks <= (int)(e & 0xFFFFF) - 1;
    }
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fill_holes_bands_kernel, FB_THREADS, lds_bytes) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int capacity = per_cu * cus;
    if (capacity > 0xFFFFE) capacity = 0xFFFFE;
    if (slot) cache[dev].store(((uint64_t)lds_bytes << 20) | (uint64_t)(capacity + 1), std::memory_order_relaxed);
    return blocks <= capacity;
}

static int fill_holes_launch(u64 *sliceA, u64 *sliceB, int ny, int nx, int wx, u64 *scratch, hipStream_t st,
                             bool ctrl_cleared = false)
{
    int RB = 0;
    int nb = fill_bands_plan(ny, wx, &RB);
    const int nsl = sliceB ? 2 : 1;
    const size_t lds = ((size_t)2 * RB * wx + (size_t)4 * FB_FOLD * wx) * sizeof(u64);
    // scratch holds ny * wx + 8 words (the one-workgroup kernel's reach set): room for 16 control + 2 * 3 * nb * wx words?
    if (nb == 0 || 16 + (int64_t)nsl * 3 * nb * wx > (int64_t)ny * wx + 8 || !fill_bands_fit(nb * nsl, lds)) {
        hipLaunchKernelGGL(fill_holes_kernel, dim3(1), dim3(FH_THREADS), 0, st, sliceA, scratch, ny, nx, wx);
        if (sliceB) hipLaunchKernelGGL(fill_holes_kernel, dim3(1), dim3(FH_THREADS), 0, st, sliceB, scratch, ny, nx, wx);
        return tomo_status();
    }
    if (!ctrl_cleared && hipMemsetAsync(scratch, 0, 16 * sizeof(u64), st) != hipSuccess) return TOMO_E_LAUNCH;
    static const int spin_limit = getenv("TOMO_FILL_SPIN_LIMIT") ? atoi(getenv("TOMO_FILL_SPIN_LIMIT")) : (1 << 22);
    FillBands p;
    p.slice[0] = sliceA; p.slice[1] = sliceB ? sliceB : sliceA;
    p.ctrl = scratch; p.comp = scratch + 16;
    p.ny = ny; p.nx = nx; p.wx = wx; p.RB = RB; p.nb = nb; p.spin_limit = spin_limit;
    hipLaunchKernelGGL(fill_holes_bands_kernel, dim3((unsigned)nb, (unsigned)nsl), dim3(FB_THREADS), lds, st, p);
    // a barrier that was abandoned (never seen with every workgroup resident) must not leave a slice half filled
    hipLaunchKernelGGL(fill_holes_redo_kernel, dim3(1), dim3(FH_THREADS), 0, st, sliceA, sliceB, scratch, scratch, ny, nx, wx);
    return tomo_status();
}

TOMO_API int tomo_fill_holes_slice(uint64_t *bits, int nz, int ny, int nx, int z, uint64_t *scratch, void *stream)
{
    if (!bits || !scratch || nz <= 0 || ny <= 0 || nx <= 0 || z < 0 || z >= nz) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    if (wx > FH_THREADS) return TOMO_E_SIZE;
    return fill_holes_launch((u64 *)bits + (int64_t)z * ny * wx, nullptr, ny, nx, wx, (u64 *)scratch, (hipStream_t)stream);
}

// Both end slices (0 and nz - 1) of a volume in ONE launch -- what _close_volume_ends does first (voxel_processor.py:60-68).
TOMO_API int tomo_fill_holes_ends(uint64_t *bits, int nz, int ny, int nx, uint64_t *scratch, void *stream)
{
    if (!bits || !scratch || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    if (wx > FH_THREADS) return TOMO_E_SIZE;
    u64 *first = (u64 *)bits, *last = nz > 1 ? (u64 *)bits + (int64_t)(nz - 1) * ny * wx : nullptr;
    return fill_holes_launch(first, last, ny, nx, wx, (u64 *)scratch, (hipStream_t)stream);
}

// Outputs za .. zb - 1 of the fused pack + stencil pass: long ranges through pack_close_ho_kernel (its first run also takes the
// (zb - za) % PC_U slices in front), short ones through pack_close_kernel in runs of zr slices.
static int pack_close_launch(const uint8_t *mask, u64 *bits, int nz, int ny, int nx, int wx, int groups, int za, int zb, int lo_fixed,
                             int hi_fixed, const u64 *below, const u64 *above, hipStream_t st)
{
    static const bool handover = !(getenv("TOMO_PACK_HANDOVER") && atoi(getenv("TOMO_PACK_HANDOVER")) == 0);
    const int G = (zb - za) / PC_U;
    const int nwg = (G + 4 * (PC_ZR / PC_U) - 1) / (4 * (PC_ZR / PC_U));         // ~PC_ZR slices per wave
    if (handover && zb - za >= 4 * PC_ZR && G >= 4 * nwg) {
        const int head = (zb - za) % PC_U;
        const int64_t blocks = (int64_t)ny * groups * nwg;
        if (blocks > 0x7fffffff) return TOMO_E_SIZE;
        hipLaunchKernelGGL(pack_close_ho_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mask, bits, nz, ny, nx, wx, groups, nwg,
                           za + head, zb, lo_fixed, hi_fixed, below, above, head);
        return tomo_status();
    }
    // a short range (the slices at an end of a slab) in runs of 8: four times the waves, a quarter of the serial march each
    const int zr = (zb - za) <= 2 * PC_ZR ? PC_ZR / 4 : PC_ZR;
    const int runs = (zb - za + zr - 1) / zr;
    const int64_t waves = (int64_t)ny * groups * runs, blocks = ceil_div64(waves, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(pack_close_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mask, bits, nz, ny, nx, wx, groups, runs, za, zb,
                       lo_fixed, hi_fixed, below, above, zr);
    return tomo_status();
}

// np.stack + _close_volume_ends (voxel_processor.py:46, :56-77) from the uint8 mask stack to the closed bit volume:
// pack the two end slices, fill their holes, then the fused pack + stencil pass.  scratch: ny * wx + 8 words (the fill's).
// Requires what pack16_kernel requires (nx % 16 == 0, 16-byte aligned mask) and nz >= 3; returns TOMO_E_ARG otherwise
// (the caller then uses tomo_pack_bits + tomo_fill_holes_ends + tomo_close_ends_scan).
TOMO_API int tomo_pack_close_ends(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, uint64_t *scratch, void *stream)
{
    if (!mask || !bits || !scratch || nz < 3 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    if (nx % 16 != 0 || (((uintptr_t)mask) & 15) != 0) return TOMO_E_ARG;
    const int wx = (int)tomo_words_per_row(nx);
    if (wx > FH_THREADS) return TOMO_E_SIZE;
    const int groups = (wx + 15) / 16;
    hipLaunchKernelGGL(pack16_ends_kernel, dim3((unsigned)ceil_div64(ceil_div64((int64_t)ny * groups, PACK_U), 4), 2), dim3(256), 0,
                       (hipStream_t)stream, mask, (u64 *)bits, (int64_t)ny, nx, wx, groups, (int64_t)(nz - 1) * ny * nx,
                       (int64_t)(nz - 1) * ny * wx, (u64 *)scratch);
    int rc = fill_holes_launch((u64 *)bits, (u64 *)bits + (int64_t)(nz - 1) * ny * wx, ny, nx, wx, (u64 *)scratch, (hipStream_t)stream,
                               true);
    if (rc) return rc;
    return pack_close_launch(mask, (u64 *)bits, nz, ny, nx, wx, groups, 1, nz - 1, 1, 1, nullptr, nullptr, (hipStream_t)stream);
}

// tomo_pack_close_slab for the output slices [z_from, z_to) only: the slices in the middle of a slab depend on nothing but
// the mask, so a Z-slab rank enqueues them before it talks to its neighbours and the end ranges afterwards.  `below` /
// `above` are needed only if the range reaches slice 0 / nz - 1 of a slab whose end is not fixed.  The run decomposition
// starts at z_from: any split of [0, nz) into ranges gives the same bits.
TOMO_API int tomo_pack_close_range(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, int z_from, int z_to,
                                   const uint64_t *below, const uint64_t *above, int lo_fixed, int hi_fixed, void *stream)
{
    if (!mask || !bits || nz < 2 || ny <= 0 || nx <= 0 || z_from < 0 || z_to > nz) return TOMO_E_ARG;
    if (nx % 16 != 0 || (((uintptr_t)mask) & 15) != 0) return TOMO_E_ARG;
    const int za = z_from > (lo_fixed ? 1 : 0) ? z_from : (lo_fixed ? 1 : 0);
    const int zb = z_to < (hi_fixed ? nz - 1 : nz) ? z_to : (hi_fixed ? nz - 1 : nz);
    if (zb <= za) return TOMO_OK;
    if ((za == 0 && !below) || (zb == nz && !above)) return TOMO_E_ARG;
    const int wx = (int)tomo_words_per_row(nx);
    const int groups = (wx + 15) / 16;
    return pack_close_launch(mask, (u64 *)bits, nz, ny, nx, wx, groups, za, zb, lo_fixed, hi_fixed, (const u64 *)below, (const u64 *)above,
                             (hipStream_t)stream);
}

// The stencil alone, on n bit-packed slices with their two neighbours given separately: out[i] = mid[i] | (prev & next) with
// prev = i ? mid[i-1] : before, next = i < n-1 ? mid[i+1] : after.  A Z-slab rank closes the ORIGINAL halo slices it got
// from a neighbour itself (slab.py): one exchange of originals instead of one for the stencil's neighbours and a second one
// for the closed halo.  `out` must not overlap `mid`.
__global__ __launch_bounds__(256) void close_stencil_kernel(const u64 *__restrict__ before, const u64 *__restrict__ mid,
                                                            const u64 *__restrict__ after, int n, int64_t slice_words,
                                                            u64 *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * slice_words) return;
    const int z = (int)(i / slice_words);
    const int64_t w = i - (int64_t)z * slice_words;
    const u64 prev = z ? mid[i - slice_words] : before[w];
    const u64 next = z < n - 1 ? mid[i + slice_words] : after[w];
    out[i] = mid[i] | (prev & next);
}

TOMO_API int tomo_close_stencil(const uint64_t *before, const uint64_t *mid, const uint64_t *after, int n, int ny, int nx,
                                uint64_t *out, void *stream)
{
    if (!before || !mid || !after || !out || n < 1 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    const int64_t slice_words = (int64_t)ny * tomo_words_per_row(nx);
    hipLaunchKernelGGL(close_stencil_kernel, dim3((unsigned)ceil_div64((int64_t)n * slice_words, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const u64 *)before, (const u64 *)mid, (const u64 *)after, n, slice_words, (u64 *)out);
    return tomo_status();
}

// Everything a Z-slab rank still has to do to its bit volume once the neighbours' ORIGINAL edge slices have arrived, in ONE
// launch (was: two tomo_pack_close_range + two tomo_close_stencil, 11 + 14 + 5 + 5 us of four nearly empty launches):
//   blocks [0, bA)            pack + close of the slab's first `edge` slices   (tomo_pack_close_range(0, edge))
//   blocks [bA, bA + bB)      pack + close of its last `edge` slices           (tomo_pack_close_range(nz - edge, nz))
//   blocks [bA + bB, ...)     the stencil on the lower neighbour's halo slices, then on the upper neighbour's
struct SlabStencil { const u64 *before, *mid, *after; u64 *out; int n; };

__global__ __launch_bounds__(256) void slab_edges_kernel(const uint8_t *__restrict__ mask, u64 *__restrict__ bits, int nz, int ny, int nx,
                                                         int wx, int groups, int lo_fixed, int hi_fixed, const u64 *__restrict__ below,
                                                         const u64 *__restrict__ above, int zaA, int zbA, int runsA, int zaB, int zbB,
                                                         int runsB, int zr, unsigned bA, unsigned bB, unsigned bLo, const SlabStencil lo,
                                                         const SlabStencil hi, int64_t slice_words)
{
    unsigned b = blockIdx.x;
    if (b < bA) {
        pack_close_body((int64_t)b * 4 + (threadIdx.x >> 6), mask, bits, nz, ny, nx, wx, groups, runsA, zaA, zbA, lo_fixed, hi_fixed,
                        below, above, zr);
        return;
    }
    b -= bA;
    if (b < bB) {
        pack_close_body((int64_t)b * 4 + (threadIdx.x >> 6), mask, bits, nz, ny, nx, wx, groups, runsB, zaB, zbB, lo_fixed, hi_fixed,
                        below, above, zr);
        return;
    }
    b -= bB;
    const SlabStencil &s = b < bLo ? lo : hi;
    if (b >= bLo) b -= bLo;
    const int64_t i = (int64_t)b * blockDim.x + threadIdx.x;
    if (i >= (int64_t)s.n * slice_words) return;
    const int z = (int)(i / slice_words);
    const int64_t w = i - (int64_t)z * slice_words;
    const u64 prev = z ? s.mid[i - slice_words] : s.before[w];
    const u64 next = z < s.n - 1 ? s.mid[i + slice_words] : s.after[w];
    s.out[i] = s.mid[i] | (prev & next);
}

TOMO_API int tomo_slab_edges(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, int edge, const uint64_t *below,
                             const uint64_t *above, int lo_fixed, int hi_fixed, const uint64_t *lo_before, const uint64_t *lo_mid,
                             const uint64_t *lo_after, int lo_n, uint64_t *lo_out, const uint64_t *hi_before, const uint64_t *hi_mid,
                             const uint64_t *hi_after, int hi_n, uint64_t *hi_out, void *stream)
{
    if (!mask || !bits || nz < 2 || ny <= 0 || nx <= 0 || edge < 1 || 2 * edge > nz || lo_n < 0 || hi_n < 0) return TOMO_E_ARG;
    if (nx % 16 != 0 || (((uintptr_t)mask) & 15) != 0) return TOMO_E_ARG;
    if ((!lo_fixed && !below) || (!hi_fixed && !above)) return TOMO_E_ARG;
    if ((lo_n > 0 && (!lo_before || !lo_mid || !lo_after || !lo_out)) || (hi_n > 0 && (!hi_before || !hi_mid || !hi_after || !hi_out)))
        return TOMO_E_ARG;
    const int wx = (int)tomo_words_per_row(nx), groups = (wx + 15) / 16;
    const int64_t slice_words = (int64_t)ny * wx;
    const int zr = edge <= 2 * PC_ZR ? PC_ZR / 4 : PC_ZR;              // short ranges in runs of 8 (as tomo_pack_close_range)
    const int zaA = lo_fixed ? 1 : 0, zbA = edge, zaB = nz - edge, zbB = hi_fixed ? nz - 1 : nz;
    const int runsA = zbA > zaA ? (zbA - zaA + zr - 1) / zr : 0, runsB = zbB > zaB ? (zbB - zaB + zr - 1) / zr : 0;
    const int64_t bA = ceil_div64((int64_t)ny * groups * runsA, 4), bB = ceil_div64((int64_t)ny * groups * runsB, 4);
    const int64_t bLo = ceil_div64((int64_t)lo_n * slice_words, 256), bHi = ceil_div64((int64_t)hi_n * slice_words, 256);
    const int64_t blocks = bA + bB + bLo + bHi;
    if (blocks <= 0) return TOMO_OK;
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    SlabStencil lo{(const u64 *)lo_before, (const u64 *)lo_mid, (const u64 *)lo_after, (u64 *)lo_out, lo_n};
    SlabStencil hi{(const u64 *)hi_before, (const u64 *)hi_mid, (const u64 *)hi_after, (u64 *)hi_out, hi_n};
    hipLaunchKernelGGL(slab_edges_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, (u64 *)bits, nz, ny, nx, wx,
                       groups, lo_fixed, hi_fixed, (const u64 *)below, (const u64 *)above, zaA, zbA, runsA, zaB, zbB, runsB, zr,
                       (unsigned)bA, (unsigned)bB, (unsigned)bLo, lo, hi, slice_words);
    return tomo_status();
}

// The fused pass for ONE Z-slab of a sharded stack (slab.py): mask = the slab's nz slices, bits = its bit volume.
// lo_fixed / hi_fixed: the slab's first / last slice is a GLOBAL end slice, already packed and filled in `bits` (kept);
// otherwise the neighbour slice `below` / `above` (bit-packed (ny, wx), ORIGINAL content, from the rank below / above) closes
// the stencil and every slice of the slab is computed.  Same layout requirements as tomo_pack_close_ends; nz >= 2.
TOMO_API int tomo_pack_close_slab(const uint8_t *mask, uint64_t *bits, int nz, int ny, int nx, const uint64_t *below,
                                  const uint64_t *above, int lo_fixed, int hi_fixed, void *stream)
{
    if (!mask || !bits || nz < 2 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    if ((!lo_fixed && !below) || (!hi_fixed && !above)) return TOMO_E_ARG;
    if (nx % 16 != 0 || (((uintptr_t)mask) & 15) != 0) return TOMO_E_ARG;
    const int wx = (int)tomo_words_per_row(nx);
    const int groups = (wx + 15) / 16;
    const int za = lo_fixed ? 1 : 0, zb = hi_fixed ? nz - 1 : nz;
    if (zb <= za) return TOMO_OK;
    const int runs = (zb - za + PC_ZR - 1) / PC_ZR;
    const int64_t waves = (int64_t)ny * groups * runs, blocks = ceil_div64(waves, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(pack_close_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, (u64 *)bits, nz, ny, nx, wx,
                       groups, runs, za, zb, lo_fixed, hi_fixed, (const u64 *)below, (const u64 *)above, PC_ZR);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------
// close-ends recurrence  c'[z] = c[z] | (c[z+1] & c'[z-1]),  z = 1 .. nz-2   (voxel_processor.py:72-75;
// the np.any guards there are pure short-cuts).  A carry chain per (y,x) column:
//   c'[z] = G | (P & carry_in)  with  G' = c[z] | (c[z+1] & G),  P' = c[z+1] & P.
// Three small kernels: per z-chunk composite (G,P); sequential combine over the chunks; apply.
// The same composite is what a Z-slab rank publishes to its neighbours in the multi-GPU path.
#define CE_CHUNK 64

__global__ __launch_bounds__(256) void close_reduce_kernel(const u64 *__restrict__ bits, u64 *__restrict__ GP,
                                                           int nz, int64_t slice_words, int nchunks)
{
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (col >= slice_words) return;
    int za = 1 + k * CE_CHUNK, zb = za + CE_CHUNK;
    if (zb > nz - 1) zb = nz - 1;
    u64 G = 0, P = ~0ull;
    u64 next = bits[(int64_t)za * slice_words + col];
    for (int z = za; z < zb; z++) {
        u64 cur = next;
        next = bits[(int64_t)(z + 1) * slice_words + col];
        G = cur | (next & G);
        P = next & P;
    }
    GP[((int64_t)k * 2) * slice_words + col] = G;
    GP[((int64_t)k * 2 + 1) * slice_words + col] = P;
}

// carry[k] = c'[za_k - 1]; carry[0] = c[0]
__global__ __launch_bounds__(256) void close_carry_kernel(const u64 *__restrict__ bits, u64 *__restrict__ GP,
                                                          int64_t slice_words, int nchunks)
{
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= slice_words) return;
    u64 carry = bits[col];
    for (int k = 0; k < nchunks; k++) {
        u64 G = GP[((int64_t)k * 2) * slice_words + col], P = GP[((int64_t)k * 2 + 1) * slice_words + col];
        GP[((int64_t)k * 2) * slice_words + col] = carry;   // reuse the G slot for the carry-in
        carry = G | (P & carry);
    }
}

// The apply kernel must see the ORIGINAL first slice of the NEXT chunk (chunks run concurrently and
// rewrite their own slices), so the slice at every chunk boundary is snapshotted first.
TOMO_API int64_t tomo_close_ends_workspace_words(int nz, int ny, int nx)
{
    int64_t nchunks = nz > 2 ? (nz - 2 + CE_CHUNK - 1) / CE_CHUNK : 0;
    return nchunks * 3 * (int64_t)ny * tomo_words_per_row(nx) + 8;
}

__global__ __launch_bounds__(256) void close_snapshot_kernel(const u64 *__restrict__ bits, u64 *__restrict__ snap,
                                                             int nz, int64_t slice_words, int nchunks)
{
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (col >= slice_words) return;
    int zb = 1 + (k + 1) * CE_CHUNK;
    if (zb > nz - 1) zb = nz - 1;
    snap[(int64_t)k * slice_words + col] = bits[(int64_t)zb * slice_words + col];
}

__global__ __launch_bounds__(256) void close_apply_snap_kernel(u64 *__restrict__ bits, const u64 *__restrict__ GP,
                                                               const u64 *__restrict__ snap, int nz,
                                                               int64_t slice_words, int nchunks)
{
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = blockIdx.y;
    if (col >= slice_words) return;
    int za = 1 + k * CE_CHUNK, zb = za + CE_CHUNK;
    if (zb > nz - 1) zb = nz - 1;
    u64 prev = GP[((int64_t)k * 2) * slice_words + col];
    u64 next = bits[(int64_t)za * slice_words + col];   // own slice, not yet rewritten by this thread
    for (int z = za; z < zb; z++) {
        u64 cur = next;
        next = (z + 1 == zb) ? snap[(int64_t)k * slice_words + col] : bits[(int64_t)(z + 1) * slice_words + col];
        u64 nw = cur | (prev & next);
        if (nw != cur) bits[(int64_t)z * slice_words + col] = nw;
        prev = nw;
    }
}

TOMO_API int tomo_close_ends_scan(uint64_t *bits, int nz, int ny, int nx, uint64_t *workspace, void *stream)
{
    if (!bits || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    if (nz <= 2) return TOMO_OK;
    if (!workspace) return TOMO_E_ARG;
    int64_t sw = (int64_t)ny * tomo_words_per_row(nx);
    int nchunks = (nz - 2 + CE_CHUNK - 1) / CE_CHUNK;
    u64 *GP = (u64 *)workspace;
    u64 *snap = GP + (int64_t)nchunks * 2 * sw;
    dim3 grid((unsigned)ceil_div64(sw, 256), (unsigned)nchunks);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(close_reduce_kernel, grid, dim3(256), 0, s, (const u64 *)bits, GP, nz, sw, nchunks);
    hipLaunchKernelGGL(close_snapshot_kernel, grid, dim3(256), 0, s, (const u64 *)bits, snap, nz, sw, nchunks);
    hipLaunchKernelGGL(close_carry_kernel, dim3(grid.x), dim3(256), 0, s, (const u64 *)bits, GP, sw, nchunks);
    hipLaunchKernelGGL(close_apply_snap_kernel, grid, dim3(256), 0, s, (u64 *)bits, (const u64 *)GP, (const u64 *)snap, nz,
                       sw, nchunks);
    return tomo_status();
}

// (G, P) of the whole chain over slices 1 .. nz-2: composition of the per-chunk pairs.  This is what a Z-slab
// rank publishes to the other ranks (multi-GPU close-ends): c'[nz-2] = G | (P & c'[0]).
__global__ __launch_bounds__(256) void close_compose_kernel(const u64 *__restrict__ GP, u64 *__restrict__ out,
                                                            int64_t slice_words, int nchunks)
{
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= slice_words) return;
    u64 G = 0, P = ~0ull;
    for (int k = 0; k < nchunks; k++) {
        u64 Gk = GP[((int64_t)k * 2) * slice_words + col], Pk = GP[((int64_t)k * 2 + 1) * slice_words + col];
        G = Gk | (Pk & G);
        P = Pk & P;
    }
    out[col] = G;
    out[slice_words + col] = P;
}

TOMO_API int tomo_close_ends_gp(const uint64_t *bits, int nz, int ny, int nx, uint64_t *workspace, uint64_t *gp_out,
                                void *stream)
{
    if (!bits || !workspace || !gp_out || nz < 3 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    int64_t sw = (int64_t)ny * tomo_words_per_row(nx);
    int nchunks = (nz - 2 + CE_CHUNK - 1) / CE_CHUNK;
    u64 *GP = (u64 *)workspace;
    dim3 grid((unsigned)ceil_div64(sw, 256), (unsigned)nchunks);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(close_reduce_kernel, grid, dim3(256), 0, s, (const u64 *)bits, GP, nz, sw, nchunks);
    hipLaunchKernelGGL(close_compose_kernel, dim3(grid.x), dim3(256), 0, s, (const u64 *)GP, (u64 *)gp_out, sw, nchunks);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------
// 6-neighbour erosion (outside = 1) / dilation (outside = 0), one thread per word.
template <int OP>
__global__ __launch_bounds__(256) void morph_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, int nz, int ny,
                                                    int nx, int wx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)nz * ny * wx;
    if (i >= total) return;
    int w = (int)(i % wx);
    int64_t r = i / wx;
    int y = (int)(r % ny);
    int z = (int)(r / ny);
    const int64_t sw = (int64_t)ny * wx;
    const u64 tailmask = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
    const u64 valid = (w == wx - 1) ? tailmask : ~0ull;
    const u64 B = OP == 0 ? ~0ull : 0ull;   // value of voxels outside the volume
    u64 c = in[i];
    if (OP == 0) c |= ~valid;               // tail bits beyond nx count as outside (=1) for erosion
    u64 zl = z > 0 ? in[i - sw] : B, zh = z < nz - 1 ? in[i + sw] : B;
    u64 yl = y > 0 ? in[i - wx] : B, yh = y < ny - 1 ? in[i + wx] : B;
    u64 pl = w > 0 ? in[i - 1] : B;         // word holding x-1 of bit 0
    u64 ph = w < wx - 1 ? in[i + 1] : B;    // word holding x+1 of bit 63
    u64 xl = (c << 1) | (pl >> 63);         // neighbour x-1 of every bit
    u64 xh = (c >> 1) | (ph << 63);         // neighbour x+1 of every bit
    u64 res = OP == 0 ? (c & zl & zh & yl & yh & xl & xh) : (c | zl | zh | yl | yh | xl | xh);
    out[i] = res & valid;
}

TOMO_API int tomo_morph_pass(const uint64_t *in, uint64_t *out, int nz, int ny, int nx, int op, void *stream)
{
    if (!in || !out || in == out || nz <= 0 || ny <= 0 || nx <= 0 || (op != 0 && op != 1)) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    int64_t total = (int64_t)nz * ny * wx;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    if (op == 0)
        hipLaunchKernelGGL(morph_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u64 *)in,
                           (u64 *)out, nz, ny, nx, wx);
    else
        hipLaunchKernelGGL(morph_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u64 *)in,
                           (u64 *)out, nz, ny, nx, wx);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------
// Fused smoothing: all H erosion/dilation passes of smooth_voxel_data (voxel_processor.py:79-97) in ONE kernel.
// A block owns a tile of rows x words (plus an H-row / 1-word halo that it recomputes) and marches along z as a
// software pipeline: stage j turns level j-1 into level j,
//     L_j[s] = inplane_j(L_{j-1}[s])  o  L_{j-1}[s-1]  o  L_{j-1}[s+1]          (o = AND for erosion, OR for dilation)
// where the in-plane part (y+-1 rows, x+-1 bits) needs the neighbours' words: one LDS exchange per stage and
// slice, two alternating tiles, one barrier per stage.  Per level a thread keeps three words in registers.
// HBM traffic: ~1.6 x 1 read + 1 write of the bit volume instead of H reads + H writes.
#define FM_TW 16      // words of a row per tile (a 1024-voxel row is one tile)

__device__ static inline u64 fm_inplane(int op, u64 c, u64 yl, u64 yh, u64 pl, u64 ph, u64 valid)
{
    if (op == 0) c |= ~valid;                // bits beyond nx count as outside (=1) for erosion
    u64 xl = (c << 1) | (pl >> 63), xh = (c >> 1) | (ph << 63);
    return op == 0 ? (c & yl & yh & xl & xh) : (c | yl | yh | xl | xh);
}

template <int H, bool ROW16>
__global__ __launch_bounds__(1024) void morph_fused_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, int nz, int ny,
                                                           int nx, int wx, u32 ops, int rows_own, int zchunk)
{
    extern __shared__ u64 fm_tile[];                     // 2 x (TYH x TWH) words
    const int TWH = blockDim.x, TYH = blockDim.y;        // tile incl. halo
    const int tw = threadIdx.x, ty = threadIdx.y;
    const int xh = (wx > FM_TW) ? 1 : 0;                 // x halo only when the row is wider than one tile
    const int w0 = (int)blockIdx.x * (TWH - 2 * xh);     // first owned word
    const int w = w0 - xh + tw;
    const int y0 = (int)blockIdx.y * rows_own;
    const int y = y0 - H + ty;
    const int za = (int)blockIdx.z * zchunk, zb = za + zchunk < nz ? za + zchunk : nz;
    const bool inv = (w >= 0 && w < wx && y >= 0 && y < ny);             // this thread's column exists
    const bool own = inv && tw >= xh && tw < TWH - xh && ty >= H && ty < TYH - H && y < y0 + rows_own;
    const u64 tailmask = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
    const u64 valid = (w == wx - 1) ? tailmask : ~0ull;
    const int64_t sw = (int64_t)ny * wx;
    const int64_t col = inv ? (int64_t)y * wx + w : 0;
    const int tidx = ty * TWH + tw;
    u64 P[H], L1[H], L2[H];
#pragma unroll
    for (int j = 0; j < H; j++) { P[j] = 0; L1[j] = 0; L2[j] = 0; }
    for (int t = za - H; t < zb + H; t++) {
        u64 X = (inv && t >= 0 && t < nz) ? in[(int64_t)t * sw + col] : 0ull;   // level 0, slice t
#pragma unroll
        for (int j = 0; j < H; j++) {
            // stage j+1 consumes X = L_j[t - j] and produces L_{j+1}[t - j - 1]
            const int op = (ops >> j) & 1;
            const u64 B = op == 0 ? ~0ull : 0ull;
            const int sX = t - j;
            const u64 Xe = (sX >= 0 && sX < nz) ? X : B;
            u64 o = op == 0 ? (P[j] & L2[j] & Xe) : (P[j] | L2[j] | Xe);
            o &= valid;
            // in-plane part of slice sX for the next iteration: exchange X with the neighbours
            u64 *tile = fm_tile + (j & 1) * (TYH * TWH);
            tile[tidx] = X;
            __syncthreads();
            u64 yl = (ty > 0) ? tile[tidx - TWH] : B;
            u64 yh = (ty < TYH - 1) ? tile[tidx + TWH] : B;
            u64 pl, ph;
            if (ROW16) {   // a tile row is exactly one 16-lane DPP row: x neighbours without LDS
                int lo = (int)(u32)X, hi = (int)(u32)(X >> 32);
                u32 pll = (u32)__builtin_amdgcn_update_dpp(0, lo, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                u32 plh = (u32)__builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, false);
                u32 phl = (u32)__builtin_amdgcn_update_dpp(0, lo, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                u32 phh = (u32)__builtin_amdgcn_update_dpp(0, hi, 0x101, 0xf, 0xf, false);
                pl = ((u64)plh << 32) | pll;
                ph = ((u64)phh << 32) | phl;
            } else {
                pl = (tw > 0) ? tile[tidx - 1] : B;
                ph = (tw < TWH - 1) ? tile[tidx + 1] : B;
            }
            if (y <= 0) yl = B;
            if (y >= ny - 1) yh = B;
            if (w <= 0) pl = B;
            if (w >= wx - 1) ph = B;
            L2[j] = L1[j];
            L1[j] = Xe;
            P[j] = fm_inplane(op, X, yl, yh, pl, ph, valid);
            X = o;
        }
        const int so = t - H;                               // slice of the final level that just completed
        if (own && so >= za && so < zb) out[(int64_t)so * sw + col] = X;
    }
}

// ------------------------------------------------------------------------------------------
// Barrier-free variant for H <= 4 passes: ONE WAVE owns a whole (x, y) tile cross-section, so no stage needs LDS or a
// barrier.  Lane = row (64 rows, 64 - 2H owned), registers = a strip of FW_W owned words of that row plus ONE "halo
// word" whose top 8 bits are the 8 voxels left of the strip and whose low 8 bits the 8 voxels right of it.  The five
// words form a ring (halo word between the last and the first owned word), which makes every word's x-neighbours the
// previous / next word of the ring -- uniform code, no edge cases; the unused middle bits of the halo word decay by one
// bit per pass, exactly like the outer rows of the tile.  y-neighbours are the adjacent lanes (DPP wave shifts),
// z-neighbours the previous values in registers (same software pipeline along z as above).
// The block version spends its time in barriers (8 per slice, 16 waves each); this one is pure VALU.
#define FW_W 4

__device__ static inline u64 dpp_prev_u64(u64 v)
{   // value of lane - 1 (lane 0: 0, by bound_ctrl -- no "old" operand to initialise)
    return ((u64)(u32)__builtin_amdgcn_mov_dpp((int)(u32)(v >> 32), 0x138 /* wave_shr:1 */, 0xf, 0xf, true) << 32) |
           (u64)(u32)__builtin_amdgcn_mov_dpp((int)(u32)v, 0x138, 0xf, 0xf, true);
}
__device__ static inline u64 dpp_next_u64(u64 v)
{   // value of lane + 1 (lane 63: 0)
    return ((u64)(u32)__builtin_amdgcn_mov_dpp((int)(u32)(v >> 32), 0x130 /* wave_shl:1 */, 0xf, 0xf, true) << 32) |
           (u64)(u32)__builtin_amdgcn_mov_dpp((int)(u32)v, 0x130, 0xf, 0xf, true);
}

// OPS >= 0: the pass mask is a compile-time constant (the masks smooth_voxel_data produces), which removes every
// per-pass select; OPS < 0: generic, mask in `ops`.
template <int H, int OPS>
__global__ __launch_bounds__(256) void morph_wave_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, int nz, int ny,
                                                         int nx, int wx, u32 ops, int zchunk, int nxt, int nyt, int64_t nwaves)
{
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= nwaves) return;
    const int xt = (int)(wid % nxt), yt = (int)((wid / nxt) % nyt), zc = (int)(wid / ((int64_t)nxt * nyt));
    const int w0 = xt * FW_W;
    const int rows_own = 64 - 2 * H;
    const int y0 = yt * rows_own, y = y0 - H + lane;
    const int za = zc * zchunk, zb = za + zchunk < nz ? za + zchunk : nz;
    const bool rowv = y >= 0 && y < ny;
    const bool own_row = rowv && lane >= H && lane < 64 - H;
    const u64 outside = rowv ? 0ull : ~0ull;              // rows outside the volume read as the pass's border value
    const u64 tailmask = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
    // the right half of the halo word are the low bits of word w0 + FW_W: if that is the row's last word its bits beyond
    // nx are outside the volume too
    const u64 hmask = (w0 + FW_W == wx - 1) ? (tailmask | ~0xffull) : ~0ull;
    const int64_t sw = (int64_t)ny * wx;
    const int64_t rowoff = rowv ? (int64_t)y * wx : 0;
    // ring positions 0..FW_W-1 = owned words w0+k, position FW_W = halo word
    u64 P[H][FW_W + 1], L1[H][FW_W + 1], L2[H][FW_W + 1];
#pragma unroll
    for (int j = 0; j < H; j++)
#pragma unroll
        for (int k = 0; k <= FW_W; k++) { P[j][k] = 0; L1[j][k] = 0; L2[j][k] = 0; }
    // level-0 words of slice t: FW_W owned words + the two neighbours the halo word is cut from (loaded one slice ahead)
    u64 N[FW_W + 2];
#define FW_LOAD(tt)                                                                              \
    {                                                                                            \
        const bool sv = rowv && (tt) >= 0 && (tt) < nz;                                          \
        const u64 *src = in + (int64_t)(tt) * sw + rowoff;                                       \
        _Pragma("unroll") for (int k = 0; k < FW_W; k++) N[k] = (sv && w0 + k < wx) ? src[w0 + k] : 0ull; \
        N[FW_W] = (sv && w0 - 1 >= 0) ? src[w0 - 1] : 0ull;                                      \
        N[FW_W + 1] = (sv && w0 + FW_W < wx) ? src[w0 + FW_W] : 0ull;                            \
    }
    FW_LOAD(za - H);
    for (int t = za - H; t < zb + H; t++) {
        u64 X[FW_W + 1];
#pragma unroll
        for (int k = 0; k < FW_W; k++) X[k] = N[k];
        X[FW_W] = (N[FW_W] & 0xff00000000000000ull) | (N[FW_W + 1] & 0xffull);
        if (t + 1 < zb + H) FW_LOAD(t + 1);               // in flight while this slice is computed
#pragma unroll
        for (int j = 0; j < H; j++) {
            const int op = OPS >= 0 ? ((OPS >> j) & 1) : (int)((ops >> j) & 1u);     // 0 erosion (border 1), 1 dilation (border 0)
            const int sX = t - j;
            const u64 zout = (sX >= 0 && sX < nz) ? 0ull : ~0ull;                   // wave-uniform
            u64 C[FW_W + 1];
#pragma unroll
            for (int k = 0; k <= FW_W; k++) {
                // level j of slice sX as the passes see it: the border value outside the volume (z, then y and the tail)
                const u64 Xe = op == 0 ? (X[k] | zout) : (X[k] & ~zout);
                const u64 o = op == 0 ? (P[j][k] & L2[j][k] & Xe) : (P[j][k] | L2[j][k] | Xe);
                L2[j][k] = L1[j][k];
                L1[j][k] = Xe;
                u64 c = op == 0 ? (X[k] | outside) : (X[k] & ~outside);
                if (op == 0) {
                    if (k < FW_W && w0 + k == wx - 1) c |= ~tailmask;
                    if (k == FW_W) c |= ~hmask;
                }
                C[k] = c;
                X[k] = o;                                 // level j + 1 of slice sX - 1 (masked below)
            }
#pragma unroll
            for (int k = 0; k <= FW_W; k++) {
                const int kp = k == 0 ? FW_W : k - 1, kn = k == FW_W ? 0 : k + 1;
                u32 lin = (u32)(C[kp] >> 32), rin = (u32)C[kn];                     // words whose bit 63 / bit 0 shift in
                if (k < FW_W) {                                                       // volume border in x
                    if (w0 + k <= 0) lin = op == 0 ? ~0u : 0u;
                    if (w0 + k >= wx - 1) rin = op == 0 ? ~0u : 0u;
                }
                const u32 clo = (u32)C[k], chi = (u32)(C[k] >> 32);
                const u64 xl = ((u64)__builtin_amdgcn_alignbit(chi, clo, 31) << 32) | __builtin_amdgcn_alignbit(clo, lin, 31);
                const u64 xh = ((u64)__builtin_amdgcn_alignbit(rin, chi, 1) << 32) | __builtin_amdgcn_alignbit(chi, clo, 1);
                const u64 yl = dpp_prev_u64(C[k]), yh = dpp_next_u64(C[k]);
                P[j][k] = op == 0 ? (C[k] & yl & yh & xl & xh) : (C[k] | yl | yh | xl | xh);
            }
            if (w0 + FW_W - 1 >= wx - 1) {                                            // wave-uniform: the strip holds the last word
#pragma unroll
                for (int k = 0; k < FW_W; k++) if (w0 + k == wx - 1) X[k] &= tailmask;
            }
            X[FW_W] &= hmask;
        }
        const int so = t - H;
        if (own_row && so >= za && so < zb) {
            u64 *dst = out + (int64_t)so * sw + rowoff;
#pragma unroll
            for (int k = 0; k < FW_W; k++) if (w0 + k < wx) dst[w0 + k] = X[k];
        }
    }
}
#undef FW_LOAD


// ------------------------------------------------------------------------------------------
// Same wave-per-cross-section scheme, rewritten on 32-bit halves for the compile-time pass masks: gfx950's v_bitop3_b32
// evaluates any 3-input boolean function in one instruction (the compiler only forms it from 32-bit expressions), which
// makes a 6-neighbour erosion as cheap as a dilation and folds the z-border value into the combining op for free.
// Per ring word and pass: out = Q op x, c = x masked, 2 alignbit + 2 DPP per half, two 3-input ops for the in-plane
// cross, one for the z partner -- 9 VALU ops per half instead of ~15, and no register rotation: level j's input
// alternates between two register sets by slice parity (loop unrolled by two), Q = inplane(s-1) op x(s-2) is carried.
// The halo word is simply {low dword of the right neighbour word, high dword of the left one}: all 32 bits are real data.
__device__ static inline u32 fw_out(int OP, u32 q, u32 x, u32 zo)
{   // erosion: q & (x | zo)   dilation: q | (x & ~zo)      (zo: all ones where the slice lies outside the volume)
    return OP == 0 ? __builtin_amdgcn_bitop3_b32(q, x, zo, 0xE0) : __builtin_amdgcn_bitop3_b32(q, x, zo, 0xF4);
}
__device__ static inline u32 fw_op3(int OP, u32 a, u32 b, u32 c)
{
    return OP == 0 ? __builtin_amdgcn_bitop3_b32(a, b, c, 0x80) : __builtin_amdgcn_bitop3_b32(a, b, c, 0xFE);
}

// Memory side (STAGED): with lane = row every direct load or store instruction touches 64 different 128-byte lines, and
// the texture addresser -- not bytes, not VALU -- then sets the kernel's time (TA_BUSY 90 %, waves in issue stall half
// their life).  So global memory is accessed row-major instead -- 4 lanes x 16 B per row and instruction: a row's window
// [w0 - 2, w0 + 6) words is one 64-byte piece of a line -- and a wave-private LDS image turns that into lane = row and
// back (80- and 48-byte row pitches: conflict-free 16-byte LDS accesses).  Needs even wx and 16-byte aligned volumes;
// everything else takes the direct path (STAGED = false).
#ifndef FW_DPP_FUSE
#define FW_DPP_FUSE 1
#endif
template <int H, int OPS, bool STAGED>
__global__ __launch_bounds__(256, 2) void morph_wave32_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, int nz, int ny,
                                                              int nx, int wx, int zchunk, int nxt, int nyt, int64_t nwaves)
{
    constexpr int W = FW_W;
    constexpr int LB_PITCH = 80, SB_PITCH = 48, WAVE_LDS = 64 * LB_PITCH + 64 * SB_PITCH;
    __shared__ __attribute__((aligned(16))) unsigned char s_stage[STAGED ? 4 * WAVE_LDS : 16];
    const int lane = threadIdx.x & 63;
    // the wave index through readfirstlane: everything derived from it (slice range, loop counter, border flags) is
    // then known to be wave-uniform and lives in SGPRs / scalar branches
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (wid >= nwaves) return;
    const int xt = (int)(wid % nxt), yt = (int)((wid / nxt) % nyt), zc = (int)(wid / ((int64_t)nxt * nyt));
    const int w0 = xt * W;
    const int rows_own = 64 - 2 * H;
    const int y0 = yt * rows_own, y = y0 - H + lane;
    const int za = zc * zchunk, zb = za + zchunk < nz ? za + zchunk : nz;
    const int zend = zb + H + 1;            // + 1: a slice is stored one iteration after it was finished (see below)
    const bool rowv = y >= 0 && y < ny;
    const bool own_row = rowv && lane >= H && lane < 64 - H;
    const u64 tailmask = (nx & 63) ? ((1ull << (nx & 63)) - 1ull) : ~0ull;
    const int64_t sw = (int64_t)ny * wx;
    const int64_t rowoff = rowv ? (int64_t)y * wx : 0;
    // Bits that lie outside the volume read as the pass's border value: orow (per lane) for rows beyond ny, mu (wave-
    // uniform, SGPRs) for words beyond wx and the tail of the last word.
    const u32 orow = rowv ? 0u : ~0u;
    u32 mu[W + 1][2];
#pragma unroll
    for (int k = 0; k < W; k++) {
        const u64 mk = (w0 + k >= wx) ? ~0ull : (w0 + k == wx - 1 ? ~tailmask : 0ull);
        mu[k][0] = (u32)mk;
        mu[k][1] = (u32)(mk >> 32);
    }
    mu[W][0] = (w0 + W >= wx) ? ~0u : (w0 + W == wx - 1 ? (u32)~tailmask : 0u);
    mu[W][1] = (w0 - 1 < 0) ? ~0u : 0u;

    u32 X[2][H + 1][W + 1][2], Q[H][W + 1][2];
#pragma unroll
    for (int k = 0; k <= W; k++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int j = 0; j <= H; j++) { X[0][j][k][h] = 0; X[1][j][k][h] = 0; }
#pragma unroll
            for (int j = 0; j < H; j++) Q[j][k][h] = 0;
        }

    // ---- direct path: addresses clamped into the volume; what is read from a clamped place is masked or replaced by zo
    int wk[W];
#pragma unroll
    for (int k = 0; k < W; k++) wk[k] = w0 + k < wx ? w0 + k : wx - 1;
    const int wl = w0 - 1 >= 0 ? w0 - 1 : 0, wr = w0 + W < wx ? w0 + W : wx - 1;
    // ---- staged path: load instruction q covers rows 16 q .. 16 q + 15 (4 lanes a row), store instruction q rows
    //      32 q .. 32 q + 31 (2 lanes a row)
    unsigned char *lb = s_stage + (STAGED ? wv * WAVE_LDS : 0), *sb = lb + (STAGED ? 64 * LB_PITCH : 0);
#ifndef FW_AHEAD
#define FW_AHEAD 1      // slices of prefetch in flight.  2 (a second set of four 16-byte registers) measured SLOWER: 0.184 against
#endif                  // 0.177 ms for the eight passes at 1024^3, 1.61 against 1.45 ms at 2048^3 (round 3)
    uint4 G2[FW_AHEAD][4];
    const int lrow = lane >> 2, lpart = lane & 3, lwp = w0 - 2 + 2 * lpart;       // first word of this lane's 16 bytes
    const bool lpv = lwp >= 0 && lwp < wx;
    const int srow = lane >> 1, spart = lane & 1, swp = w0 + 2 * spart;
    const bool spv = swp < wx;

#define FW_ISSUE(G, tt)                                                                          \
    {                                                                                            \
        const int tc = (tt) < 0 ? 0 : ((tt) >= nz ? nz - 1 : (tt));                              \
        const u64 *sl = in + (int64_t)tc * sw;                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                                          \
            const int yy = y0 - H + 16 * q + lrow;                                               \
            if (lpv && yy >= 0 && yy < ny) G[q] = *(const uint4 *)(sl + (int64_t)yy * wx + lwp); \
        }                                                                                        \
    }
#define FW_LAND(G, dst)                                                                          \
    {                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; q++)                                            \
            *(uint4 *)(lb + (16 * q + lrow) * LB_PITCH + 16 * lpart) = G[q];                     \
        const uint4 a = *(const uint4 *)(lb + lane * LB_PITCH + 16);                             \
        const uint4 b = *(const uint4 *)(lb + lane * LB_PITCH + 32);                             \
        dst[0][0] = a.x; dst[0][1] = a.y; dst[1][0] = a.z; dst[1][1] = a.w;                      \
        dst[2][0] = b.x; dst[2][1] = b.y; dst[3][0] = b.z; dst[3][1] = b.w;                      \
        dst[W][1] = *(const u32 *)(lb + lane * LB_PITCH + 12);                                   \
        dst[W][0] = *(const u32 *)(lb + lane * LB_PITCH + 48);                                   \
    }
#define FW_LOAD32(dst, tt)                                                                       \
    {                                                                                            \
        const int tc = (tt) < 0 ? 0 : ((tt) >= nz ? nz - 1 : (tt));                              \
        const u32 *src = (const u32 *)(in + (int64_t)tc * sw + rowoff);                          \
        _Pragma("unroll") for (int k = 0; k < W; k++) {                                          \
            const uint2 v = *(const uint2 *)(src + 2 * wk[k]);                                   \
            dst[k][0] = v.x; dst[k][1] = v.y;                                                    \
        }                                                                                        \
        dst[W][0] = src[2 * wr];                                                                 \
        dst[W][1] = src[2 * wl + 1];                                                             \
    }
    if (STAGED) {
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int a = 0; a < FW_AHEAD; a++) G2[a][q] = make_uint4(0, 0, 0, 0);
        // the loop is unrolled by two slices (p = 0, 1): with two slices of prefetch slice t lands from set p and slice t + 2 is
        // issued into it right afterwards; with one, everything goes through set 0
        FW_ISSUE(G2[0], za - H);
        if (FW_AHEAD == 2) FW_ISSUE(G2[1], za - H + 1);
    } else {
        FW_LOAD32(X[0][0], za - H);
    }
    for (int t0 = za - H; t0 < zend; t0 += 2) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int t = t0 + p;
            if (t < zend) {
                if (STAGED) {
                    // top of the iteration: land the prefetched slice, store the slice finished in the previous iteration,
                    // issue the next loads -- all before the first level, so the loads have the whole iteration to arrive
                    FW_LAND(G2[FW_AHEAD == 2 ? p : 0], X[p][0]);
                    const int so = t - 1 - H;
                    const bool sv = so >= za && so < zb;
                    if (sv) {
                        *(uint4 *)(sb + lane * SB_PITCH) =
                            make_uint4(X[p ^ 1][H][0][0] & ~mu[0][0], X[p ^ 1][H][0][1] & ~mu[0][1],
                                       X[p ^ 1][H][1][0] & ~mu[1][0], X[p ^ 1][H][1][1] & ~mu[1][1]);
                        *(uint4 *)(sb + lane * SB_PITCH + 16) =
                            make_uint4(X[p ^ 1][H][2][0] & ~mu[2][0], X[p ^ 1][H][2][1] & ~mu[2][1],
                                       X[p ^ 1][H][3][0] & ~mu[3][0], X[p ^ 1][H][3][1] & ~mu[3][1]);
                        u64 *sl = out + (int64_t)so * sw;
#pragma unroll
                        for (int q = 0; q < 2; q++) {
                            const int r = 32 * q + srow, yy = y0 - H + r;
                            const uint4 v = *(const uint4 *)(sb + r * SB_PITCH + 16 * spart);
                            if (spv && r >= H && r < 64 - H && yy >= 0 && yy < ny) *(uint4 *)(sl + (int64_t)yy * wx + swp) = v;
                        }
                    }
                    FW_ISSUE(G2[FW_AHEAD == 2 ? p : 0], t + FW_AHEAD);
                }
#pragma unroll
                for (int j = 0; j < H; j++) {
                    const int OP = (OPS >> j) & 1;                  // a constant once the level loop is unrolled
                    const int sX = t - j;
                    const u32 zo = (sX >= 0 && sX < nz) ? 0u : ~0u;                  // wave-uniform
                    const u32 zp = (sX - 1 >= 0 && sX - 1 < nz) ? 0u : ~0u;
                    u32 C[W + 1][2];
#pragma unroll
                    for (int k = 0; k <= W; k++)
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const u32 x = X[p][j][k][h];
                            X[p][j + 1][k][h] = fw_out(OP, Q[j][k][h], x, zo);
                            // erosion: x | orow | mu     dilation: x & ~orow & ~mu
                            C[k][h] = OP == 0 ? __builtin_amdgcn_bitop3_b32(x, orow, mu[k][h], 0xFE)
                                              : __builtin_amdgcn_bitop3_b32(x, orow, mu[k][h], 0x10);
                        }
#pragma unroll
                    for (int k = 0; k <= W; k++) {
                        const int kp = k == 0 ? W : k - 1, kn = k == W ? 0 : k + 1;
                        const u32 clo = C[k][0], chi = C[k][1], lin = C[kp][1], rin = C[kn][0];
                        const u32 xl0 = __builtin_amdgcn_alignbit(clo, lin, 31), xl1 = __builtin_amdgcn_alignbit(chi, clo, 31);
                        const u32 xh0 = __builtin_amdgcn_alignbit(chi, clo, 1), xh1 = __builtin_amdgcn_alignbit(rin, chi, 1);
                        const u32 yl0 = (u32)__builtin_amdgcn_mov_dpp((int)clo, 0x138, 0xf, 0xf, true);
                        const u32 yl1 = (u32)__builtin_amdgcn_mov_dpp((int)chi, 0x138, 0xf, 0xf, true);
                        const u32 yh0 = (u32)__builtin_amdgcn_mov_dpp((int)clo, 0x130, 0xf, 0xf, true);
                        const u32 yh1 = (u32)__builtin_amdgcn_mov_dpp((int)chi, 0x130, 0xf, 0xf, true);
#if FW_DPP_FUSE
                        // c op row-above op row-below as TWO 2-input ops that each take their shifted operand through DPP
                        // (v_and_b32_dpp / v_or_b32_dpp: the compiler folds the v_mov_b32_dpp into a VOP2 consumer, never
                        // into the VOP3 v_bitop3_b32) -- 8 instead of 9 VALU ops per half and pass.  The empty asm keeps the
                        // two ops from being re-fused into one 3-input bitop3 with two separate DPP moves.
                        u32 a0 = OP == 0 ? (clo & yl0) : (clo | yl0), a1 = OP == 0 ? (chi & yl1) : (chi | yl1);
                        asm("" : "+v"(a0), "+v"(a1));
                        u32 b0 = OP == 0 ? (a0 & yh0) : (a0 | yh0), b1 = OP == 0 ? (a1 & yh1) : (a1 | yh1);
                        asm("" : "+v"(b0), "+v"(b1));
                        Q[j][k][0] = fw_out(OP, fw_op3(OP, b0, xl0, xh0), X[p ^ 1][j][k][0], zp);
                        Q[j][k][1] = fw_out(OP, fw_op3(OP, b1, xl1, xh1), X[p ^ 1][j][k][1], zp);
#else
                        Q[j][k][0] = fw_out(OP, fw_op3(OP, fw_op3(OP, clo, yl0, yh0), xl0, xh0), X[p ^ 1][j][k][0], zp);
                        Q[j][k][1] = fw_out(OP, fw_op3(OP, fw_op3(OP, chi, yl1, yh1), xl1, xh1), X[p ^ 1][j][k][1], zp);
#endif
                    }
                    if (j == 0) {
                        // The slice finished in the previous iteration is stored here, before the next loads are issued
                        // (the waitcnt in front of loaded data also covers every store issued before it).
                        const int so = t - 1 - H;
                        const bool sv = so >= za && so < zb;
                        if (!STAGED) {
                            if (own_row && sv) {
                                u32 *dst = (u32 *)(out + (int64_t)so * sw + rowoff);
#pragma unroll
                                for (int kk = 0; kk < W; kk++)
                                    if (w0 + kk < wx)
                                        *(uint2 *)(dst + 2 * (w0 + kk)) =
                                            make_uint2(X[p ^ 1][H][kk][0] & ~mu[kk][0], X[p ^ 1][H][kk][1] & ~mu[kk][1]);
                            }
                            FW_LOAD32(X[p ^ 1][0], t + 1);      // in flight during the other levels (clamped, so always legal)
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);      // levels one after the other: interleaving them only costs registers
                }
            }
        }
    }
}
#undef FW_LOAD32
#undef FW_LAND
#undef FW_ISSUE

template <int H>
static int morph_wave_launch(const u64 *in, u64 *out, int nz, int ny, int nx, int wx, u32 ops, hipStream_t s)
{
    const int rows_own = 64 - 2 * H;
    const int nxt = (wx + FW_W - 1) / FW_W, nyt = (ny + rows_own - 1) / rows_own;
    // z chunks: the kernel holds two waves per SIMD (VGPR-bound) and a wave's time is its slice count, so the waves
    // should fill the machine's wave slots in ONE round -- a second, mostly empty round costs as much as the first.
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = (cus > 0 ? cus : 256) * 4 * 2;
    }
    const int64_t cols = (int64_t)nxt * nyt;
    int64_t chunks = cols >= slots ? 1 : slots / cols;
    int zchunk = (int)ceil_div64(nz, chunks);
    if (zchunk < 16) zchunk = nz < 16 ? nz : 16;             // below this the 2 H halo slices dominate
    const int64_t nwaves = cols * ceil_div64(nz, zchunk);
    const int64_t blocks = ceil_div64(nwaves, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
#define FW_GO(OPSC) hipLaunchKernelGGL((morph_wave_kernel<H, OPSC>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, nz, ny, nx, \
                                       wx, ops, zchunk, nxt, nyt, nwaves)
#define FW_GO32(OPSC)                                                                                                      \
    {                                                                                                                      \
        if (staged) hipLaunchKernelGGL((morph_wave32_kernel<H, OPSC, true>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, nz, \
                                       ny, nx, wx, zchunk, nxt, nyt, nwaves);                                              \
        else hipLaunchKernelGGL((morph_wave32_kernel<H, OPSC, false>), dim3((unsigned)blocks), dim3(256), 0, s, in, out, nz, ny,  \
                                nx, wx, zchunk, nxt, nyt, nwaves);                                                         \
    }
    // TOMO_MORPH_PATH = generic | direct : A/B switches for the tests (default: the fastest path that applies)
    static const char *force = getenv("TOMO_MORPH_PATH");
    const bool generic_only = force && force[0] == 'g';
    const bool staged = !(force && force[0] == 'd') && (wx % 2 == 0) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    // the pass masks of smooth_voxel_data: E D D E (opening + first closing), D E D E (two closings), D E, E D
    if (generic_only) FW_GO(-1);
    else if (H == 4 && ops == 6u) FW_GO32(6)
    else if (H == 4 && ops == 5u) FW_GO32(5)
    else if (H == 2 && ops == 1u) FW_GO32(1)
    else if (H == 2 && ops == 2u) FW_GO32(2)
    else FW_GO(-1);
#undef FW_GO32
#undef FW_GO
    return tomo_status();
}

TOMO_API int tomo_morph_fused(const uint64_t *in, uint64_t *out, int nz, int ny, int nx, uint32_t ops, int nops,
                              void *stream)
{
    if (!in || !out || in == out || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    if (nops != 2 && nops != 4 && nops != 6 && nops != 8) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    if (nops == 2) return morph_wave_launch<2>((const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, (hipStream_t)stream);
    if (nops == 4) return morph_wave_launch<4>((const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, (hipStream_t)stream);
    int xh = wx > FM_TW ? 1 : 0;
    int TWH = (wx > FM_TW ? FM_TW : wx) + 2 * xh;
    int TYH = 1024 / TWH;
    if (TYH > 64) TYH = 64;
    int rows_own = TYH - 2 * nops;
    if (rows_own < 4) return TOMO_E_SIZE;
    if (rows_own > ny) { rows_own = ny; TYH = rows_own + 2 * nops; }
    int zchunk = nz;
    // enough blocks to fill the chip: split z
    int64_t tiles = ceil_div64(wx, TWH - 2 * xh) * ceil_div64(ny, rows_own);
    while (zchunk > 32 && tiles * ceil_div64(nz, zchunk) < 512) zchunk = (zchunk + 1) / 2;
    dim3 grid((unsigned)ceil_div64(wx, TWH - 2 * xh), (unsigned)ceil_div64(ny, rows_own), (unsigned)ceil_div64(nz, zchunk));
    dim3 block((unsigned)TWH, (unsigned)TYH);
    size_t lds = (size_t)2 * TYH * TWH * sizeof(u64);
    hipStream_t s = (hipStream_t)stream;
    switch (nops) {
    case 2: if (TWH == 16) hipLaunchKernelGGL((morph_fused_kernel<2, true>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        else hipLaunchKernelGGL((morph_fused_kernel<2, false>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        break;
    case 4: if (TWH == 16) hipLaunchKernelGGL((morph_fused_kernel<4, true>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        else hipLaunchKernelGGL((morph_fused_kernel<4, false>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        break;
    case 6: if (TWH == 16) hipLaunchKernelGGL((morph_fused_kernel<6, true>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        else hipLaunchKernelGGL((morph_fused_kernel<6, false>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        break;
    default: if (TWH == 16) hipLaunchKernelGGL((morph_fused_kernel<8, true>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        else hipLaunchKernelGGL((morph_fused_kernel<8, false>), grid, block, lds, s, (const u64 *)in, (u64 *)out, nz, ny, nx, wx, ops, rows_own, zchunk);
        break;
    }
    return tomo_status();
}

// ------------------------------------------------------------------------------------------
// extended bit volume: ext[ez][ey][e]  <->  padded voxel (Z,Y,X) = (ez-2, ey-2, e-(4-pad)) with the
// 'reflect' rule of scipy applied on the PADDED array and zeros in the pad ring.  Data column x sits at
// ext bit x+4 (nibble aligned), so interior words are a 4-bit funnel shift of the source words.
__device__ static inline int ext_src_index(int E, int n, int pad)
{   // ext index (already minus 2) on an axis of n data voxels -> data index or -1 (zero)
    int N = n + 2 * pad;
    int r = reflect_index(E, N) - pad;
    return (r >= 0 && r < n) ? r : -1;
}

__global__ __launch_bounds__(256) void extend_kernel(const u64 *__restrict__ bits, u64 *__restrict__ ext, int nz, int ny,
                                                     int nx, int pad, int wx, int EZ, int EY, int EWX)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)EZ * EY * EWX;
    if (i >= total) return;
    int W = (int)(i % EWX);
    int64_t r = i / EWX;
    int ey = (int)(r % EY), ez = (int)(r / EY);
    int z = ext_src_index(ez - 2, nz, pad), y = ext_src_index(ey - 2, ny, pad);
    u64 res = 0;
    if (z >= 0 && y >= 0) {
        const u64 *row = bits + ((int64_t)z * ny + y) * wx;
        u64 cur = W < wx ? row[W] : 0ull;
        u64 prv = (W >= 1 && W - 1 < wx) ? row[W - 1] : 0ull;
        const int sh = 4 + pad;                  // ext bit e = padded X + 4 = data x + pad + 4 (source tail bits are zero)
        res = (cur << sh) | (prv >> (64 - sh));
        // the four reflected columns X = -2, -1, Nx, Nx+1 of the padded array
        const int Nx = nx + 2 * pad;
        const int XS[4] = {-2, -1, Nx, Nx + 1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int e = XS[k] + 4;
            if ((e >> 6) != W) continue;
            int x = ext_src_index(XS[k], nx, pad);
            u64 bit = x >= 0 ? (row[x >> 6] >> (x & 63)) & 1ull : 0ull;
            res = (res & ~(1ull << (e & 63))) | (bit << (e & 63));
        }
    }
    ext[i] = res;
}

TOMO_API int tomo_extend_bits(const uint64_t *bits, uint64_t *ext, int nz, int ny, int nx, int pad, void *stream)
{
    if (!bits || !ext || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1)) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    int EZ = (int)tomo_ext_slices(nz, pad), EY = (int)tomo_ext_rows(ny, pad), EWX = (int)tomo_ext_words_per_row(nx, pad);
    int64_t total = (int64_t)EZ * EY * EWX;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(extend_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u64 *)bits,
                       (u64 *)ext, nz, ny, nx, pad, wx, EZ, EY, EWX);
    return tomo_status();
}
