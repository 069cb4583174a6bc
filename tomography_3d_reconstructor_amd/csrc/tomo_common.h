// tomo_common.h -- shared helpers for the gfx950 kernels of libtomo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tomo_hip.h"

#define TOMO_API extern "C" __attribute__((visibility("default")))
#define WAVE 64
// Marching-cubes segment s = float columns [256 s - SEG_SHIFT, 256 s - SEG_SHIFT + 256) of a field row, so that
// wave w of the field kernel (columns 32 + 256 w ...) is exactly segment w + 1.
#define SEG_SHIFT 224

// vertex / voxel key = (row << TOMO_KEY_ROW_SHIFT) | (X << 2) | slot, row = Z * Ny + Y (mc.hip, mesh.hip)
#define TOMO_KEY_ROW_SHIFT 22

// Sort segments of the mc3 chain (mc.hip writes their offsets, mesh.hip sorts inside them): per slice Z the in-plane
// vertices cut into bands of tomo_sort_band(Ny) owner rows, then the between-plane vertices -- TOMO_SORT_NB(Ny) + 1 segments.
// The order inside a plane is local to an owner ROW, so any band height is correct.  Round 4: the hand-written sort (mesh.hip,
// uq3_sortrank_kernel) keeps a segment in LDS -- 2 048 entries in its small variant -- and does not mind many segments: bands of
// 512 rows from 640 rows on (a 1024^2 plane of the bench ellipsoid holds ~2 300 vertices, a 2048^2 one ~3 900).  (Rounds 2-3,
// rocPRIM's segmented sort, one workgroup per segment: whole planes up to 1280 rows -- its time grows with the NUMBER of
// segments, 32-row bands at 1024^3: 166 us against 85 -- and bands of 512 above: beyond 4 096 entries per segment it falls
// off a cliff, 922 us for 3.7 M vertices at 2048^2.  It is still the path for segments too long for LDS.)
__host__ __device__ static inline int tomo_sort_band(int Ny) { return Ny <= 640 ? (Ny > 0 ? Ny : 1) : 512; }
#define TOMO_SORT_NB(Ny) (((Ny) + tomo_sort_band(Ny) - 1) / tomo_sort_band(Ny))

typedef unsigned long long u64;
typedef unsigned int u32;

static inline int tomo_status()
{
    return hipGetLastError() == hipSuccess ? TOMO_OK : TOMO_E_LAUNCH;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// scipy.ndimage 'reflect' (half-sample symmetric) index map onto [0, n): -1 -> 0, -2 -> 1, n -> n-1, ...
__host__ __device__ static inline int reflect_index(int i, int n)
{
    int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

// ---- wave-level helpers (wave64) ----------------------------------------------------------
// value of lane-1 (lane 0 gets `fill`)
__device__ static inline int dpp_from_prev(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
// value of lane+1 (lane 63 gets `fill`)
__device__ static inline int dpp_from_next(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
__device__ static inline double dpp_from_prev_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(dpp_from_prev(hi, 0), dpp_from_prev(lo, 0));
}
__device__ static inline double dpp_from_next_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(dpp_from_next(hi, 0), dpp_from_next(lo, 0));
}
__device__ static inline float dpp_from_next_f32(float v)
{
    return __int_as_float(dpp_from_next(__float_as_int(v), 0));
}

// inclusive wave scan (sum) of a u32 across the 64 lanes
__device__ static inline u32 wave_inclusive_scan(u32 v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}
__device__ static inline u32 wave_sum(u32 v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ static inline u64 wave_sum64(u64 v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
