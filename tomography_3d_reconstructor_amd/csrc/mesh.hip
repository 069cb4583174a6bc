// mesh.hip -- vertex finalisation and the np.unique / face-remap stage on the device.
//
// Replaces surface_extractor.py:57-65 (-1 shift, y/x scale), :82-113 (_apply_variable_slice_depths),
// :115-126 (_ensure_manifold_mesh = np.unique(axis=0, return_inverse) + per-face degenerate filter)
// and :128-149 (mesh volume / surface area).  The final vertex index of the reference is the rank of
// the vertex row in the lexicographic (z,y,x) order of the float32 rows, so this is a sort problem:
// an LSD radix sort (rocPRIM device primitive) of order-preserving uint keys, x first, then (z,y).
#include <cstring>
#include <cstdlib>
#include "tomo_common.h"
#include <rocprim/rocprim.hpp>

// ------------------------------------------------------------------------------------------ S5-S7
__global__ __launch_bounds__(256) void vertex_finalize_kernel(float *__restrict__ vpos, int64_t nv, int shift,
                                                              const double *__restrict__ cum, int64_t ncum,
                                                              const double *__restrict__ adj, int64_t nadj, float mm_y,
                                                              float mm_x)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    float *p = vpos + 3 * i;
    float z = p[0], y = p[1], x = p[2];
    if (shift) { z -= 1.0f; y -= 1.0f; x -= 1.0f; }
    if (nadj > 0) {
        if (z < 0.0f) z = 0.0f;
        else if ((double)z >= (double)(ncum - 1)) z = (float)cum[ncum - 1];
        else {
            int64_t lo = (int64_t)floorf(z);
            float frac = z - (float)lo;
            int64_t k = lo < nadj - 1 ? lo : nadj - 1;
            z = (float)(cum[lo] + (double)frac * adj[k]);
        }
    }
    p[0] = z; p[1] = y * mm_y; p[2] = x * mm_x;
}

TOMO_API int tomo_vertex_finalize(float *vpos, int64_t nv, int shift, const double *cum, int64_t ncum, const double *adj,
                                  int64_t nadj, float mm_y, float mm_x, void *stream)
{
    if (nv < 0 || (nv > 0 && !vpos) || (nadj > 0 && (!cum || !adj || ncum != nadj + 1))) return TOMO_E_ARG;
    if (nv == 0) return TOMO_OK;
    int64_t blocks = ceil_div64(nv, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(vertex_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vpos, nv, shift,
                       cum, ncum, adj, nadj, mm_y, mm_x);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ unique
__device__ static inline u32 fkey32(float f)
{   // order-preserving float -> uint
    u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void uq_init_kernel(const float *__restrict__ vpos, int64_t nv, u32 *__restrict__ kx,
                                                      u32 *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    kx[i] = fkey32(vpos[3 * i + 2]);
    idx[i] = (u32)i;
}

__global__ __launch_bounds__(256) void uq_gather_kernel(const float *__restrict__ vpos, int64_t nv,
                                                        const u32 *__restrict__ idx, u64 *__restrict__ kzy)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const float *p = vpos + 3 * (int64_t)idx[i];
    kzy[i] = ((u64)fkey32(p[0]) << 32) | (u64)fkey32(p[1]);
}

// 48-bit keys of the one-sort path: (bucket, sub).  bucket = 2 Z for a vertex in slice plane Z of its owner voxel
// (x- / y-edge: sub = y), 2 Z + 1 for a vertex between planes Z and Z + 1 (z-edge or cell centre: sub = z).
__global__ __launch_bounds__(256) void uq_keys_bucket_kernel(const float *__restrict__ vpos, const u64 *__restrict__ vkey,
                                                             int64_t nv, int key_row_shift, int Ny, u64 *__restrict__ keys,
                                                             u32 *__restrict__ idx)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const u64 k = vkey[i];
    const int slot = (int)(k & 3ull);
    const u64 Z = (k >> key_row_shift) / (u64)Ny;
    const float *p = vpos + 3 * i;
    const u64 bucket = 2ull * Z + (slot >= 2 ? 1ull : 0ull);
    keys[i] = (bucket << 32) | (u64)fkey32(slot >= 2 ? p[0] : p[1]);
    idx[i] = (u32)i;
}

// head[i] = row idx[i] differs from row idx[i-1]; with `violations` also counts the places where two consecutive
// rows DEscend in the lexicographic (z, y, x) order (the one-sort path is only valid when there are none)
__global__ __launch_bounds__(256) void uq_heads_kernel(const float *__restrict__ vpos, int64_t nv,
                                                       const u32 *__restrict__ idx, u32 *__restrict__ head,
                                                       u64 *__restrict__ violations)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    u32 h = 1;
    if (i > 0) {
        const float *a = vpos + 3 * (int64_t)idx[i], *b = vpos + 3 * (int64_t)idx[i - 1];
        h = (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) ? 1u : 0u;
        if (violations && (a[0] < b[0] || (a[0] == b[0] && (a[1] < b[1] || (a[1] == b[1] && a[2] < b[2])))))
            atomicAdd(violations, 1ull);
    }
    head[i] = h;
}

// head flag of sorted position i as a function object: lets the scan read the flags through a transform iterator
// instead of an array (one kernel and 14 MB of traffic less per 3.5 M vertices)
// Sorted order of the one-sort path: idx, except for positions [off[1], off[3]) -- the between-planes part of slab 0 and
// the in-plane part of slab 1 -- which are read from `alt`, where uq_merge_kernel has merged those two runs (off == null:
// plain idx).
struct UqOrder {
    const u32 *idx, *alt, *off;
    __device__ u32 at(u32 i) const
    {
        if (off != nullptr && i >= off[1] && i < off[3]) return alt[i];
        return idx[i];
    }
};

struct UqHead {
    const float *vpos;
    UqOrder ord;
    __device__ u32 operator()(u32 i) const
    {
        if (i == 0) return 1u;
        const float *a = vpos + 3 * (int64_t)ord.at(i), *b = vpos + 3 * (int64_t)ord.at(i - 1);
        return (a[0] != b[0] || a[1] != b[1] || a[2] != b[2]) ? 1u : 0u;
    }
};

// With padding, the vertices on the z edges between padded slices 0 and 1 (a mask that touches the first slice) all get
// z' = 0 from the slice-depth map (surface_extractor.py:100-101 clamps z < 0), the same z' as the in-plane vertices of
// slice 1: two runs, each sorted by (z', y, x), that np.unique interleaves.  One thread per element of either run finds
// its rank in the other by binary search (ties: the first run first) -- a merge that is the identity whenever the first
// run's z' lie below the second's, i.e. for every other pair of neighbouring buckets and without the clamp.
__global__ __launch_bounds__(256) void uq_merge_kernel(const float *__restrict__ vpos, const u32 *__restrict__ off,
                                                       const u32 *__restrict__ idx, u32 *__restrict__ alt)
{
    const u32 o1 = off[1], o2 = off[2], o3 = off[3];
    const u32 na = o2 - o1, nb = o3 - o2;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < na + nb; t += gridDim.x * blockDim.x) {
        const bool inA = t < na;
        const u32 src = idx[inA ? o1 + t : o2 + (t - na)];
        const float *k = vpos + 3 * (int64_t)src;
        const float k0 = k[0], k1 = k[1], k2 = k[2];
        const u32 base = inA ? o2 : o1;
        u32 lo = 0, hi = inA ? nb : na;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            const float *m = vpos + 3 * (int64_t)idx[base + mid];
            const bool less = m[0] < k0 || (m[0] == k0 && (m[1] < k1 || (m[1] == k1 && m[2] < k2)));
            const bool equal = m[0] == k0 && m[1] == k1 && m[2] == k2;
            if (less || (!inA && equal)) lo = mid + 1; else hi = mid;      // B counts the A elements <= itself
        }
        alt[o1 + (inA ? t : t - na) + lo] = src;
    }
}

// scatter of the one-sort path: recomputes the head flag (same two rows the order check needs anyway), counts the places
// where the sorted result descends lexicographically, writes uniq / rank
__global__ __launch_bounds__(256) void uq_scatter_check_kernel(const float *__restrict__ vpos, int64_t nv,
                                                               const UqOrder ord, const u32 *__restrict__ hscan,
                                                               float *__restrict__ uniq, int32_t *__restrict__ rank,
                                                               u64 *__restrict__ totals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const u32 src = ord.at((u32)i);
    const float *a = vpos + 3 * (int64_t)src;
    const float a0 = a[0], a1 = a[1], a2 = a[2];
    bool head = true;
    if (i > 0) {
        const float *b = vpos + 3 * (int64_t)ord.at((u32)i - 1u);
        const float b0 = b[0], b1 = b[1], b2 = b[2];
        head = (a0 != b0 || a1 != b1 || a2 != b2);
        if (a0 < b0 || (a0 == b0 && (a1 < b1 || (a1 == b1 && a2 < b2)))) atomicAdd(&totals[2], 1ull);
    }
    const u32 u = hscan[i] - 1u;
    rank[src] = (int32_t)u;
    if (head) {
        float *q = uniq + 3 * (int64_t)u;
        q[0] = a0; q[1] = a1; q[2] = a2;
    }
    if (i == nv - 1) totals[0] = (u64)hscan[i];
}

__global__ __launch_bounds__(256) void uq_scatter_kernel(const float *__restrict__ vpos, int64_t nv,
                                                         const u32 *__restrict__ idx, const u32 *__restrict__ head,
                                                         const u32 *__restrict__ hscan, float *__restrict__ uniq,
                                                         int32_t *__restrict__ rank, u64 *__restrict__ totals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    u32 u = hscan[i] - 1u;
    u32 src = idx[i];
    rank[src] = (int32_t)u;
    if (head[i]) {
        const float *p = vpos + 3 * (int64_t)src;
        float *q = uniq + 3 * (int64_t)u;
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
    }
    if (i == nv - 1) totals[0] = (u64)hscan[i];
}

// ---- one-sort path, bucket partition + segmented sort
// class of a vertex from its key: 1 = between two slice planes (z-edge / cell centre), 0 = in a plane
struct UqBetween {
    __host__ __device__ u32 operator()(u64 key) const { return (key & 3ull) >= 2ull ? 1u : 0u; }
};

// slab_start[Z] = index of the first vertex whose owner voxel is in slice >= Z (vertices arrive ordered by slice);
// slab_start[0 .. Nz] inclusive, slab_start[Nz] = nv
__global__ __launch_bounds__(256) void uq_slabs_kernel(const u64 *__restrict__ vkey, int64_t nv, int Ny, int Nz,
                                                       u32 *__restrict__ slab_start)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    int64_t Z = (int64_t)((vkey[i] >> TOMO_KEY_ROW_SHIFT) / (u64)Ny);
    if (Z > Nz - 1) Z = Nz - 1;
    int64_t Zp = i == 0 ? -1 : (int64_t)((vkey[i - 1] >> TOMO_KEY_ROW_SHIFT) / (u64)Ny);
    if (Zp > Nz - 1) Zp = Nz - 1;
    for (int64_t z = Zp + 1; z <= Z; z++) slab_start[z] = (u32)i;
    if (i == nv - 1) for (int64_t z = Z + 1; z <= Nz; z++) slab_start[z] = (u32)nv;
}

// stable partition of every slab into [in-plane vertices][between-plane vertices]: B[i] = number of between-plane
// vertices before i.  Writes the 32-bit sub key (y in a plane, z between planes) and the source index at the
// destination, and the segment offsets (2 Z -> plane part of slab Z, 2 Z + 1 -> between part; offsets[2 Nz] = nv).
__global__ __launch_bounds__(256) void uq_partition_kernel(const float *__restrict__ vpos, const u64 *__restrict__ vkey,
                                                           int64_t nv, int Ny, int Nz, const u32 *__restrict__ B,
                                                           const u32 *__restrict__ slab_start, u32 *__restrict__ keys,
                                                           u32 *__restrict__ idx, u32 *__restrict__ offsets)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (int64_t)Nz) {                                   // the first Nz threads also publish the segment offsets
        const u32 s0 = slab_start[i], s1 = slab_start[i + 1];
        const u32 nb = B[s1] - B[s0];
        offsets[2 * i] = s0;
        offsets[2 * i + 1] = s1 - nb;
        if (i == Nz - 1) offsets[2 * (int64_t)Nz] = (u32)nv;
    }
    if (i >= nv) return;
    const u64 k = vkey[i];
    int64_t Z = (int64_t)((k >> TOMO_KEY_ROW_SHIFT) / (u64)Ny);
    if (Z > Nz - 1) Z = Nz - 1;
    const bool between = (k & 3ull) >= 2ull;
    const u32 s0 = slab_start[Z], s1 = slab_start[Z + 1];
    const u32 b0 = B[s0], bi = B[i];
    const u32 nplane = (s1 - s0) - (B[s1] - b0);
    const u32 dest = between ? s0 + nplane + (bi - b0) : s0 + ((u32)i - s0) - (bi - b0);
    const float *p = vpos + 3 * i;
    keys[dest] = fkey32(between ? p[0] : p[1]);
    idx[dest] = (u32)i;
}

__global__ void uq_total_kernel(const u64 *__restrict__ vkey, int64_t nv, u32 *__restrict__ B)
{   // closes the exclusive scan: B[nv] = number of between-plane vertices
    if (threadIdx.x == 0 && blockIdx.x == 0) B[nv] = B[nv - 1] + (((vkey[nv - 1] & 3ull) >= 2ull) ? 1u : 0u);
}

#define UQ_MAX_SLABS 65536          // slices the bucket tables of the one-sort path are sized for

struct UqLayout {
    size_t kx_a, kx_b, idx_a, idx_b, idx_c, kzy_a, kzy_b, head, hscan, seg, temp, temp_bytes, total;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// The segments (one per slice plane / between-planes bucket) hold ~1 000 - 6 000 vertices at 1024^2 slices.  rocPRIM's default
// (128 threads x 17 items) sorts at most 2 176 of them in one go and sends longer segments through several global-memory
// passes of one small block: 512 x 8 keeps every segment up to 4 096 in registers / LDS (measured, unique stage at 1024^3:
// default 0.43 ms, 256x16 0.32, 512x8 0.315, 1024x4 0.36, 512x6 0.35, 512x10 0.36, 256x24 0.39).
// The partitioning threshold (4th WarpSortConfig parameter: the number of SEGMENTS from which rocPRIM first partitions them by
// size) is set out of reach: that path copies its segment counts to the host and waits for them (device_segmented_radix_sort.hpp,
// memcpy_and_sync) -- one hidden host round trip per pass of a chain that is built to have none until its single download.
typedef rocprim::segmented_radix_sort_config<8, rocprim::kernel_config<512, 8>, rocprim::WarpSortConfig<32, 4, 256, 0x7fffffff, 32, 4, 256>, true>
    UqSegCfg;


static UqLayout uq_layout(int64_t nv)
{
    UqLayout L;
    size_t n = (size_t)(nv > 0 ? nv : 1), off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.kx_a = take(n * 4); L.kx_b = take(n * 4);
    L.idx_a = take(n * 4); L.idx_b = take(n * 4); L.idx_c = take(n * 4);
    L.kzy_a = take(n * 8); L.kzy_b = take(n * 8);
    L.head = take(n * 4); L.hscan = take(n * 4);
    L.seg = take((size_t)(3 * UQ_MAX_SLABS + 16) * 4);              // slab_start[Nz + 1] | offsets[2 Nz + 1]
    size_t t1 = 0, t2 = 0, t3 = 0;
    (void)rocprim::radix_sort_pairs(nullptr, t1, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, n, 0, 32,
                              (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, t2, (u64 *)nullptr, (u64 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, n, 0, 64,
                              (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, t3, (u32 *)nullptr, (u32 *)nullptr, n, rocprim::plus<u32>(), (hipStream_t)0);
    size_t t4 = 0;
    auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u32>(0u), UqHead{nullptr, nullptr});
    (void)rocprim::inclusive_scan(nullptr, t4, heads, (u32 *)nullptr, n, rocprim::plus<u32>(), (hipStream_t)0);
    L.temp_bytes = t1 > t2 ? t1 : t2;
    if (t3 > L.temp_bytes) L.temp_bytes = t3;
    if (t4 > L.temp_bytes) L.temp_bytes = t4;
    size_t t5 = 0, t6 = 0;
    (void)rocprim::segmented_radix_sort_pairs(nullptr, t5, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                                              (unsigned)n, (unsigned)(2 * UQ_MAX_SLABS), (const u32 *)nullptr, (const u32 *)nullptr,
                                              0, 32, (hipStream_t)0);
    {
        size_t ta = 0;
        (void)rocprim::segmented_radix_sort_pairs<UqSegCfg>(nullptr, ta, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                                                            (unsigned)n, (unsigned)(2 * UQ_MAX_SLABS), (const u32 *)nullptr,
                                                            (const u32 *)nullptr, 0, 32, (hipStream_t)0);
        if (ta > t5) t5 = ta;
    }
    auto cls = rocprim::make_transform_iterator((const u64 *)nullptr, UqBetween());
    (void)rocprim::exclusive_scan(nullptr, t6, cls, (u32 *)nullptr, 0u, n + 1, rocprim::plus<u32>(), (hipStream_t)0);
    if (t5 > L.temp_bytes) L.temp_bytes = t5;
    if (t6 > L.temp_bytes) L.temp_bytes = t6;
    L.temp = take(L.temp_bytes + 256);
    L.total = off;
    return L;
}

TOMO_API int64_t tomo_mesh_unique_workspace_bytes(int64_t nv) { return (int64_t)uq_layout(nv).total; }

TOMO_API int tomo_mesh_unique(const float *vpos, int64_t nv, float *uniq, int32_t *rank, unsigned long long *totals,
                              void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!vpos || !uniq || !rank || !totals || !workspace || nv <= 0) return TOMO_E_ARG;
    if (nv >= 0x7fffffffll) return TOMO_E_SIZE;
    UqLayout L = uq_layout(nv);
    if ((size_t)workspace_bytes < L.total) return TOMO_E_WORKSPACE;
    char *ws = (char *)workspace;
    u32 *kx_a = (u32 *)(ws + L.kx_a), *kx_b = (u32 *)(ws + L.kx_b);
    u32 *idx_a = (u32 *)(ws + L.idx_a), *idx_b = (u32 *)(ws + L.idx_b), *idx_c = (u32 *)(ws + L.idx_c);
    u64 *kzy_a = (u64 *)(ws + L.kzy_a), *kzy_b = (u64 *)(ws + L.kzy_b);
    u32 *head = (u32 *)(ws + L.head), *hscan = (u32 *)(ws + L.hscan);
    void *temp = ws + L.temp;
    size_t tb = L.temp_bytes;
    hipStream_t s = (hipStream_t)stream;
    unsigned blocks = (unsigned)ceil_div64(nv, 256);
    hipLaunchKernelGGL(uq_init_kernel, dim3(blocks), dim3(256), 0, s, vpos, nv, kx_a, idx_a);
    if (rocprim::radix_sort_pairs(temp, tb, kx_a, kx_b, idx_a, idx_b, (size_t)nv, 0, 32, s) != hipSuccess) return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(uq_gather_kernel, dim3(blocks), dim3(256), 0, s, vpos, nv, (const u32 *)idx_b, kzy_a);
    tb = L.temp_bytes;
    if (rocprim::radix_sort_pairs(temp, tb, kzy_a, kzy_b, idx_b, idx_c, (size_t)nv, 0, 64, s) != hipSuccess) return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(uq_heads_kernel, dim3(blocks), dim3(256), 0, s, vpos, nv, (const u32 *)idx_c, head, (u64 *)nullptr);
    tb = L.temp_bytes;
    if (rocprim::inclusive_scan(temp, tb, head, hscan, (size_t)nv, rocprim::plus<u32>(), s) != hipSuccess) return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(uq_scatter_kernel, dim3(blocks), dim3(256), 0, s, vpos, nv, (const u32 *)idx_c, (const u32 *)head,
                       (const u32 *)hscan, uniq, rank, (u64 *)totals);
    return tomo_status();
}

// One-sort variant for rows that arrive in marching-cubes order (owner voxel z, y, x, then slot) together with their
// vertex keys.  In that order the vertices of a slice plane Z come before the vertices between planes Z and Z + 1, inside
// a plane the rows ascend, inside a row the x-edge vertices (y = the row) ascend in x and precede the y-edge vertices, and
// vertices that share their fractional coordinate ascend in the other two.  So ONE stable radix sort on a 48-bit key
// (bucket 2 Z / 2 Z + 1, then y inside a plane or z between planes) yields the lexicographic order -- 6 digit passes
// instead of the 12 of the general two-sort path.  "Almost": float32 rounding (a fractional coordinate landing exactly
// on a plane / row value) or a zero slice depth can break it; every place where the result descends is counted in
// totals[2] and the caller must then redo the call with tomo_mesh_unique (exact if and only if totals[2] == 0).
// vkey: the keys tomo_mc_emit wrote (row << key_row_shift | x << 2 | slot, row = Z * Ny + Y).  Same workspace size.
TOMO_API int tomo_mesh_unique_presorted(const float *vpos, const unsigned long long *vkey, int64_t nv, int Ny, int Nz,
                                        float *uniq, int32_t *rank, unsigned long long *totals, void *workspace,
                                        int64_t workspace_bytes, void *stream)
{
    if (!vpos || !vkey || !uniq || !rank || !totals || !workspace || nv <= 0 || Ny <= 0) return TOMO_E_ARG;
    if (nv >= 0x7fffffffll) return TOMO_E_SIZE;
    UqLayout L = uq_layout(nv);
    if ((size_t)workspace_bytes < L.total) return TOMO_E_WORKSPACE;
    char *ws = (char *)workspace;
    u32 *idx_b = (u32 *)(ws + L.idx_b), *idx_c = (u32 *)(ws + L.idx_c);
    u64 *kzy_a = (u64 *)(ws + L.kzy_a), *kzy_b = (u64 *)(ws + L.kzy_b);
    u32 *hscan = (u32 *)(ws + L.hscan);
    void *temp = ws + L.temp;
    size_t tb = L.temp_bytes;
    hipStream_t s = (hipStream_t)stream;
    unsigned blocks = (unsigned)ceil_div64(nv, 256);
    UqOrder order{(const u32 *)idx_c, nullptr, nullptr};
    if (Nz <= 0 || Nz > UQ_MAX_SLABS) {
        hipLaunchKernelGGL(uq_keys_bucket_kernel, dim3(blocks), dim3(256), 0, s, vpos, (const u64 *)vkey, nv, TOMO_KEY_ROW_SHIFT, Ny,
                           kzy_a, idx_b);
        if (rocprim::radix_sort_pairs(temp, tb, kzy_a, kzy_b, idx_b, idx_c, (size_t)nv, 0, 48, s) != hipSuccess) return TOMO_E_LAUNCH;
    } else {
        // The bucket (slice, in-plane / between planes) part of that key needs no device-wide sort: vertices arrive grouped
        // by slice, so it is a stable two-way partition inside every slab (one scan + one scatter); what remains is a
        // 32-bit sort INSIDE each of the 2 Nz buckets -- a segmented sort, one pass over the data, no look-back chain.
        u32 *kx_a = (u32 *)(ws + L.kx_a), *kx_b = (u32 *)(ws + L.kx_b);
        u32 *Bscan = (u32 *)kzy_a;                                   // nv + 1 entries (the 64-bit key area is free here)
        u32 *slab_start = (u32 *)(ws + L.seg), *offsets = slab_start + UQ_MAX_SLABS + 8;
        unsigned blocks2 = (unsigned)ceil_div64(nv > Nz ? nv : Nz, 256);
        hipLaunchKernelGGL(uq_slabs_kernel, dim3(blocks), dim3(256), 0, s, (const u64 *)vkey, nv, Ny, Nz, slab_start);
        auto cls = rocprim::make_transform_iterator((const u64 *)vkey, UqBetween());
        if (rocprim::exclusive_scan(temp, tb, cls, Bscan, 0u, (size_t)nv, rocprim::plus<u32>(), s) != hipSuccess) return TOMO_E_LAUNCH;
        hipLaunchKernelGGL(uq_total_kernel, dim3(1), dim3(64), 0, s, (const u64 *)vkey, nv, Bscan);
        hipLaunchKernelGGL(uq_partition_kernel, dim3(blocks2), dim3(256), 0, s, vpos, (const u64 *)vkey, nv, Ny, Nz,
                           (const u32 *)Bscan, (const u32 *)slab_start, kx_a, idx_b, offsets);
        tb = L.temp_bytes;
        if (rocprim::segmented_radix_sort_pairs<UqSegCfg>(temp, tb, kx_a, kx_b, idx_b, idx_c, (unsigned)nv, (unsigned)(2 * Nz),
                                                          (const u32 *)offsets, (const u32 *)offsets + 1, 0, 32, s) != hipSuccess)
            return TOMO_E_LAUNCH;
        if (Nz >= 2) {                              // the clamped run of slab 0 and the plane of slab 1 (see uq_merge_kernel)
            hipLaunchKernelGGL(uq_merge_kernel, dim3(256), dim3(256), 0, s, vpos, (const u32 *)offsets, (const u32 *)idx_c, idx_b);
            order.alt = idx_b;
            order.off = offsets;
        }
    }
    tb = L.temp_bytes;
    auto heads = rocprim::make_transform_iterator(rocprim::make_counting_iterator<u32>(0u), UqHead{vpos, order});
    if (rocprim::inclusive_scan(temp, tb, heads, hscan, (size_t)nv, rocprim::plus<u32>(), s) != hipSuccess) return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(uq_scatter_check_kernel, dim3(blocks), dim3(256), 0, s, vpos, nv, order, (const u32 *)hscan,
                       uniq, rank, (u64 *)totals);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ mc3: sort + rank
// The unique stage of the mc3 chain (mc.hip): vertices arrive FINALISED as 16-byte records {z', y', x', id}, already
// partitioned into the 2 Nz buckets with their 32-bit sort keys; the offsets (mc3_bands_kernel) cut every plane's bucket
// further into bands of owner rows -- the order inside a plane is local to a row.  One segmented sort inside the segments, the clamped-run merge of the first two buckets (see uq_merge_kernel), and one gather that writes
// the rows in order, table[id] = position, and counts every place where the result does not ascend STRICTLY: with a
// count of zero the sorted position is np.unique's index (no duplicate rows, no rounding coincidence) -- otherwise the
// caller redoes the stage with tomo_mesh_unique on the rows.
__device__ static inline bool rec_less(const float4 &a, const float4 &b)
{
    return a.x < b.x || (a.x == b.x && (a.y < b.y || (a.y == b.y && a.z < b.z)));
}

__global__ __launch_bounds__(256) void uq3_merge_kernel(const float4 *__restrict__ vrec, const u32 *__restrict__ off,
                                                        const u32 *__restrict__ idx, u32 *__restrict__ alt)
{
    const u32 o1 = off[1], o2 = off[2], o3 = off[3];
    const u32 na = o2 - o1, nb = o3 - o2;
    for (u32 t = blockIdx.x * blockDim.x + threadIdx.x; t < na + nb; t += gridDim.x * blockDim.x) {
        const bool inA = t < na;
        const u32 src = idx[inA ? o1 + t : o2 + (t - na)];
        const float4 k = vrec[src];
        const u32 base = inA ? o2 : o1;
        u32 lo = 0, hi = inA ? nb : na;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            const float4 m = vrec[idx[base + mid]];
            const bool less = rec_less(m, k);
            const bool equal = m.x == k.x && m.y == k.y && m.z == k.z;
            if (less || (!inA && equal)) lo = mid + 1; else hi = mid;      // B counts the A elements <= itself
        }
        alt[o1 + (inA ? t : t - na) + lo] = src;
    }
}

__global__ __launch_bounds__(256) void uq3_rank_kernel(const float4 *__restrict__ vrec, int64_t cap_v, const UqOrder ord,
                                                       float *__restrict__ uniq, int32_t *__restrict__ table,
                                                       u64 *__restrict__ tot, const float z_top)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool viol = false;
    const u64 nv = tot[1];
    const bool live = tot[3] == 0 && i < cap_v && (u64)i < nv;
    // A sort that did not yield the lexicographic order (rows tie on the key but not as rows) can leave the merged range
    // with slots the merge never wrote: whatever they hold must not be used as an index.  Such a slot -- or a stale one
    // that repeats an element -- always shows up as a place where the rows do not ascend strictly.
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        u32 sa = ord.at((u32)i);
        if ((u64)sa >= nv) { sa = 0; viol = true; }
        a = vrec[sa];
    }
    // the predecessor's row is the neighbouring lane's own row (one gather per element instead of two); lane 0 fetches it
    float4 b;
    b.x = __shfl_up(a.x, 1, 64); b.y = __shfl_up(a.y, 1, 64); b.z = __shfl_up(a.z, 1, 64);
    if (live) {
        if ((threadIdx.x & 63) == 0 && i > 0) {
            u32 sb = ord.at((u32)i - 1u);
            if ((u64)sb >= nv) { sb = 0; viol = true; }
            b = vrec[sb];
        }
        if (i > 0) viol = viol || !rec_less(b, a);
        typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));      // one 12-byte store per row
        *(f3u *)(uniq + 3 * i) = (f3u){a.x, a.y, a.z};
        table[__float_as_uint(a.w)] = (int32_t)i;
    }
    const u64 n = (u64)__popcll(__ballot(viol));
    if ((threadIdx.x & 63) == 0 && n) atomicAdd((unsigned long long *)&tot[4], (unsigned long long)n);
    // rows on the plane z' == z_top (a Z-slab rank's shared plane with the rank above; NaN: nobody asks) -> tot[7]
    const u64 nt = (u64)__popcll(__ballot(live && a.x == z_top));
    if ((threadIdx.x & 63) == 0 && nt) atomicAdd((unsigned long long *)&tot[7], (unsigned long long)nt);
}

struct Uq3Layout { size_t keys_alt, idx_alt, idx_merge, temp, temp_bytes, total; };

static Uq3Layout uq3_layout(int64_t cap_v, int64_t nseg)
{
    Uq3Layout L;
    size_t n = (size_t)(cap_v > 0 ? cap_v : 1), off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.keys_alt = take(n * 4); L.idx_alt = take(n * 4); L.idx_merge = take(n * 4);
    size_t t = 0;
    (void)rocprim::segmented_radix_sort_pairs<UqSegCfg>(nullptr, t, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr, (u32 *)nullptr,
                                                        (unsigned)n, (unsigned)(nseg > 0 ? nseg : 1), (const u32 *)nullptr,
                                                        (const u32 *)nullptr, 0, 32, (hipStream_t)0);
    L.temp_bytes = t;
    L.temp = take(t + 256);
    L.total = off;
    return L;
}

TOMO_API int64_t tomo_mc3_sort_workspace_bytes(int64_t cap_v, int64_t nseg) { return (int64_t)uq3_layout(cap_v, nseg).total; }

TOMO_API int tomo_mc3_sort_rank_top(const float *vrec, uint32_t *keys, uint32_t *idx, int64_t cap_v, int Nz, int Ny,
                                    const uint32_t *slice_tab, unsigned long long *tot, float *uniq, int32_t *table, void *workspace,
                                    int64_t workspace_bytes, float z_top, void *stream)
{
    if (!vrec || !keys || !idx || !slice_tab || !tot || !uniq || !table || !workspace || cap_v <= 0 || Nz < 1 || Ny < 1) return TOMO_E_ARG;
    if (cap_v >= 0x7fffffffll || Nz > UQ_MAX_SLABS) return TOMO_E_SIZE;
    // segments: per slice TOMO_SORT_NB(Ny) bands of owner rows of the plane, then the between-plane bucket (mc3_bands_kernel)
    const int64_t nseg = (int64_t)(TOMO_SORT_NB(Ny) + 1) * Nz;
    if (nseg >= 0x7fffffffll) return TOMO_E_SIZE;
    Uq3Layout L = uq3_layout(cap_v, nseg);
    if ((size_t)workspace_bytes < L.total) return TOMO_E_WORKSPACE;
    char *ws = (char *)workspace;
    u32 *keys_alt = (u32 *)(ws + L.keys_alt), *idx_alt = (u32 *)(ws + L.idx_alt), *idx_merge = (u32 *)(ws + L.idx_merge);
    const u32 *offsets = slice_tab + 2 * ((int64_t)Nz + 1);
    const u32 *merge3 = offsets + nseg + 1;
    hipStream_t s = (hipStream_t)stream;
    size_t tb = L.temp_bytes;
    if (rocprim::segmented_radix_sort_pairs<UqSegCfg>(ws + L.temp, tb, keys, keys_alt, idx, idx_alt, (unsigned)cap_v, (unsigned)nseg,
                                                      offsets, offsets + 1, 0, 32, s) != hipSuccess)
        return TOMO_E_LAUNCH;
    UqOrder order{(const u32 *)idx_alt, nullptr, nullptr};
    if (Nz >= 2) {                                  // (UqOrder and the merge kernel read off[1], off[2], off[3])
        hipLaunchKernelGGL(uq3_merge_kernel, dim3(256), dim3(256), 0, s, (const float4 *)vrec, merge3 - 1, (const u32 *)idx_alt, idx_merge);
        order.alt = idx_merge;
        order.off = merge3 - 1;
    }
    hipLaunchKernelGGL(uq3_rank_kernel, dim3((unsigned)ceil_div64(cap_v, 256)), dim3(256), 0, s, (const float4 *)vrec, cap_v, order,
                       uniq, table, (u64 *)tot, z_top);
    return tomo_status();
}

TOMO_API int tomo_mc3_sort_rank(const float *vrec, uint32_t *keys, uint32_t *idx, int64_t cap_v, int Nz, int Ny, const uint32_t *slice_tab,
                                unsigned long long *tot, float *uniq, int32_t *table, void *workspace, int64_t workspace_bytes,
                                void *stream)
{
    return tomo_mc3_sort_rank_top(vrec, keys, idx, cap_v, Nz, Ny, slice_tab, tot, uniq, table, workspace, workspace_bytes,
                                  __builtin_nanf(""), stream);
}

// ------------------------------------------------------------------------------------------ mc3: sort + rank in ONE kernel (round 4)
// The default path of the unique stage since round 4: 97 + 5 us at 1024^3 against 83 + 5 + 45 for rocPRIM's segmented sort +
// uq3_merge_kernel + uq3_rank_kernel, which stay as the path for segments too long for LDS (pipeline.FUSED_SORT /
// TOMO_FUSED_SORT=0: the A/B switch).  One workgroup per sort segment -- a plane band or a between-plane bucket, the segments of
// mc3_bands_kernel:
//   1. the segment's 32-bit keys become 64-bit words key << 32 | position (the position breaks ties the way a stable sort
//      would: vertices arrive in cell scan order); every wave sorts runs of 512 words in registers (8 per lane, a bitonic network:
//      partners at a distance < 8 are registers of the same lane, the others sit in lane ^ m -- DPP for m = 1, 2, 4, 8, ds_bpermute
//      for 16 and 32) -- no LDS buffer, no barrier;
//   2. merge rounds in LDS: a thread finds where its 8 consecutive OUTPUTS begin in the two runs being merged (one binary
//      search along the merge path) and merges them sequentially; two runs that are already in order (a plane's vertices arrive
//      grouped by owner row) are copied.  Words are stored with one word of padding per 8 (a thread's window starts 72 B after
//      its neighbour's: 2-way instead of 16-way bank conflicts);
//   3. every gather of a thread's rows is issued at once, the rows pass through LDS (the predecessor of a row is its neighbour
//      there), then final rows, table[id] = position, and the count of places that do not ascend STRICTLY (tot[4]);
//   4. uq3_seams_kernel checks the seams BETWEEN segments.
// Measured at 1024^3 (3.5 M vertices in ~2 600 non-empty segments of ~1 200, up to 2 400 near the poles; profiles/
// r04_sort_experiments.md), launches in us: every element's rank in the sibling run by binary search 171 (LDS pipeline);
// merge-path rounds from runs of 8: 85 + 73 for a second launch that takes the segments beyond 2 048 -- ~130 of them, nothing
// overlaps it; runs of 512 sorted in registers first: 74 + 64; a counting sort over the key range with in-bin ranking 80 + 70
// (the voxelised surface has terraces: hundreds of exactly equal keys in one bin); one WAVE per segment, everything in registers
// (a bitonic network over up to 64 words per lane): 149 + 113 (200 VGPRs, 64-bit compare-exchanges at VALU rate); this version
// with 2 560 entries in the first launch (46 KiB: three workgroups per CU, every segment of the bench workloads): 104 + 5, and
// with the first and the last slices dispatched first: 100 + 5; ONE launch with 2 048 entries (36 KiB, four workgroups per CU) in
// which a longer segment is sorted as two halves + one merge (this version): 97 -- those ~130 workgroups are its critical path.
// All of them are bound by chains of dependent LDS accesses behind barriers.  A ticket + __threadfence per workgroup for a "last one checks the seams" cost 480 us (the fence writes the
// L2 back, 3 000 times): hence the separate seam kernel.
// The clamped run of a padded stack -- the between-plane bucket of slice 0 and the plane of slice 1 have the SAME z' when the
// depth map clamps z < 0 to 0 (uq3_merge_kernel merges them in the rocPRIM path) -- is ONE segment here, ordered by (y', x')
// through two stable passes (x', then y').  Segments beyond SR_CAP entries (1 024 for the clamped run) set bit 8 of
// tot[3]: the host repeats the stage on the rocPRIM path and remembers it for this geometry (pipeline._MC3_LARGE).
#define SR_THREADS 256
#define SR_E 8                       // words per thread and round
#define SR_CAP 4096                  // longest segment the kernel takes (two halves)
#define SR_HALF 2048                 // entries it sorts in one go (36 KiB of LDS: four workgroups per CU)
#define SR_PAD(i) ((i) + ((i) >> 3))
#define SR_WORDS(cap) ((cap) + ((cap) >> 3))
#define SR_RUN (64 * SR_E)           // words a wave sorts in registers

// the word of lane ^ M: DPP where the pattern is one (quad_perm for 1 and 2, a rotation by 8 inside a row of 16 for 8, the two
// row shifts by 4 for 4) -- VALU rate, no LDS crossbar; ds_bpermute for 16 and 32 (3 of the 21 cross-lane stages)
template <int M>
__device__ static inline u32 sr_lane_xor(u32 x, const int lane)
{
    const int v = (int)x;
    if (M == 1) return (u32)__builtin_amdgcn_update_dpp(v, v, 0xB1 /* quad_perm:[1,0,3,2] */, 0xf, 0xf, false);
    if (M == 2) return (u32)__builtin_amdgcn_update_dpp(v, v, 0x4E /* quad_perm:[2,3,0,1] */, 0xf, 0xf, false);
    if (M == 8) return (u32)__builtin_amdgcn_update_dpp(v, v, 0x128 /* row_ror:8 */, 0xf, 0xf, false);
    if (M == 4) {
        const int up = __builtin_amdgcn_update_dpp(v, v, 0x104 /* row_shl:4: lane + 4 */, 0xf, 0xf, false);
        const int dn = __builtin_amdgcn_update_dpp(v, v, 0x114 /* row_shr:4: lane - 4 */, 0xf, 0xf, false);
        return (u32)((lane & 4) ? dn : up);
    }
    return (u32)__shfl_xor(v, M, 64);
}
template <int M>
__device__ static inline u64 sr_shfl_xor(u64 v, const int lane)
{
    return ((u64)sr_lane_xor<M>((u32)(v >> 32), lane) << 32) | sr_lane_xor<M>((u32)v, lane);
}

// Bitonic sorting network over the 512 words of a wave, word e = 8 * lane + r in register v[r]: stage (K, J) compares words at
// distance J inside blocks of K that alternate between ascending and descending.
template <int K, int J>
__device__ static inline void sr_stage(u64 (&v)[SR_E], const int lane)
{
    if (J >= SR_E) {
        constexpr int M = J / SR_E;
        const bool keep_min = ((((lane * SR_E) & K) == 0) == ((lane & M) == 0));            // (ascending block) == (lower partner)
#pragma unroll
        for (int r = 0; r < SR_E; r++) {
            const u64 o = sr_shfl_xor<(M > 0 ? M : 1)>(v[r], lane);
            v[r] = (keep_min == (o < v[r])) ? o : v[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < SR_E; r++) {
            if ((r & J) == 0) {
                const bool up = K < SR_E ? ((r & K) == 0) : (((lane * SR_E) & K) == 0);
                const u64 x = v[r], y = v[r | J];
                const bool sw = (y < x) == up;                                             // out of order for this block's direction
                v[r] = sw ? y : x;
                v[r | J] = sw ? x : y;
            }
        }
    }
}
template <int K, int J>
struct SrLevel {
    __device__ static inline void run(u64 (&v)[SR_E], const int lane)
    {
        sr_stage<K, J>(v, lane);
        SrLevel<K, J / 2>::run(v, lane);
    }
};
template <int K>
struct SrLevel<K, 0> {
    __device__ static inline void run(u64 (&)[SR_E], const int) {}
};
template <int K>
struct SrNet {
    __device__ static inline void run(u64 (&v)[SR_E], const int lane)
    {
        SrNet<K / 2>::run(v, lane);
        SrLevel<K, K / 2>::run(v, lane);
    }
};
template <>
struct SrNet<1> {
    __device__ static inline void run(u64 (&)[SR_E], const int) {}
};
__device__ static inline void sr_wave_sort(u64 (&v)[SR_E], const int lane) { SrNet<SR_RUN>::run(v, lane); }

// Sorts n distinct words ascending.  make(i) -> word i (called for i < n only); a, b: padded LDS buffers (SR_PAD) with room for
// n rounded up to a whole run.  Every thread of the workgroup calls it; -> the buffer that holds the result.  Ends with a barrier.
template <typename Make>
__device__ static u64 *sr_sort(u64 *a, u64 *b, const int n, Make make)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int base = wave * SR_RUN; base < n; base += (SR_THREADS / 64) * SR_RUN) {          // (uniform per wave: shuffles need every lane)
        const int g0 = base + lane * SR_E;
        u64 v[SR_E];
#pragma unroll
        for (int k = 0; k < SR_E; k++) v[k] = g0 + k < n ? make(g0 + k) : ~0ull;
        sr_wave_sort(v, lane);
#pragma unroll
        for (int k = 0; k < SR_E; k++) a[SR_PAD(g0) + k] = v[k];              // (the padding past n sorts to the end of the run: never read)
    }
    u64 *src = a, *dst = b;
    for (int L = SR_RUN; L < n; L <<= 1) {
        __syncthreads();
        for (int g0 = tid * SR_E; g0 < n; g0 += SR_THREADS * SR_E) {
            // runs A = [pb, pb + la), B = [pb + L, pb + L + lb); this thread writes outputs g0 .. g0 + 7 of their merge
            const int pb = g0 & ~(2 * L - 1);
            const int la = n - pb < L ? n - pb : L;
            const int lb = n - pb - L < 0 ? 0 : (n - pb - L < L ? n - pb - L : L);
            const int a0 = pb, b0 = pb + L, d = g0 - pb;
            u64 out[SR_E];
            if (lb == 0 || src[SR_PAD(a0 + la - 1)] < src[SR_PAD(b0)]) {
                // already in order (a plane's vertices arrive grouped by owner row: its runs rarely interleave): a copy
#pragma unroll
                for (int k = 0; k < SR_E; k++) out[k] = src[SR_PAD(g0) + k];
            } else {
                int lo = d - lb > 0 ? d - lb : 0, hi = d < la ? d : la;
                while (lo < hi) {                                   // the merge path crosses diagonal d at (lo, d - lo)
                    const int mid = (lo + hi) >> 1;
                    if (src[SR_PAD(a0 + mid)] < src[SR_PAD(b0 + d - 1 - mid)]) lo = mid + 1; else hi = mid;
                }
                int ia = lo, ib = d - lo;
                u64 va = ia < la ? src[SR_PAD(a0 + ia)] : ~0ull, vb = ib < lb ? src[SR_PAD(b0 + ib)] : ~0ull;
#pragma unroll
                for (int k = 0; k < SR_E; k++) {
                    const bool ta = va < vb;
                    out[k] = ta ? va : vb;
                    ia += ta ? 1 : 0; ib += ta ? 0 : 1;
                    const int nx = ta ? a0 + ia : b0 + ib;
                    const bool ok = ta ? ia < la : ib < lb;
                    const u64 w = ok ? src[SR_PAD(nx)] : ~0ull;
                    va = ta ? w : va; vb = ta ? vb : w;
                }
            }
#pragma unroll
            for (int k = 0; k < SR_E; k++) dst[SR_PAD(g0) + k] = out[k];       // (past n: ~0 or stale, never read)
        }
        u64 *t = src; src = dst; dst = t;
    }
    __syncthreads();
    return src;
}

// SR_HALF entries of LDS per workgroup (18 B each: the two padded sort buffers, later the gathered rows): 36 KiB, four workgroups
// per CU.  A segment of up to SR_HALF entries is sorted in one go; a longer one (up to SR_CAP: the between-plane buckets at the poles
// of a closed body) as two halves -- the first half's sorted words wait in registers while the second is sorted -- then ONE merge
// whose 16 consecutive outputs per thread go straight to the gathers (no room in LDS for a merged copy: a row's predecessor comes
// from the neighbouring lane).  Those workgroups are dispatched first and overlap the rest.
__global__ __launch_bounds__(SR_THREADS) void uq3_sortrank_kernel(const float4 *__restrict__ vrec, const u32 *__restrict__ keys,
                                                                  const u32 *__restrict__ offsets, const int NB, const int Nz,
                                                                  float *__restrict__ uniq, int32_t *__restrict__ table,
                                                                  u64 *__restrict__ tot, const float z_top)
{
    constexpr int CAP = SR_HALF, EP = CAP / SR_THREADS;
    extern __shared__ __attribute__((aligned(16))) u64 s_buf[];   // 2 * SR_WORDS(CAP) words
    u64 *const s_a = s_buf, *const s_b = s_buf + SR_WORDS(CAP);
    float4 *const s_rows = (float4 *)s_buf;                       // (after the sort: the rows in final order)
    u32 *const s_red = (u32 *)s_b;                                // (used before the buffers are filled)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroups are dispatched in index order: the segments of the first and the last slices go first, alternately (the long
    // between-plane buckets of a closed body sit at its poles; started last they would finish alone)
    const int seg = (blockIdx.x & 1) ? (int)gridDim.x - 1 - (int)(blockIdx.x >> 1) : (int)(blockIdx.x >> 1);
    u32 o0 = offsets[seg];
    int n = (int)(offsets[seg + 1] - o0);
    const u64 nv = tot[1];
    // ---- the clamped run: bucket of slice 0 (segment NB) + plane of slice 1 (segments NB + 1 .. 2 NB)
    bool uni = false;
    if (Nz >= 2 && seg >= NB && seg <= 2 * NB) {
        const u32 o1 = offsets[NB], o2 = offsets[NB + 1], o3 = offsets[2 * NB + 1];
        if (o2 > o1 && o3 > o2) {
            u32 kmax = 0;
            for (u32 i = o1 + tid; i < o2; i += SR_THREADS) kmax = max(kmax, keys[i]);
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) kmax = max(kmax, (u32)__shfl_xor((int)kmax, d, 64));
            if (lane == 0) s_red[wave] = kmax;
            __syncthreads();
            kmax = max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3]));
            if (kmax >= fkey32(vrec[o2].x)) {                     // a row of the bucket does not sort in front of the plane: one run
                if (seg == NB) { uni = true; o0 = o1; n = (int)(o3 - o1); }
                else n = 0;                                       // a band of the plane: the bucket's workgroup takes it
            }
            __syncthreads();
        }
    }
    constexpr int UCAP = CAP / 2 / SR_RUN * SR_RUN;               // the clamped run keeps its first order in one buffer: whole runs per half
    if (n > (uni ? UCAP : SR_CAP)) {
        if (tid == 0) atomicOr((unsigned long long *)&tot[3], 8ull);            // too long: the host takes the rocPRIM path
        return;
    }
    if (n == 0) return;
    if (n > CAP) {
        // ---- a long segment: two half sorts, one merge, rows straight from the merge
        const int nB = n - CAP;
        const u64 *sa = sr_sort(s_a, s_b, CAP, [=](int i) { return ((u64)keys[o0 + i] << 32) | (u32)i; });
        u64 ra[EP], rb[EP];
#pragma unroll
        for (int e = 0; e < EP; e++) ra[e] = sa[SR_PAD(tid + SR_THREADS * e)];
        __syncthreads();
        const u64 *sb = sr_sort(s_a, s_b, nB, [=](int i) { return ((u64)keys[o0 + CAP + i] << 32) | (u32)(CAP + i); });
#pragma unroll
        for (int e = 0; e < EP; e++) rb[e] = sb[SR_PAD(min(tid + SR_THREADS * e, nB - 1))];
        __syncthreads();
        u64 *const M = s_buf;                                    // [first half | second half], one word of padding per 16
#define SR_MP(i) ((i) + ((i) >> 4))
#pragma unroll
        for (int e = 0; e < EP; e++) {
            const int i = tid + SR_THREADS * e;
            M[SR_MP(i)] = ra[e];
            if (i < nB) M[SR_MP(CAP + i)] = rb[e];
        }
        float4 *const s_tail = (float4 *)(s_buf + SR_MP(SR_CAP) + 2);           // the last row of every wave (behind the merge area)
        __syncthreads();
        constexpr int EO = SR_CAP / SR_THREADS;                  // 16 consecutive outputs per thread
        const int g0 = tid * EO < n ? tid * EO : n;              // (threads past the end merge nothing)
        int lo = g0 - nB > 0 ? g0 - nB : 0, hi = g0 < CAP ? g0 : CAP;
        while (lo < hi) {                                         // the merge path crosses diagonal g0 at (lo, g0 - lo)
            const int mid = (lo + hi) >> 1;
            if (M[SR_MP(mid)] < M[SR_MP(CAP + g0 - 1 - mid)]) lo = mid + 1; else hi = mid;
        }
        int ia = lo, ib = g0 - lo;
        u64 va = ia < CAP ? M[SR_MP(ia)] : ~0ull, vb = ib < nB ? M[SR_MP(CAP + ib)] : ~0ull;
        float4 r[EO];
#pragma unroll
        for (int k = 0; k < EO; k++) {
            const bool ta = va < vb;
            const u64 w = ta ? va : vb;
            ia += ta ? 1 : 0; ib += ta ? 0 : 1;
            const bool ok = ta ? ia < CAP : ib < nB;
            const u64 nw = ok ? M[SR_MP(ta ? ia : CAP + ib)] : ~0ull;
            va = ta ? nw : va; vb = ta ? vb : nw;
            r[k] = vrec[o0 + (w == ~0ull ? 0u : (u32)w)];        // the gather is issued as soon as its place is known
        }
#pragma unroll
        for (int k = 0; k < EO; k++) asm volatile("" : "+v"(r[k].x), "+v"(r[k].y), "+v"(r[k].z), "+v"(r[k].w));
        // the row in front of this thread's first one: the last row of the lane below (of the wave below for lane 0)
        float4 below;
        below.x = __shfl_up(r[EO - 1].x, 1, 64); below.y = __shfl_up(r[EO - 1].y, 1, 64); below.z = __shfl_up(r[EO - 1].z, 1, 64);
        if (lane == 63) s_tail[wave] = r[EO - 1];
        __syncthreads();
        if (lane == 0 && wave > 0) below = s_tail[wave - 1];
        u32 nviol = 0, ntop = 0;
        float4 prev = below;
#pragma unroll
        for (int k = 0; k < EO; k++) {
            const int i = tid * EO + k;
            if (i < n && (u64)(o0 + i) < nv) {
                if (i > 0 && !rec_less(prev, r[k])) nviol++;
                typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
                *(f3u *)(uniq + 3 * (int64_t)(o0 + i)) = (f3u){r[k].x, r[k].y, r[k].z};
                table[__float_as_uint(r[k].w)] = (int32_t)(o0 + i);
                if (r[k].x == z_top) ntop++;
            }
            prev = r[k];
        }
        nviol = wave_sum(nviol); ntop = wave_sum(ntop);
        if (lane == 0 && nviol) atomicAdd((unsigned long long *)&tot[4], (unsigned long long)nviol);
        if (lane == 0 && ntop) atomicAdd((unsigned long long *)&tot[7], (unsigned long long)ntop);
        return;
    }
    const u64 *sorted;
    const u32 *first = nullptr;                                   // (clamped run) position after the first pass
    if (!uni) {
        sorted = sr_sort(s_a, s_b, n, [=](int i) { return ((u64)keys[o0 + i] << 32) | (u32)i; });
    } else {
        // stable by x', then stable by y': the (y', x') order of rows that share their z'.  Both passes in the two halves of
        // s_a (n <= UCAP), the first order kept in s_b
        u64 *const h0 = s_a, *const h1 = s_a + SR_WORDS(UCAP);
        const u64 *p1 = sr_sort(h0, h1, n, [=](int i) { return ((u64)fkey32(vrec[o0 + i].z) << 32) | (u32)i; });
        u32 *ord = (u32 *)s_b;
        for (int i = tid; i < n; i += SR_THREADS) ord[i] = (u32)p1[SR_PAD(i)];
        __syncthreads();
        sorted = sr_sort(h0, h1, n, [=](int i) { return ((u64)fkey32(vrec[o0 + ord[i]].y) << 32) | (u32)i; });
        first = ord;
    }
    // ---- the rows in order: every gather of a thread is in flight at once, the rows pass through LDS (the predecessor of a
    //      row is its neighbour there), then rows, table and the strict-ascending check inside the segment
    u32 p[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) {                                // (no store under a condition: the arrays stay in registers)
        const int i = tid + SR_THREADS * e, ii = i < n ? i : n - 1;
        const u32 q = (u32)sorted[SR_PAD(ii)];
        p[e] = first ? first[q] : q;
    }
    __syncthreads();
    float4 r[EP];
#pragma unroll
    for (int e = 0; e < EP; e++) r[e] = vrec[o0 + p[e]];
#pragma unroll
    for (int e = 0; e < EP; e++)                                  // every gather is issued before the first one is waited for (the
        asm volatile("" : "+v"(r[e].x), "+v"(r[e].y), "+v"(r[e].z), "+v"(r[e].w));   // compiler sinks each load to its store otherwise)
#pragma unroll
    for (int e = 0; e < EP; e++) if (tid + SR_THREADS * e < n) s_rows[tid + SR_THREADS * e] = r[e];
    __syncthreads();
    u32 nviol = 0, ntop = 0;
#pragma unroll
    for (int e = 0; e < EP; e++) {
        const int i = tid + SR_THREADS * e;
        if (i < n && (u64)(o0 + i) < nv) {
            const float4 a = s_rows[i];
            if (i > 0 && !rec_less(s_rows[i - 1], a)) nviol++;
            typedef float f3u __attribute__((ext_vector_type(3), aligned(4)));
            *(f3u *)(uniq + 3 * (int64_t)(o0 + i)) = (f3u){a.x, a.y, a.z};
            table[__float_as_uint(a.w)] = (int32_t)(o0 + i);
            if (a.x == z_top) ntop++;
        }
    }
    nviol = wave_sum(nviol); ntop = wave_sum(ntop);
    if (lane == 0 && nviol) atomicAdd((unsigned long long *)&tot[4], (unsigned long long)nviol);
    if (lane == 0 && ntop) atomicAdd((unsigned long long *)&tot[7], (unsigned long long)ntop);
}

// the seams between the segments: row offsets[s] - 1 must sort in front of row offsets[s] (one thread per segment)
__global__ __launch_bounds__(256) void uq3_seams_kernel(const float *__restrict__ uniq, const u32 *__restrict__ offsets, const int nseg,
                                                        u64 *__restrict__ tot)
{
    const int s = 1 + blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (s < nseg && tot[3] == 0) {
        const u32 q0 = offsets[s];
        if (q0 != 0 && (u64)q0 < tot[1] && offsets[s + 1] != q0) {            // nothing in front / an empty segment: its successor checks
            const float *r = uniq + 3 * (int64_t)(q0 - 1);
            bad = !(r[0] < r[3] || (r[0] == r[3] && (r[1] < r[4] || (r[1] == r[4] && r[2] < r[5]))));
        }
    }
    const u64 nbad = (u64)__popcll(__ballot(bad));
    if ((threadIdx.x & 63) == 0 && nbad) atomicAdd((unsigned long long *)&tot[4], (unsigned long long)nbad);
}

// keys: the 32-bit sort keys mc3_vertices wrote (its idx array is not needed: the sort carries positions); slice_tab: as for
// tomo_mc3_sort_rank_top.  No workspace.  Segments too long for the kernel's LDS are reported in tot[3] (bit 8).
TOMO_API int tomo_mc3_sort_rank_fused(const float *vrec, const uint32_t *keys, int64_t cap_v, int Nz, int Ny, uint32_t *slice_tab,
                                      unsigned long long *tot, float *uniq, int32_t *table, float z_top, void *stream)
{
    if (!vrec || !keys || !slice_tab || !tot || !uniq || !table || cap_v <= 0 || Nz < 1 || Ny < 1) return TOMO_E_ARG;
    if (cap_v >= 0x7fffffffll || Nz > UQ_MAX_SLABS) return TOMO_E_SIZE;
    const int NB = TOMO_SORT_NB(Ny);
    const int64_t nseg = (int64_t)(NB + 1) * Nz;
    if (nseg >= 0x7fffffffll) return TOMO_E_SIZE;
    const u32 *offsets = slice_tab + 2 * ((int64_t)Nz + 1);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 2 * SR_WORDS(SR_HALF) * sizeof(u64);
    hipLaunchKernelGGL(uq3_sortrank_kernel, dim3((unsigned)nseg), dim3(SR_THREADS), lds, s, (const float4 *)vrec, (const u32 *)keys,
                       offsets, NB, Nz, uniq, table, (u64 *)tot, z_top);
    hipLaunchKernelGGL(uq3_seams_kernel, dim3((unsigned)ceil_div64(nseg, 256)), dim3(256), 0, s, (const float *)uniq, offsets, (int)nseg,
                       (u64 *)tot);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ lookup
// Index of every query row in a lexicographically sorted, duplicate-free (U,3) row list (binary search); a row that is
// not there gets -1 and is counted in *missing.  The Z-slab job uses it to number the few thousand shared-plane vertices
// a rank receives from below against its own unique list instead of sorting them in.
__global__ __launch_bounds__(256) void rows_lookup_kernel(const float *__restrict__ uniq, int64_t nu,
                                                          const float *__restrict__ query, int64_t nq,
                                                          int32_t *__restrict__ out, u64 *__restrict__ missing)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    const float qz = query[3 * i], qy = query[3 * i + 1], qx = query[3 * i + 2];
    int64_t lo = 0, hi = nu;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const float *m = uniq + 3 * mid;
        const bool less = m[0] < qz || (m[0] == qz && (m[1] < qy || (m[1] == qy && m[2] < qx)));
        if (less) lo = mid + 1; else hi = mid;
    }
    const bool found = lo < nu && uniq[3 * lo] == qz && uniq[3 * lo + 1] == qy && uniq[3 * lo + 2] == qx;
    out[i] = found ? (int32_t)lo : -1;
    if (!found) atomicAdd(missing, 1ull);
}

TOMO_API int tomo_mesh_lookup(const float *uniq, int64_t nu, const float *query, int64_t nq, int32_t *out,
                              unsigned long long *missing, void *stream)
{
    if (nq < 0 || nu < 0 || (nq > 0 && (!query || !out || !missing)) || (nu > 0 && !uniq)) return TOMO_E_ARG;
    if (nq == 0) return TOMO_OK;
    if (nu >= 0x7fffffffll) return TOMO_E_SIZE;
    hipLaunchKernelGGL(rows_lookup_kernel, dim3((unsigned)ceil_div64(nq, 256)), dim3(256), 0, (hipStream_t)stream, uniq, nu,
                       query, nq, out, (u64 *)missing);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ Z-slab numbering on the device
// A Z-slab rank (slab.py) numbers its vertices against its neighbours': the rows on the plane it shares with the rank above
// close its sorted list (tot[7] of them, counted by uq3_rank_kernel), go up, are looked up there, and their indices come
// back.  With the message capacities known from the last pass none of the counts has to reach the host before the triangles
// are written: these three kernels (and mc3_faces_kernel<true>, which writes the GLOBAL indices) take every count from device memory.
//   message up   float32 (cap + 1, 3): row 0 = {number of rows as uint32 bits, 0, 0}, then the rows, zero padded
//   summary      int64[8] per rank, all-gathered: kept rows | rows from below not found | flags | nv | n_top |
//                rows announced from below | list length | triangles
//                flags: 1 a chain buffer overflowed, 2 rows do not ascend strictly, 4 n_top > cap, 8 set by the caller
__device__ static inline bool slab_counts_ok(const u64 *tot, int64_t cap_v)
{
    return tot[3] == 0 && tot[1] <= (u64)cap_v && tot[7] <= tot[1];
}

__global__ __launch_bounds__(256) void slab_top_rows_kernel(const float *__restrict__ uniq, const u64 *__restrict__ tot,
                                                            int64_t cap_v, int64_t cap, float *__restrict__ msg)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = slab_counts_ok(tot, cap_v);
    const u64 nv = ok ? tot[1] : 0, nt = ok ? tot[7] : 0;
    if (i == 0) {
        msg[0] = __uint_as_float((u32)(nt < 0xffffffffull ? nt : 0xffffffffull));
        msg[1] = msg[2] = 0.f;
    }
    if (i >= cap) return;
    float z = 0.f, y = 0.f, x = 0.f;
    if ((u64)i < nt) {
        const float *src = uniq + 3 * (nv - nt + (u64)i);
        z = src[0]; y = src[1]; x = src[2];
    }
    msg[3 + 3 * i] = z; msg[4 + 3 * i] = y; msg[5 + 3 * i] = x;
}

__global__ __launch_bounds__(256) void slab_lookup_kernel(const float *__restrict__ uniq, const u64 *__restrict__ tot,
                                                          int64_t cap_v, const float *__restrict__ msg, int64_t cap,
                                                          int32_t *__restrict__ out, u64 *__restrict__ missing)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const int64_t nu = slab_counts_ok(tot, cap_v) ? (int64_t)tot[1] : 0;
    const int64_t nq = (int64_t)__float_as_uint(msg[0]);
    if (i >= nq) { out[i] = -1; return; }
    const float qz = msg[3 + 3 * i], qy = msg[4 + 3 * i], qx = msg[5 + 3 * i];
    int64_t lo = 0, hi = nu;
    while (lo < hi) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        const float *m = uniq + 3 * mid;
        const bool less = m[0] < qz || (m[0] == qz && (m[1] < qy || (m[1] == qy && m[2] < qx)));
        if (less) lo = mid + 1; else hi = mid;
    }
    const bool found = lo < nu && uniq[3 * lo] == qz && uniq[3 * lo + 1] == qy && uniq[3 * lo + 2] == qx;
    out[i] = found ? (int32_t)lo : -1;
    if (!found) atomicAdd((unsigned long long *)missing, 1ull);
}

__global__ void slab_summary_kernel(const u64 *__restrict__ tot, int64_t cap_v, const float *__restrict__ msg_in,
                                    u64 *__restrict__ missing, int64_t cap_top, int64_t caller_flags,
                                    int64_t *__restrict__ out)
{
    if (blockIdx.x | threadIdx.x) return;
    int64_t flags = caller_flags ? 8 : 0;
    if (tot[3] != 0 || tot[1] > (u64)cap_v || tot[7] > tot[1]) flags |= 1;
    if (tot[4] != 0) flags |= 2;
    if (tot[7] > (u64)cap_top) flags |= 4;
    const int64_t from_prev = msg_in ? (int64_t)__float_as_uint(msg_in[0]) : 0;
    out[0] = (flags & 1) ? 0 : (int64_t)(tot[1] - tot[7]);
    out[1] = missing ? (int64_t)*missing : 0;
    if (missing) *missing = 0ull;               // read once per pass: cleared here for the next lookup (no memset per pass)
    out[2] = flags;
    out[3] = (int64_t)tot[1];
    out[4] = (int64_t)tot[7];
    out[5] = from_prev;
    out[6] = (int64_t)tot[0];
    out[7] = (int64_t)tot[2];
}

// tomo_slab_lookup + tomo_slab_summary in ONE launch: every workgroup looks its share of the received rows up (a binary search
// each: ~22 dependent loads, so the few thousand rows of a shared plane are spread over many workgroups -- one workgroup alone
// took 36 us), adds its misses to scratch[1] and takes a ticket from scratch[0]; the workgroup that draws the last ticket sees
// every miss (fence + atomic), writes the summary and leaves both words zero for the next pass.  scratch: uint64[2], zeroed
// ONCE by the caller.  msg == null (the lowest rank): the summary alone.
__global__ __launch_bounds__(256) void slab_lookup_summary_kernel(const float *__restrict__ uniq, const u64 *__restrict__ tot,
                                                                  int64_t cap_v, const float *__restrict__ msg, int64_t cap,
                                                                  int32_t *__restrict__ out, int64_t cap_top, int64_t caller_flags,
                                                                  u64 *__restrict__ scratch, int64_t *__restrict__ summary)
{
    __shared__ int s_last;
    bool missed = false;
    if (msg != nullptr) {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const int64_t nu = slab_counts_ok(tot, cap_v) ? (int64_t)tot[1] : 0;
        const int64_t nq = (int64_t)__float_as_uint(msg[0]);
        if (i < cap) {
            if (i >= nq) out[i] = -1;
            else {
                const float qz = msg[3 + 3 * i], qy = msg[4 + 3 * i], qx = msg[5 + 3 * i];
                int64_t lo = 0, hi = nu;
                while (lo < hi) {
                    const int64_t mid = lo + ((hi - lo) >> 1);
                    const float *m = uniq + 3 * mid;
                    const bool less = m[0] < qz || (m[0] == qz && (m[1] < qy || (m[1] == qy && m[2] < qx)));
                    if (less) lo = mid + 1; else hi = mid;
                }
                const bool found = lo < nu && uniq[3 * lo] == qz && uniq[3 * lo + 1] == qy && uniq[3 * lo + 2] == qx;
                out[i] = found ? (int32_t)lo : -1;
                missed = !found;
            }
        }
    }
    const u64 nm = (u64)__popcll(__ballot(missed));
    if ((threadIdx.x & 63) == 0 && nm) atomicAdd((unsigned long long *)&scratch[1], (unsigned long long)nm);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd((unsigned long long *)&scratch[0], 1ull) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!s_last || threadIdx.x != 0) return;
    __threadfence();
    const u64 misses = atomicExch((unsigned long long *)&scratch[1], 0ull);
    scratch[0] = 0ull;
    int64_t flags = caller_flags ? 8 : 0;
    if (tot[3] != 0 || tot[1] > (u64)cap_v || tot[7] > tot[1]) flags |= 1;
    if (tot[4] != 0) flags |= 2;
    if (tot[7] > (u64)cap_top) flags |= 4;
    summary[0] = (flags & 1) ? 0 : (int64_t)(tot[1] - tot[7]);
    summary[1] = (int64_t)misses;
    summary[2] = flags;
    summary[3] = (int64_t)tot[1];
    summary[4] = (int64_t)tot[7];
    summary[5] = msg ? (int64_t)__float_as_uint(msg[0]) : 0;
    summary[6] = (int64_t)tot[0];
    summary[7] = (int64_t)tot[2];
}

TOMO_API int tomo_slab_lookup_summary(const float *uniq, const unsigned long long *tot, int64_t cap_v, const float *msg, int64_t cap,
                                      int32_t *out, int64_t cap_top, int64_t caller_flags, unsigned long long *scratch, int64_t *summary,
                                      void *stream)
{
    if (!uniq || !tot || !summary || !scratch || cap_v < 1 || cap_top < 0 || (msg && (!out || cap < 1))) return TOMO_E_ARG;
    const int64_t blocks = msg ? ceil_div64(cap, 256) : 1;
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(slab_lookup_summary_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, uniq, (const u64 *)tot,
                       cap_v, msg, cap, out, cap_top, caller_flags, (u64 *)scratch, summary);
    return tomo_status();
}

TOMO_API int tomo_slab_top_rows(const float *uniq, const unsigned long long *tot, int64_t cap_v, int64_t cap, float *msg, void *stream)
{
    if (!uniq || !tot || !msg || cap < 1 || cap_v < 1) return TOMO_E_ARG;
    hipLaunchKernelGGL(slab_top_rows_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, (hipStream_t)stream, uniq,
                       (const u64 *)tot, cap_v, cap, msg);
    return tomo_status();
}

TOMO_API int tomo_slab_lookup(const float *uniq, const unsigned long long *tot, int64_t cap_v, const float *msg, int64_t cap,
                              int32_t *out, unsigned long long *missing, void *stream)
{
    if (!uniq || !tot || !msg || !out || !missing || cap < 1 || cap_v < 1) return TOMO_E_ARG;
    hipLaunchKernelGGL(slab_lookup_kernel, dim3((unsigned)ceil_div64(cap, 256)), dim3(256), 0, (hipStream_t)stream, uniq,
                       (const u64 *)tot, cap_v, msg, cap, out, (u64 *)missing);
    return tomo_status();
}

TOMO_API int tomo_slab_summary(const unsigned long long *tot, int64_t cap_v, const float *msg_in, unsigned long long *missing,
                               int64_t cap_top, int64_t caller_flags, int64_t *out, void *stream)
{
    if (!tot || !out || cap_v < 1 || cap_top < 0) return TOMO_E_ARG;
    hipLaunchKernelGGL(slab_summary_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const u64 *)tot, cap_v, msg_in,
                       (u64 *)missing, cap_top, caller_flags, out);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ faces
// provisional vertex ids (int32, from mc_emit) -> final ids through `rank`, drop triangles with fewer than
// three distinct indices (order kept), widen to int64 like np.unique's inverse.
__global__ __launch_bounds__(256) void faces_rank_kernel(const int32_t *__restrict__ faces32, int64_t nf,
                                                         const int32_t *__restrict__ rank, int32_t *__restrict__ ids,
                                                         u32 *__restrict__ keep)
{
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    int32_t a = rank[faces32[3 * f + 0]], b = rank[faces32[3 * f + 1]], c = rank[faces32[3 * f + 2]];
    ids[3 * f + 0] = a; ids[3 * f + 1] = b; ids[3 * f + 2] = c;
    keep[f] = (a != b && b != c && a != c) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void faces_compact_kernel(const int32_t *__restrict__ ids, const u32 *__restrict__ keep,
                                                            const u32 *__restrict__ kscan, int64_t nf,
                                                            int64_t *__restrict__ faces_out, u64 *__restrict__ totals)
{
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nf) return;
    if (keep[f]) {
        int64_t o = (int64_t)kscan[f] - 1;
        faces_out[3 * o + 0] = ids[3 * f + 0];
        faces_out[3 * o + 1] = ids[3 * f + 1];
        faces_out[3 * o + 2] = ids[3 * f + 2];
    }
    if (f == nf - 1) totals[1] = (u64)kscan[f];
}

// Speculative single pass: final ids straight to int64 at the face's own position, counting the degenerate faces.  If the
// count is 0 -- no two vertices of a triangle merged, the usual case -- this IS the result (totals[1] = nf) and the
// rank / scan / compact passes of tomo_mesh_faces are not needed; otherwise the caller runs tomo_mesh_faces.
__global__ __launch_bounds__(256) void faces_direct_kernel(const int32_t *__restrict__ faces32, int64_t nf,
                                                           const int32_t *__restrict__ rank, int64_t *__restrict__ faces_out,
                                                           u64 *__restrict__ totals)
{
    int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool bad = false;
    if (f < nf) {
        const int32_t a = rank[faces32[3 * f + 0]], b = rank[faces32[3 * f + 1]], c = rank[faces32[3 * f + 2]];
        faces_out[3 * f + 0] = a; faces_out[3 * f + 1] = b; faces_out[3 * f + 2] = c;
        bad = (a == b || b == c || a == c);
    }
    const u64 nbad = (u64)__popcll(__ballot(bad));
    if ((threadIdx.x & 63) == 0 && nbad) atomicAdd(&totals[3], (unsigned long long)nbad);
    if (f == 0) totals[1] = (u64)nf;
}

TOMO_API int tomo_mesh_faces_direct(const int32_t *faces32, int64_t nf, const int32_t *rank, int64_t *faces_out,
                                    unsigned long long *totals, void *stream)
{
    if (!faces32 || !rank || !faces_out || !totals || nf <= 0) return TOMO_E_ARG;
    if (nf >= 0x7fffffffll) return TOMO_E_SIZE;
    hipLaunchKernelGGL(faces_direct_kernel, dim3((unsigned)ceil_div64(nf, 256)), dim3(256), 0, (hipStream_t)stream, faces32, nf,
                       rank, faces_out, (u64 *)totals);
    return tomo_status();
}

struct FcLayout { size_t ids, keep, kscan, temp, temp_bytes, total; };

static FcLayout fc_layout(int64_t nf)
{
    FcLayout L;
    size_t n = (size_t)(nf > 0 ? nf : 1), off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.ids = take(n * 12); L.keep = take(n * 4); L.kscan = take(n * 4);
    size_t t = 0;
    (void)rocprim::inclusive_scan(nullptr, t, (u32 *)nullptr, (u32 *)nullptr, n, rocprim::plus<u32>(), (hipStream_t)0);
    L.temp_bytes = t;
    L.temp = take(t + 256);
    L.total = off;
    return L;
}

TOMO_API int64_t tomo_mesh_faces_workspace_bytes(int64_t nf) { return (int64_t)fc_layout(nf).total; }

TOMO_API int tomo_mesh_faces(const int32_t *faces32, int64_t nf, const int32_t *rank, int64_t *faces_out,
                             unsigned long long *totals, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!faces32 || !rank || !faces_out || !totals || !workspace || nf <= 0) return TOMO_E_ARG;
    if (nf >= 0x7fffffffll) return TOMO_E_SIZE;
    FcLayout L = fc_layout(nf);
    if ((size_t)workspace_bytes < L.total) return TOMO_E_WORKSPACE;
    char *ws = (char *)workspace;
    int32_t *ids = (int32_t *)(ws + L.ids);
    u32 *keep = (u32 *)(ws + L.keep), *kscan = (u32 *)(ws + L.kscan);
    hipStream_t s = (hipStream_t)stream;
    unsigned blocks = (unsigned)ceil_div64(nf, 256);
    hipLaunchKernelGGL(faces_rank_kernel, dim3(blocks), dim3(256), 0, s, faces32, nf, rank, ids, keep);
    size_t tb = L.temp_bytes;
    if (rocprim::inclusive_scan(ws + L.temp, tb, keep, kscan, (size_t)nf, rocprim::plus<u32>(), s) != hipSuccess)
        return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(faces_compact_kernel, dim3(blocks), dim3(256), 0, s, (const int32_t *)ids, (const u32 *)keep,
                       (const u32 *)kscan, nf, faces_out, (u64 *)totals);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ S10/S11
__global__ __launch_bounds__(256) void volume_area_kernel(const float *__restrict__ v, const int64_t *__restrict__ faces,
                                                          int64_t nf, double *__restrict__ out)
{
    double vol = 0.0, area = 0.0;
    for (int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += (int64_t)gridDim.x * blockDim.x) {
        const float *a = v + 3 * faces[3 * f], *b = v + 3 * faces[3 * f + 1], *c = v + 3 * faces[3 * f + 2];
        // dot(v0, cross(v1, v2)) in float32, then /6.0 in float64 (NumPy 1.x scalar promotion)
        float cx = b[1] * c[2] - b[2] * c[1], cy = b[2] * c[0] - b[0] * c[2], cz = b[0] * c[1] - b[1] * c[0];
        float d = a[0] * cx + a[1] * cy + a[2] * cz;
        vol += (double)d / 6.0;
        float ux = b[0] - a[0], uy = b[1] - a[1], uz = b[2] - a[2];
        float wx = c[0] - a[0], wy = c[1] - a[1], wz = c[2] - a[2];
        float nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
        float nrm = sqrtf(nx * nx + ny * ny + nz * nz);
        area += (double)(0.5f * nrm);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { vol += __shfl_xor(vol, d, 64); area += __shfl_xor(area, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], vol); atomicAdd(&out[1], area); }
}

TOMO_API int tomo_mesh_volume_area(const float *verts, const int64_t *faces, int64_t nf, double *out, void *stream)
{
    if (!verts || !faces || !out || nf < 0) return TOMO_E_ARG;
    if (nf == 0) return TOMO_OK;
    int64_t blocks = ceil_div64(nf, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(volume_area_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, verts, faces, nf, out);
    return tomo_status();
}
