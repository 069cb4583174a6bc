// mc.hip -- Lewiner marching cubes on gfx950: classify -> compact -> evaluate -> scan -> emit.
//
// Replaces skimage.measure.marching_cubes(volume, level=0.5) as called at surface_extractor.py:55
// (Lewiner MC33, step 1, allow_degenerate=True).  The serial original walks cells z->y->x and numbers
// vertices by first touch through two "face layers".  Data-parallel formulation:
//   * a vertex is identified by the edge it sits on: owner voxel (the edge's lower corner) + slot
//     (0/1/2 = x/y/z edge, 3 = cell-centre vertex).  An edge vertex exists iff the field changes side of
//     the iso level along the edge (every MC33 tiling uses exactly the bichromatic edges of its cube);
//     a centre vertex exists iff the cell's tiling row contains a 12;
//   * pass 1 never touches the float field: the field kernel leaves one SIGN BIT per voxel behind (32-byte
//     records per row and 256-column segment, stored only for 16-row groups that are not constant -- a class byte
//     per group says which), and the ACTIVE voxels (8 cube corners not all on one side) of a segment follow from
//     the records of four rows with a few 64-bit operations per segment; fields that did not come from the field
//     kernel get their sign records from field_signs_kernel;
//   * scan (rocPRIM, single pass) -> offsets; pass 2 writes the compact, ordered list of active voxels;
//   * pass 3 evaluates MC33 once per active voxel, ONE VOXEL PER LANE (full lane utilisation for the
//     branchy code), giving triangle and vertex counts; scan -> output offsets;
//   * pass 4 writes vertices and triangles at their final positions: triangle order == the reference's
//     cell scan order, LUT order inside a cell; triangle corners are resolved to vertex indices through ONE list
//     lookup per owner voxel: position = segment offset + rank of the voxel's bit in the segment's ballots.
// No atomics decide any position: output is deterministic.
#include <cstring>
#include <cstdlib>
#include "tomo_common.h"
#include <rocprim/rocprim.hpp>

#define MC_LUT_QUAL __device__ const
#define MC_FN __device__ static inline
#include "mc_cell.h"

#define SEG 256
#define KEY_XBITS 20
static_assert(KEY_XBITS + 2 == TOMO_KEY_ROW_SHIFT, "key layout shared with mesh.hip");

struct McGrid {
    int Nz, Ny, Nx;
    int64_t pitch;
    int xorg;
    int segs_per_row;
    double iso;
};

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float float3u __attribute__((ext_vector_type(3), aligned(4)));      // one 12-byte store per vertex / triangle
typedef int int3u __attribute__((ext_vector_type(3), aligned(4)));
typedef u32 u32x2u __attribute__((ext_vector_type(2), aligned(4)));

// q = a / b, returns a % b, for 0 <= a and 0 < b < 2^31: 32-bit division whenever a fits (rows, segments and tasks of every
// volume the path meets do; a 64-bit division costs the GPU ~100 VALU instructions, and these sit in per-lane prologues)
__device__ static inline int divmod_pos(int64_t a, int b, int64_t *q)
{
    if ((u64)a <= 0xffffffffull) {
        const u32 qa = (u32)a / (u32)b;
        *q = (int64_t)qa;
        return (int)((u32)a - qa * (u32)b);
    }
    *q = a / b;
    return (int)(a - *q * b);
}

__device__ static inline u64 make_key(int64_t row, int X, int slot)
{
    return ((u64)row << (KEY_XBITS + 2)) | ((u64)(u32)X << 2) | (u64)slot;
}

// ------------------------------------------------------------------------------------------ pass 1
// Sign records: for every (slice Z, segment s, row Y) four 64-bit words, bit L of word k = [field > iso] at
// column 256 s - 224 + 4 L + k, laid out [Z][s][Y][4] (a wave's consecutive rows are consecutive records and a
// Z-slab is a contiguous view).  The field ("SDF") kernel writes them as a by-product (field.hip); this generic
// kernel derives them from any float field (dense test volumes, the halo slice of a Z-slab, very wide rows).
__device__ static inline int64_t tomo_sign_rows_dev(int Ny) { return ((int64_t)Ny + 15) / 16 * 16; }

__global__ __launch_bounds__(256) void field_signs_kernel(const float *__restrict__ field, const McGrid g, int z_begin,
                                                          int64_t ntasks, u64 *__restrict__ signs)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (Z - z_begin, s, Y), Y fastest
    if (task >= ntasks) return;
    int64_t tz, zq;
    const int Y = divmod_pos(task, g.Ny, &tz);
    const int s = divmod_pos(tz, g.segs_per_row, &zq);
    const int Z = z_begin + (int)zq;
    const int X0 = s * SEG + lane * 4 - SEG_SHIFT - g.xorg;
    const int64_t NyP = tomo_sign_rows_dev(g.Ny);
    const float *row = field + ((int64_t)Z * g.Ny + Y) * g.pitch + g.xorg;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (X0 + k >= 0 && X0 + k < g.Nx) ? row[X0 + k] : 0.0f;
    u64 b0 = __ballot((double)v[0] > g.iso), b1 = __ballot((double)v[1] > g.iso);
    u64 b2 = __ballot((double)v[2] > g.iso), b3 = __ballot((double)v[3] > g.iso);
    if (lane < 4)
        signs[((((int64_t)Z * g.segs_per_row + s) * NyP) + Y) * 4 + lane] = lane == 0 ? b0 : (lane == 1 ? b1 : (lane == 2 ? b2 : b3));
}

// One LANE per (Z, s, Y): the active-voxel ballots of the segment from the sign records of the four rows
// (Z,Y) (Z,Y+1) (Z+1,Y) (Z+1,Y+1) (neighbours clamped at the volume border) -- 64-bit logic only.
// Every segment gets its number of active voxels in seg_cnt (what the scan reads: 4 B instead of 32 B per segment);
// the four ballots of a NON-EMPTY segment go, as one aligned 32-byte record, into seg_act (other records stay unwritten
// and are never read).
struct Rec4 { u64 b[4]; };

__device__ static inline Rec4 load_rec(const u64 *__restrict__ signs, int64_t idx)
{
    const ulonglong2 *q = (const ulonglong2 *)(signs + idx * 4);
    ulonglong2 lo = q[0], hi = q[1];
    Rec4 r; r.b[0] = lo.x; r.b[1] = lo.y; r.b[2] = hi.x; r.b[3] = hi.y;
    return r;
}

__device__ static inline Rec4 rec_of(const u64 *__restrict__ signs, int64_t idx, int gc)
{   // record idx of a group of class gc: all zero / all one without touching memory, or the stored record
    if (gc == 2) return load_rec(signs, idx);
    Rec4 r;
    r.b[0] = r.b[1] = r.b[2] = r.b[3] = gc ? ~0ull : 0ull;
    return r;
}

__global__ __launch_bounds__(256) void mc_classify_bits_kernel(const u64 *__restrict__ signs,
                                                               const unsigned char *__restrict__ gcls, const McGrid g,
                                                               int64_t ntasks, u64 *__restrict__ seg_act,
                                                               u32 *__restrict__ seg_cnt)
{
    const int64_t task = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (Z, s, Y), Y fastest
    if (task >= ntasks) return;
    int64_t tz, zq;
    const int Y = divmod_pos(task, g.Ny, &tz);
    const int s = divmod_pos(tz, g.segs_per_row, &zq);
    const int Z = (int)zq;
    const int S = g.segs_per_row;
    const int Yn = Y + 1 < g.Ny ? Y + 1 : g.Ny - 1, Z1 = Z + 1 < g.Nz ? Z + 1 : g.Nz - 1;
    const int Xbase = s * SEG - SEG_SHIFT - g.xorg;      // X of (lane 0, k 0)
    const int64_t segi = ((int64_t)Z * g.Ny + Y) * S + s;
    if (Xbase >= g.Nx || Xbase + SEG <= 0) { seg_cnt[segi] = 0u; return; }   // no voxel of the row in this segment
    const bool has_next = s + 1 < S;
    const int64_t NyP = tomo_sign_rows_dev(g.Ny);
    const int64_t ia = ((int64_t)Z * S + s) * NyP + Y, ib = ((int64_t)Z * S + s) * NyP + Yn;
    const int64_t ic = ((int64_t)Z1 * S + s) * NyP + Y, id = ((int64_t)Z1 * S + s) * NyP + Yn;
    // classes of the four 16-row groups of records involved (and of their right neighbours): 0 / 1 = constant, no records
    const int64_t G = NyP >> 4;
    const int64_t qa = ((int64_t)Z * G + (Y >> 4)) * S + s, qb = ((int64_t)Z * G + (Yn >> 4)) * S + s;
    const int64_t qc = ((int64_t)Z1 * G + (Y >> 4)) * S + s, qd = ((int64_t)Z1 * G + (Yn >> 4)) * S + s;
    const int ga = gcls[qa], gb = gcls[qb], gc = gcls[qc], gd = gcls[qd];
    const int na = has_next ? gcls[qa + 1] : ga, nb = has_next ? gcls[qb + 1] : gb;
    const int nc = has_next ? gcls[qc + 1] : gc, nd = has_next ? gcls[qd + 1] : gd;
    if (ga != 2 && ga == gb && ga == gc && ga == gd && na == ga && nb == ga && nc == ga && nd == ga) {
        seg_cnt[segi] = 0u;                              // the whole neighbourhood has one sign
        return;
    }
    Rec4 R[4] = {rec_of(signs, ia, ga), rec_of(signs, ib, gb), rec_of(signs, ic, gc), rec_of(signs, id, gd)};
    u64 nx0[4];                                          // first column of the next segment, same rows
    nx0[0] = has_next ? (na == 2 ? signs[(ia + NyP) * 4] & 1ull : (u64)na) : 0ull;
    nx0[1] = has_next ? (nb == 2 ? signs[(ib + NyP) * 4] & 1ull : (u64)nb) : 0ull;
    nx0[2] = has_next ? (nc == 2 ? signs[(ic + NyP) * 4] & 1ull : (u64)nc) : 0ull;
    nx0[3] = has_next ? (nd == 2 ? signs[(id + NyP) * 4] & 1ull : (u64)nd) : 0ull;
    // per element k: which lanes hold an existing voxel, and which lane holds the last voxel X = Nx-1 (its
    // x+1 neighbour is clamped to itself)
    u64 act[4];
    u64 any_nonzero = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // lanes L with 0 <= Xbase + 4L + k < Nx
        int lo = Xbase + k < 0 ? (-(Xbase + k) + 3) / 4 : 0;
        int hi = (g.Nx - 1 - Xbase - k) >= 0 ? (g.Nx - 1 - Xbase - k) / 4 : -1;      // last valid lane
        if (hi > 63) hi = 63;
        u64 vmask = 0;
        if (hi >= lo) vmask = (hi - lo == 63) ? ~0ull : (((1ull << (hi - lo + 1)) - 1ull) << lo);
        int dl = g.Nx - 1 - Xbase - k;                                               // X == Nx-1 <=> 4L == dl
        u64 lastm = (dl >= 0 && (dl & 3) == 0 && (dl >> 2) < 64) ? (1ull << (dl >> 2)) : 0ull;
        u64 any = 0, all = ~0ull;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            u64 cur = R[r].b[k];
            u64 nxt = k < 3 ? R[r].b[k + 1] : ((R[r].b[0] >> 1) | (nx0[r] << 63));
            nxt = (nxt & ~lastm) | (cur & lastm);
            any |= cur | nxt;
            all &= cur & nxt;
        }
        act[k] = any & ~all & vmask;
        any_nonzero |= act[k];
    }
    seg_cnt[segi] = any_nonzero ? (u32)(__popcll(act[0]) + __popcll(act[1]) + __popcll(act[2]) + __popcll(act[3])) : 0u;
    if (any_nonzero) {
        ulonglong2 *o = (ulonglong2 *)(seg_act + segi * 4);
        o[0] = make_ulonglong2(act[0], act[1]);
        o[1] = make_ulonglong2(act[2], act[3]);
    }
}

static inline int make_grid(McGrid &g, const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso)
{
    if (!field || Nz < 2 || Ny < 2 || Nx < 2 || pitch < Nx + xorg) return TOMO_E_ARG;
    if (Nx >= (1 << KEY_XBITS)) return TOMO_E_SIZE;
    g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = pitch; g.xorg = xorg; g.iso = iso;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    return TOMO_OK;
}

TOMO_API int tomo_field_signs(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                              int z_begin, int z_end, unsigned long long *signs, uint8_t *gcls, void *stream)
{
    McGrid g;
    int rc = make_grid(g, field, Nz, Ny, Nx, pitch, xorg, iso);
    if (rc) return rc;
    if (!signs || !gcls || z_begin < 0 || z_end > Nz || z_begin > z_end) return TOMO_E_ARG;
    {   // every group of these slices has stored records
        const int64_t per_slice = (tomo_sign_rows(Ny) >> 4) * g.segs_per_row;
        if (z_end > z_begin && hipMemsetAsync(gcls + z_begin * per_slice, 2, (size_t)((z_end - z_begin) * per_slice),
                                              (hipStream_t)stream) != hipSuccess)
            return TOMO_E_LAUNCH;
    }
    int64_t ntasks = (int64_t)(z_end - z_begin) * g.segs_per_row * Ny;
    if (ntasks == 0) return TOMO_OK;
    int64_t blocks = ceil_div64(ntasks, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(field_signs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, g, z_begin,
                       ntasks, (u64 *)signs);
    return tomo_status();
}

TOMO_API int tomo_mc_classify(const unsigned long long *signs, const uint8_t *gcls, int Nz, int Ny, int Nx, int xorg,
                              unsigned long long *seg_act, uint32_t *seg_cnt, void *stream)
{
    if (!signs || !gcls || !seg_act || !seg_cnt || Nz < 2 || Ny < 2 || Nx < 2) return TOMO_E_ARG;
    if (Nx >= (1 << KEY_XBITS)) return TOMO_E_SIZE;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    int64_t nseg = (int64_t)Nz * Ny * g.segs_per_row;
    int64_t blocks = ceil_div64(nseg, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_classify_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const u64 *)signs, gcls, g, nseg, (u64 *)seg_act, seg_cnt);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ scan
// Exclusive scans of packed counts (low 16 bits: a, high 16 bits: b) with rocPRIM's single-pass (decoupled look-back)
// scan: input through a transform iterator that unpacks a count into an (a, b) pair, output through a zip iterator
// straight into the two offset arrays.  Used over segments (a = active voxels) and over active voxels (a = vertices,
// b = triangles).  off[0] = 0 and the totals are written by a one-thread kernel.
struct ScanPair { u32 a, b; };
struct ScanUnpack {
    __host__ __device__ rocprim::tuple<u32, u32> operator()(u32 c) const { return rocprim::make_tuple(c & 0xffffu, c >> 16); }
};
struct ScanAdd {
    __host__ __device__ rocprim::tuple<u32, u32> operator()(const rocprim::tuple<u32, u32> &x,
                                                            const rocprim::tuple<u32, u32> &y) const
    {
        return rocprim::make_tuple(rocprim::get<0>(x) + rocprim::get<0>(y), rocprim::get<1>(x) + rocprim::get<1>(y));
    }
};

__global__ void scan_finish_kernel(u32 *__restrict__ off_a, u32 *__restrict__ off_b, int64_t n, u64 *__restrict__ totals)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        off_a[0] = 0u;
        if (off_b) off_b[0] = 0u;
        totals[0] = off_a[n];
        totals[1] = off_b ? off_b[n] : 0ull;
        totals[2] = 0ull;
        totals[3] = 0ull;
    }
}

static size_t scan_temp_bytes(int64_t n)
{
    size_t t1 = 0, t2 = 0;
    size_t m = (size_t)(n > 0 ? n : 1);
    (void)rocprim::inclusive_scan(nullptr, t1, (u32 *)nullptr, (u32 *)nullptr, m, rocprim::plus<u32>(), (hipStream_t)0);
    auto in = rocprim::make_transform_iterator((const u32 *)nullptr, ScanUnpack());
    auto out = rocprim::make_zip_iterator(rocprim::make_tuple((u32 *)nullptr, (u32 *)nullptr));
    (void)rocprim::inclusive_scan(nullptr, t2, in, out, m, ScanAdd(), (hipStream_t)0);
    return t1 > t2 ? t1 : t2;
}

TOMO_API int64_t tomo_mc_scan_workspace_bytes(int64_t n) { return (int64_t)scan_temp_bytes(n) + 256; }

// off_a[i] = sum of the low halves of counts[0 .. i), off_b likewise for the high halves (off_b may be NULL: then the
// counts are plain numbers < 2^32 and off_a their exclusive scan); both have n + 1 entries.
static int scan_launch(const u32 *counts, int64_t n, uint32_t *off_a, uint32_t *off_b, unsigned long long *totals,
                       void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!counts || !off_a || !totals || !workspace || n <= 0) return TOMO_E_ARG;
    if (n >= 0xffffffffll) return TOMO_E_SIZE;
    size_t tb = scan_temp_bytes(n);
    if (workspace_bytes < (int64_t)tb) return TOMO_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipError_t rc;
    if (off_b) {
        auto in = rocprim::make_transform_iterator(counts, ScanUnpack());
        auto out = rocprim::make_zip_iterator(rocprim::make_tuple(off_a + 1, off_b + 1));
        rc = rocprim::inclusive_scan(workspace, tb, in, out, (size_t)n, ScanAdd(), s);
    } else {
        rc = rocprim::inclusive_scan(workspace, tb, counts, off_a + 1, (size_t)n, rocprim::plus<u32>(), s);
    }
    if (rc != hipSuccess) return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(scan_finish_kernel, dim3(1), dim3(64), 0, s, off_a, off_b, n, (u64 *)totals);
    return tomo_status();
}

TOMO_API int tomo_mc_scan(const uint32_t *counts, int64_t n, uint32_t *off_a, uint32_t *off_b, uint32_t *nz_ids,
                          unsigned long long *totals, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (nz_ids) return TOMO_E_ARG;            // the list of non-zero entries is no longer produced (ABI 2)
    return scan_launch(counts, n, off_a, off_b, totals, workspace, workspace_bytes, stream);
}

TOMO_API int tomo_mc_scan_segments(const uint32_t *seg_cnt, int64_t nseg, uint32_t *seg_aoff, unsigned long long *totals,
                                   void *workspace, int64_t workspace_bytes, void *stream)
{
    return scan_launch(seg_cnt, nseg, seg_aoff, nullptr, totals, workspace, workspace_bytes, stream);
}

// ------------------------------------------------------------------------------------------ pass 2
// one LANE per segment; the non-empty ones (consecutive lanes write consecutive ranges of vox_key) expand the four
// ballots written by pass 1 into voxel keys, in x order
__global__ __launch_bounds__(256) void mc_list_kernel(const McGrid g, const u32 *__restrict__ seg_aoff,
                                                      const u64 *__restrict__ seg_act, int64_t nseg,
                                                      u64 *__restrict__ vox_key, u32 cap)
{
    int64_t seg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= nseg) return;
    if (seg_aoff[seg + 1] == seg_aoff[seg]) return;             // empty: its ballot record was never written
    if (seg_aoff[seg + 1] > cap) return;                        // (capped call) the list is longer than the buffer: the caller redoes it
    int64_t row;
    int s = divmod_pos(seg, g.segs_per_row, &row);
    const ulonglong2 *q = (const ulonglong2 *)(seg_act + seg * 4);
    ulonglong2 lo = q[0], hi = q[1];
    u64 b0 = lo.x, b1 = lo.y, b2 = hi.x, b3 = hi.y;
    u32 o = seg_aoff[seg];
    const int Xs = s * SEG - SEG_SHIFT - g.xorg;
    u64 any = b0 | b1 | b2 | b3;
    while (any) {                                   // lanes (groups of 4 voxels) that hold an active voxel, ascending
        int L = __ffsll((long long)any) - 1;
        any &= any - 1;
        int X0 = Xs + 4 * L;
        if ((b0 >> L) & 1ull) vox_key[o++] = make_key(row, X0, 0);
        if ((b1 >> L) & 1ull) vox_key[o++] = make_key(row, X0 + 1, 0);
        if ((b2 >> L) & 1ull) vox_key[o++] = make_key(row, X0 + 2, 0);
        if ((b3 >> L) & 1ull) vox_key[o++] = make_key(row, X0 + 3, 0);
    }
}

static int mc_list_launch(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const unsigned long long *seg_act,
                          unsigned long long *vox_key, u32 cap, void *stream);

TOMO_API int tomo_mc_list(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const unsigned long long *seg_act,
                          unsigned long long *vox_key, void *stream)
{
    return mc_list_launch(Nz, Ny, Nx, xorg, seg_aoff, seg_act, vox_key, 0xffffffffu, stream);
}

// The same into a buffer of `cap` keys whose sufficiency the caller has not checked yet (it launches ahead of reading the
// segment scan's total): segments that would not fit are skipped, nothing is written past the buffer.
TOMO_API int tomo_mc_list_capped(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const unsigned long long *seg_act,
                                 unsigned long long *vox_key, int64_t cap, void *stream)
{
    if (cap <= 0 || cap >= 0xffffffffll) return TOMO_E_ARG;
    return mc_list_launch(Nz, Ny, Nx, xorg, seg_aoff, seg_act, vox_key, (u32)cap, stream);
}

static int mc_list_launch(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const unsigned long long *seg_act,
                          unsigned long long *vox_key, u32 cap, void *stream)
{
    if (Nz < 2 || Ny < 2 || Nx < 2 || !seg_aoff || !seg_act || !vox_key) return TOMO_E_ARG;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    const int64_t nseg = (int64_t)Nz * Ny * g.segs_per_row;
    int64_t blocks = ceil_div64(nseg, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_list_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, seg_aoff,
                       (const u64 *)seg_act, nseg, (u64 *)vox_key, cap);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ pass 3 / 4
struct Cell {
    double v[8];       // corner values minus iso, Lewiner order
    int index;
    bool cell_ok;
    int flags;         // bit0/1/2: x/y/z edge vertex owned by this voxel, bit3: centre vertex
    int Z, Y, X;
    int64_t row;
};

__device__ static inline void load_cell(const float *__restrict__ field, const McGrid &g, u64 key, Cell &c)
{
    c.row = (int64_t)(key >> (KEY_XBITS + 2));
    c.X = (int)((key >> 2) & ((1u << KEY_XBITS) - 1u));
    int64_t zq;
    c.Y = divmod_pos(c.row, g.Ny, &zq);
    c.Z = (int)zq;
    int X1 = c.X + 1 < g.Nx ? c.X + 1 : g.Nx - 1;
    int Y1 = c.Y + 1 < g.Ny ? c.Y + 1 : g.Ny - 1;
    int Z1 = c.Z + 1 < g.Nz ? c.Z + 1 : g.Nz - 1;
    c.cell_ok = (c.X + 1 < g.Nx) && (c.Y + 1 < g.Ny) && (c.Z + 1 < g.Nz);
    const float *r00 = field + ((int64_t)c.Z * g.Ny + c.Y) * g.pitch + g.xorg;
    const float *r01 = field + ((int64_t)c.Z * g.Ny + Y1) * g.pitch + g.xorg;
    const float *r10 = field + ((int64_t)Z1 * g.Ny + c.Y) * g.pitch + g.xorg;
    const float *r11 = field + ((int64_t)Z1 * g.Ny + Y1) * g.pitch + g.xorg;
    float a0, a1, b0, b1, c0, c1, d0, d1;
    if (X1 != c.X) {       // columns X, X+1 of the four rows: one 8-byte (4-byte aligned) load each
        const float2u pa = *(const float2u *)(r00 + c.X), pb = *(const float2u *)(r01 + c.X);
        const float2u pc = *(const float2u *)(r10 + c.X), pd = *(const float2u *)(r11 + c.X);
        a0 = pa.x; a1 = pa.y; b0 = pb.x; b1 = pb.y; c0 = pc.x; c1 = pc.y; d0 = pd.x; d1 = pd.y;
    } else {
        a0 = a1 = r00[c.X]; b0 = b1 = r01[c.X]; c0 = c1 = r10[c.X]; d0 = d1 = r11[c.X];
    }
    c.v[0] = (double)a0 - g.iso; c.v[1] = (double)a1 - g.iso;
    c.v[2] = (double)b1 - g.iso; c.v[3] = (double)b0 - g.iso;
    c.v[4] = (double)c0 - g.iso; c.v[5] = (double)c1 - g.iso;
    c.v[6] = (double)d1 - g.iso; c.v[7] = (double)d0 - g.iso;
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) idx |= (c.v[i] > 0.0 ? 1 : 0) << i;
    c.index = idx;
    int s0 = idx & 1;
    c.flags = (s0 != ((idx >> 1) & 1) ? 1 : 0) | (s0 != ((idx >> 3) & 1) ? 2 : 0) | (s0 != ((idx >> 4) & 1) ? 4 : 0);
}

__global__ __launch_bounds__(256) void mc_eval_kernel(const float *__restrict__ field, const McGrid g,
                                                      const u64 *__restrict__ vox_key, int64_t na,
                                                      u32 *__restrict__ vox_counts, uint8_t *__restrict__ vox_flags,
                                                      const u64 *__restrict__ na_dev)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    if (na_dev != nullptr) {
        // (capped call) entries beyond the list count as nothing in the scan; a list longer than the buffer was not written
        // completely (tomo_mc_list_capped skips what does not fit), so none of its keys may be trusted -- the caller redoes it
        const u64 nad = *na_dev;
        if (nad > (u64)na || (u64)i >= nad) {
            vox_counts[i] = 0u;
            vox_flags[i] = 0;
            return;
        }
    }
    Cell c;
    load_cell(field, g, vox_key[i], c);
    int ntri = 0;
    if (c.cell_ok && c.index != 0 && c.index != 255) {
        McTiling t = mc_cell_tiling(c.v, c.index);
        ntri = t.ntri;
        if (t.centre) c.flags |= 8;
    }
    vox_counts[i] = ((u32)ntri << 16) | (u32)__popc(c.flags);
    vox_flags[i] = (uint8_t)c.flags;
}

static int mc_eval_launch(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                          const unsigned long long *vox_key, int64_t na, uint32_t *vox_counts, uint8_t *vox_flags,
                          const unsigned long long *na_dev, void *stream);

TOMO_API int tomo_mc_eval(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                          const unsigned long long *vox_key, int64_t na, uint32_t *vox_counts, uint8_t *vox_flags,
                          void *stream)
{
    return mc_eval_launch(field, Nz, Ny, Nx, pitch, xorg, iso, vox_key, na, vox_counts, vox_flags, nullptr, stream);
}

// The same over a list buffer of `cap` entries of which only the first *na_dev (device memory, e.g. the total of the
// segment scan) are valid: the rest get count 0, so a scan over all `cap` counts gives the same totals.
TOMO_API int tomo_mc_eval_capped(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                                 const unsigned long long *vox_key, int64_t cap, const unsigned long long *na_dev,
                                 uint32_t *vox_counts, uint8_t *vox_flags, void *stream)
{
    if (!na_dev) return TOMO_E_ARG;
    return mc_eval_launch(field, Nz, Ny, Nx, pitch, xorg, iso, vox_key, cap, vox_counts, vox_flags, na_dev, stream);
}

static int mc_eval_launch(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                          const unsigned long long *vox_key, int64_t na, uint32_t *vox_counts, uint8_t *vox_flags,
                          const unsigned long long *na_dev, void *stream)
{
    McGrid g;
    int rc = make_grid(g, field, Nz, Ny, Nx, pitch, xorg, iso);
    if (rc) return rc;
    if (!vox_key || !vox_counts || !vox_flags) return TOMO_E_ARG;
    if (na <= 0) return TOMO_OK;
    int64_t blocks = ceil_div64(na, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_eval_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, g,
                       (const u64 *)vox_key, na, vox_counts, vox_flags, (const u64 *)na_dev);
    return tomo_status();
}

// index of the vertex (owner voxel key, slot).  The owner's position in the active-voxel list is its segment's offset
// plus its rank among the set bits of the segment's four ballots (list order: lanes ascending, k ascending within a lane,
// see mc_list_kernel) -- two independent loads and a few popcounts instead of a binary search over vox_key.
__device__ static inline u32 find_vertex(u64 okey, int slot, const McGrid &g, const u64 *__restrict__ seg_act,
                                         const u32 *__restrict__ seg_aoff, const u32 *__restrict__ vox_voff,
                                         const uint8_t *__restrict__ vox_flags)
{
    const u64 row = okey >> (KEY_XBITS + 2);
    const u32 c = ((u32)(okey >> 2) & ((1u << KEY_XBITS) - 1u)) + (u32)g.xorg + SEG_SHIFT;      // float column + shift
    const u64 seg = row * (u64)g.segs_per_row + (c >> 8);
    const u32 a0 = seg_aoff[seg], a1 = seg_aoff[seg + 1];
    if (a1 == a0) return 0xffffffffu;                    // empty segment: its ballot record was never written
    const Rec4 r = load_rec(seg_act, (int64_t)seg);
    const int L = (int)((c & 255u) >> 2), k = (int)(c & 3u);
    const u64 bk = k == 0 ? r.b[0] : (k == 1 ? r.b[1] : (k == 2 ? r.b[2] : r.b[3]));
    if (!((bk >> L) & 1ull)) return 0xffffffffu;
    const u64 m = (1ull << L) - 1ull;
    u32 rank = (u32)(__popcll(r.b[0] & m) + __popcll(r.b[1] & m) + __popcll(r.b[2] & m) + __popcll(r.b[3] & m));
    rank += (k > 0 ? (u32)((r.b[0] >> L) & 1ull) : 0u) + (k > 1 ? (u32)((r.b[1] >> L) & 1ull) : 0u) +
            (k > 2 ? (u32)((r.b[2] >> L) & 1ull) : 0u);
    const u32 pos = a0 + rank;
    const u32 fl = vox_flags[pos];
    if (!(fl & (1u << slot))) return 0xffffffffu;
    return vox_voff[pos] + (u32)__popc(fl & ((1u << slot) - 1u));
}

// (vertex base << 4 | flags) of the active voxel `okey`, or 0xffffffff if it is not in the list
__device__ static inline u32 find_owner(u64 okey, const McGrid &g, const u64 *__restrict__ seg_act,
                                        const u32 *__restrict__ seg_aoff, const u32 *__restrict__ vox_voff,
                                        const uint8_t *__restrict__ vox_flags, u32 *flags_out)
{
    const u64 row = okey >> (KEY_XBITS + 2);
    const u32 c = ((u32)(okey >> 2) & ((1u << KEY_XBITS) - 1u)) + (u32)g.xorg + SEG_SHIFT;
    const u64 seg = row * (u64)g.segs_per_row + (c >> 8);
    const u32 a0 = seg_aoff[seg], a1 = seg_aoff[seg + 1];
    if (a1 == a0) return 0xffffffffu;
    const Rec4 r = load_rec(seg_act, (int64_t)seg);
    const int L = (int)((c & 255u) >> 2), k = (int)(c & 3u);
    const u64 bk = k == 0 ? r.b[0] : (k == 1 ? r.b[1] : (k == 2 ? r.b[2] : r.b[3]));
    if (!((bk >> L) & 1ull)) return 0xffffffffu;
    const u64 m = (1ull << L) - 1ull;
    u32 rank = (u32)(__popcll(r.b[0] & m) + __popcll(r.b[1] & m) + __popcll(r.b[2] & m) + __popcll(r.b[3] & m));
    rank += (k > 0 ? (u32)((r.b[0] >> L) & 1ull) : 0u) + (k > 1 ? (u32)((r.b[1] >> L) & 1ull) : 0u) +
            (k > 2 ? (u32)((r.b[2] >> L) & 1ull) : 0u);
    const u32 pos = a0 + rank;
    *flags_out = vox_flags[pos];
    return vox_voff[pos];
}

__device__ static inline u32 slot_vertex(u32 base, u32 flags, int slot)
{   // index of the vertex in `slot` of a voxel whose first vertex is `base`, 0xffffffff if it has none there
    if (base == 0xffffffffu || !(flags & (1u << slot))) return 0xffffffffu;
    return base + (u32)__popc(flags & ((1u << slot) - 1u));
}

__global__ __launch_bounds__(256) void mc_emit_kernel(const float *__restrict__ field, const McGrid g,
                                                      const u64 *__restrict__ vox_key, int64_t na,
                                                      const u64 *__restrict__ seg_act,
                                                      const u32 *__restrict__ seg_aoff, const u32 *__restrict__ vox_voff,
                                                      const u32 *__restrict__ vox_foff, const uint8_t *__restrict__ vox_flags,
                                                      int z_offset, u64 *__restrict__ vkey, float *__restrict__ vpos,
                                                      int32_t *__restrict__ faces, u64 *__restrict__ totals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    Cell c;
    u64 key = vox_key[i];
    load_cell(field, g, key, c);
    const int flags = vox_flags[i];
    u32 vo = vox_voff[i];
    const int Zg = c.Z + z_offset;   // slab offset: positions are produced in global padded coordinates
    const float fZ = (float)Zg, fY = (float)c.Y, fX = (float)c.X;
    if (flags & 1) {
        vkey[vo] = key | 0ull;
        float *p = vpos + 3 * (int64_t)vo;
        *(float3u *)p = (float3u){fZ, fY, (float)((double)c.X + mc_edge_offset(c.v[0], c.v[1]))};
        vo++;
    }
    if (flags & 2) {
        vkey[vo] = key | 1ull;
        float *p = vpos + 3 * (int64_t)vo;
        *(float3u *)p = (float3u){fZ, (float)((double)c.Y + mc_edge_offset(c.v[0], c.v[3])), fX};
        vo++;
    }
    if (flags & 4) {
        vkey[vo] = key | 2ull;
        float *p = vpos + 3 * (int64_t)vo;
        *(float3u *)p = (float3u){(float)((double)Zg + mc_edge_offset(c.v[0], c.v[4])), fY, fX};
        vo++;
    }
    if (flags & 8) {
        double ox, oy, oz;
        mc_centre_offset(c.v, &ox, &oy, &oz);
        vkey[vo] = key | 3ull;
        float *p = vpos + 3 * (int64_t)vo;
        *(float3u *)p = (float3u){(float)((double)Zg + oz), (float)((double)c.Y + oy), (float)((double)c.X + ox)};
        vo++;
    }
    if (!(c.cell_ok && c.index != 0 && c.index != 255)) return;
    u32 fo = vox_foff[i];
    const int64_t rowY = (int64_t)g.Ny;
    // Vertex index of every cube edge that carries one (the edge is bichromatic), ONE list lookup per owner voxel:
    // edges 1,9 belong to voxel (x+1,y,z), 2,11 to (x,y+1,z), 4,7 to (x,y,z+1), 10 / 5 / 6 to the three diagonal ones.
    const int ix = c.index;
#define BICH(a, b) ((((ix >> (a)) ^ (ix >> (b))) & 1) != 0)
    const bool e1 = BICH(1, 2), e2 = BICH(2, 3), e4 = BICH(4, 5), e5 = BICH(5, 6), e6 = BICH(6, 7), e7 = BICH(7, 4);
    const bool e9 = BICH(1, 5), e10 = BICH(2, 6), e11 = BICH(3, 7);
#undef BICH
    u32 id[13];
    id[0] = slot_vertex(vox_voff[i], (u32)flags, 0);
    id[3] = slot_vertex(vox_voff[i], (u32)flags, 1);
    id[8] = slot_vertex(vox_voff[i], (u32)flags, 2);
    id[12] = slot_vertex(vox_voff[i], (u32)flags, 3);
    {   // (x+1, y, z): the next entry of the list if it is active at all (coalesced loads)
        const int64_t in = i + 1 < na ? i + 1 : i;
        const u64 kn = vox_key[in];
        u32 base = vox_voff[in], fl = vox_flags[in];
        if (!((e1 || e9) && kn == key + 4ull)) { base = 0xffffffffu; fl = 0; }
        id[1] = slot_vertex(base, fl, 1); id[9] = slot_vertex(base, fl, 2);
    }
    {   // The five other owner voxels: (y+1), (z+1), (y+1,x+1), (z+1,x+1), (y+1,z+1).  Their lookups are three dependent
        // loads each (segment offsets -> ballot record -> flags / vertex base); run as straight-line code in three
        // phases, all five in flight per phase, instead of five branches that each serialise their own chain.
        // A lookup that is not needed (edge not bichromatic) goes to this voxel's own key: always a valid address.
        const bool need[5] = {e2 || e11, e4 || e7, e10, e5, e6};
        const u64 okey[5] = {make_key(c.row + 1, c.X, 0), make_key(c.row + rowY, c.X, 0), make_key(c.row + 1, c.X + 1, 0),
                             make_key(c.row + rowY, c.X + 1, 0), make_key(c.row + 1 + rowY, c.X, 0)};
        u64 seg[5];
        u32 cc[5], a0[5], a1[5];
#pragma unroll
        for (int n = 0; n < 5; n++) {
            const u64 k = need[n] ? okey[n] : key;
            cc[n] = ((u32)(k >> 2) & ((1u << KEY_XBITS) - 1u)) + (u32)g.xorg + SEG_SHIFT;
            seg[n] = (k >> (KEY_XBITS + 2)) * (u64)g.segs_per_row + (cc[n] >> 8);
            a0[n] = seg_aoff[seg[n]];
            a1[n] = seg_aoff[seg[n] + 1];
        }
        Rec4 rec[5];
#pragma unroll
        for (int n = 0; n < 5; n++) rec[n] = load_rec(seg_act, (int64_t)seg[n]);     // garbage for an empty segment: masked below
        u32 pos[5];
        bool have[5];
#pragma unroll
        for (int n = 0; n < 5; n++) {
            const Rec4 &r = rec[n];
            const int L = (int)((cc[n] & 255u) >> 2), k = (int)(cc[n] & 3u);
            const u64 bk = k == 0 ? r.b[0] : (k == 1 ? r.b[1] : (k == 2 ? r.b[2] : r.b[3]));
            const u64 m = (1ull << L) - 1ull;
            u32 rank = (u32)(__popcll(r.b[0] & m) + __popcll(r.b[1] & m) + __popcll(r.b[2] & m) + __popcll(r.b[3] & m));
            rank += (k > 0 ? (u32)((r.b[0] >> L) & 1ull) : 0u) + (k > 1 ? (u32)((r.b[1] >> L) & 1ull) : 0u) +
                    (k > 2 ? (u32)((r.b[2] >> L) & 1ull) : 0u);
            have[n] = need[n] && a1[n] != a0[n] && ((bk >> L) & 1ull);
            pos[n] = have[n] ? a0[n] + rank : (u32)i;
        }
        u32 fl[5], base[5];
#pragma unroll
        for (int n = 0; n < 5; n++) { fl[n] = vox_flags[pos[n]]; base[n] = vox_voff[pos[n]]; }
#pragma unroll
        for (int n = 0; n < 5; n++) if (!have[n]) { fl[n] = 0; base[n] = 0xffffffffu; }
        id[2] = slot_vertex(base[0], fl[0], 0); id[11] = slot_vertex(base[0], fl[0], 2);
        id[4] = slot_vertex(base[1], fl[1], 0); id[7] = slot_vertex(base[1], fl[1], 1);
        id[10] = slot_vertex(base[2], fl[2], 2);
        id[5] = slot_vertex(base[3], fl[3], 1);
        id[6] = slot_vertex(base[4], fl[4], 0);
    }
    // the tiling (table walks, the ambiguity tests) after the lookups were issued: their latency hides behind it
    McTiling t = mc_cell_tiling(c.v, c.index);
    bool bad = false;
    for (int tI = 0; tI < t.ntri; tI++) {
        int32_t tv[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int ed = t.tris[3 * tI + j];
            u32 v = id[0];                          // select chain: id[] stays in registers
#pragma unroll
            for (int e = 1; e < 13; e++) v = ed == e ? id[e] : v;
            bad |= (v == 0xffffffffu);
            tv[j] = (int32_t)v;
        }
        int32_t *fp = faces + 3 * (int64_t)fo;
        *(int3u *)fp = (int3u){tv[2], tv[1], tv[0]};    // np.fliplr(faces) of the skimage wrapper
        fo++;
    }
    if (bad) atomicAdd(&totals[3], 1ull);
}

TOMO_API int tomo_mc_emit(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                          const unsigned long long *vox_key, int64_t na, const unsigned long long *seg_act,
                          const uint32_t *seg_aoff, const uint32_t *vox_voff, const uint32_t *vox_foff,
                          const uint8_t *vox_flags, int z_offset, unsigned long long *vkey, float *vpos, int32_t *faces,
                          unsigned long long *totals, void *stream)
{
    McGrid g;
    int rc = make_grid(g, field, Nz, Ny, Nx, pitch, xorg, iso);
    if (rc) return rc;
    if (!vox_key || !seg_act || !seg_aoff || !vox_voff || !vox_foff || !vox_flags || !vkey || !vpos || !faces || !totals)
        return TOMO_E_ARG;
    if (na <= 0) return TOMO_OK;
    int64_t blocks = ceil_div64(na, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, g,
                       (const u64 *)vox_key, na, (const u64 *)seg_act, seg_aoff, vox_voff, vox_foff, vox_flags, z_offset,
                       (u64 *)vkey, vpos, faces, (u64 *)totals);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ first touch
// manifold=False exposes skimage's vertex NUMBERING: vertices are numbered in the order the serial z->y->x
// scan first touches them (cell order, then order of first appearance in the cell's triangle list).  Every
// existing cell around a bichromatic edge references it, so the vertex on an edge is created by the
// scan-minimal cell around it: for an x edge with owner voxel (ox,oy,oz) that is cell (ox, oy-1|oy, oz-1|oz)
// taking the lower index where it exists; likewise for y and z edges; a centre vertex by its own cell.
// mode 0: created[i] = number of vertices cell i creates;  mode 1: ft_rank[provisional id] = base[i] + j.
__global__ __launch_bounds__(256) void mc_first_touch_kernel(const float *__restrict__ field, const McGrid g,
                                                             const u64 *__restrict__ vox_key, int64_t na,
                                                             const u64 *__restrict__ seg_act,
                                                             const u32 *__restrict__ seg_aoff,
                                                             const u32 *__restrict__ vox_voff,
                                                             const uint8_t *__restrict__ vox_flags, int mode,
                                                             u32 *__restrict__ created, const u32 *__restrict__ base,
                                                             int32_t *__restrict__ ft_rank, u64 *__restrict__ totals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    Cell c;
    load_cell(field, g, vox_key[i], c);
    u32 n = 0;
    if (c.cell_ok && c.index != 0 && c.index != 255) {
        McTiling t = mc_cell_tiling(c.v, c.index);
        const int flags = vox_flags[i];
        u32 seen = 0;
        const int64_t rowY = (int64_t)g.Ny;
        for (int k = 0; k < 3 * t.ntri; k++) {
            int ed = t.tris[k];
            if (seen & (1u << ed)) continue;
            seen |= 1u << ed;
            int dx = (ed == 1 || ed == 5 || ed == 9 || ed == 10) ? 1 : 0;
            int dy = (ed == 2 || ed == 6 || ed == 10 || ed == 11) ? 1 : 0;
            int dz = (ed >= 4 && ed <= 7) ? 1 : 0;
            int slot = ed == 12 ? 3 : (ed >= 8 ? 2 : (ed & 1));
            int ox = c.X + dx, oy = c.Y + dy, oz = c.Z + dz;
            bool creator;
            if (slot == 0) creator = (dy == (oy > 0 ? 1 : 0)) && (dz == (oz > 0 ? 1 : 0));
            else if (slot == 1) creator = (dx == (ox > 0 ? 1 : 0)) && (dz == (oz > 0 ? 1 : 0));
            else if (slot == 2) creator = (dx == (ox > 0 ? 1 : 0)) && (dy == (oy > 0 ? 1 : 0));
            else creator = true;
            if (!creator) continue;
            if (mode == 1) {
                u32 v;
                if ((dx | dy | dz) == 0) v = (flags & (1 << slot)) ? vox_voff[i] + (u32)__popc(flags & ((1 << slot) - 1)) : 0xffffffffu;
                else v = find_vertex(make_key(c.row + dy + dz * rowY, ox, 0), slot, g, seg_act, seg_aoff, vox_voff, vox_flags);
                if (v == 0xffffffffu) atomicAdd(&totals[3], 1ull);
                else ft_rank[v] = (int32_t)(base[i] + n);
            }
            n++;
        }
    }
    if (mode == 0) created[i] = n;
}

TOMO_API int tomo_mc_first_touch(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                                 const unsigned long long *vox_key, int64_t na, const unsigned long long *seg_act,
                                 const uint32_t *seg_aoff, const uint32_t *vox_voff, const uint8_t *vox_flags, int mode,
                                 uint32_t *created,
                                 const uint32_t *base, int32_t *ft_rank, unsigned long long *totals, void *stream)
{
    McGrid g;
    int rc = make_grid(g, field, Nz, Ny, Nx, pitch, xorg, iso);
    if (rc) return rc;
    if (!vox_key || !seg_act || !seg_aoff || !vox_voff || !vox_flags || !totals || (mode == 0 && !created) ||
        (mode == 1 && (!base || !ft_rank)) || (mode != 0 && mode != 1))
        return TOMO_E_ARG;
    if (na <= 0) return TOMO_OK;
    int64_t blocks = ceil_div64(na, 256);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_first_touch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, g,
                       (const u64 *)vox_key, na, (const u64 *)seg_act, seg_aoff, vox_voff, vox_flags, mode, created, base, ft_rank,
                       (u64 *)totals);
    return tomo_status();
}

// ==========================================================================================================
// "mc3": the same marching cubes with the mesh stages folded in (surface_extractor.py:55-72 in one chain).
//
//   list  -> eval -> scan -> vertices -> [segmented sort + rank: mesh.hip] -> faces
//
// What changes against passes 2-4 above, and why (1024^3 ellipsoid, rocprofv3: 810 us for MC + finalize + unique +
// remap before; the device-wide scans alone were 190 us of it, the unique stage 255 us):
//   * no device-wide scan primitive anywhere: counts are reduced per block of 256 entries by the kernel that produces or
//     consumes them, ONE small single-block kernel scans the block sums, and the consumers add block base + local prefix;
//   * eval does all float64 work of a cell once: MC33 tiling (a 4-byte reference to the tiling row is kept, so nothing
//     re-evaluates it), the vertex flags and the vertex positions along the owned edges (staged as float32, 12 B/voxel);
//   * vertices are written FINALISED (-1 shift, slice-depth map, y/x scale: surface_extractor.py:57-65, :82-113) and
//     already PARTITIONED the way np.unique's order needs them -- per slice: the in-plane vertices (x / y edges), then
//     the vertices between this plane and the next (z edges, cell centres) -- each with its 32-bit sort key (y' in a
//     plane, z' between planes).  What remains of np.unique(axis=0) is one segmented sort inside the 2 Nz buckets and one
//     gather that checks the result is STRICTLY ascending (no duplicates, no rounding coincidence): then the sorted
//     position IS the final index.  Anything else is counted and the caller redoes the stage with the general sort;
//   * a vertex is named by (list position of its owner voxel, slot): id = 4 pos + slot.  A triangle corner therefore
//     needs the owner's list position only (segment offset + ballot rank: two loads in flight together) and ONE gather
//     from the table id -> final index; faces are written once, as final int64 triples, after the sort.
// Everything reads the list length / totals from device memory (`tot`), so the whole chain can be enqueued into buffers
// sized from a hint before any count has come back to the host; overflows are flagged in tot[3], nothing is written
// past a buffer.
//
// tot (device uint64[8]): [0] active voxels (list length)  [1] vertices  [2] triangles  [3] overflow flags (1 list,
// 2 vertices, 4 triangles)  [4] places where the sorted rows do not ascend strictly  [5] degenerate triangles
// [6] triangle corners without vertex (internal consistency, must stay 0)
#define MC3_BLK 256
#define MC3_LOC_A(l) ((l) & 1023u)
#define MC3_LOC_B(l) (((l) >> 10) & 1023u)
#define MC3_LOC_T(l) ((l) >> 20)

// ---- sums of seg_cnt per block of 256 segments: one wave per block of segments, 16 B per lane
__global__ __launch_bounds__(256) void mc3_segsum_kernel(const u32 *__restrict__ seg_cnt, int64_t nseg, u32 *__restrict__ seg_blk,
                                                         int64_t nblk)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= nblk) return;
    const int64_t i0 = b * MC3_BLK + lane * 4;
    u32 s = 0;
    if (i0 + 3 < nseg) {
        const uint4 v = *(const uint4 *)(seg_cnt + i0);
        s = v.x + v.y + v.z + v.w;
    } else {
        for (int k = 0; k < 4; k++) if (i0 + k < nseg) s += seg_cnt[i0 + k];
    }
    s = wave_sum(s);
    if (lane == 0) seg_blk[b] = s;
}

// ---- exclusive scan, in place, of K u32 arrays of n entries (a[k * stride + i]) by ONE workgroup of 1024 threads
// (n = a few 10^4: block sums); total[k] = sum of array k.  8 consecutive entries per thread and round (two 16-byte loads),
// wave scans, one more wave scan across the 16 wave totals.
// total_out[k] is the sum in 64 bits: the prefixes themselves are 32-bit (what every consumer indexes with), so a caller must
// treat a total beyond 2^32 - 1 as an overflow of its buffers (a round of 8192 entries cannot wrap: block sums are < 2^19).
template <int K>
__device__ static inline void scan_small_inplace(u32 *__restrict__ a, int64_t stride, int64_t n, u64 *total_out, u32 *lds /* K * 16 + K */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u32 carry[K];
    u64 wide[K];
#pragma unroll
    for (int k = 0; k < K; k++) { carry[k] = 0; wide[k] = 0; }
    for (int64_t base = 0; base < n; base += 1024 * 8) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 8;
        u32 v[K][8], s[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            u32 *p = a + k * stride + i0;
            if (i0 + 7 < n && ((uintptr_t)p & 15) == 0) {
                const uint4 lo = *(const uint4 *)p, hi = *(const uint4 *)(p + 4);
                v[k][0] = lo.x; v[k][1] = lo.y; v[k][2] = lo.z; v[k][3] = lo.w; v[k][4] = hi.x; v[k][5] = hi.y; v[k][6] = hi.z; v[k][7] = hi.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) v[k][j] = i0 + j < n ? p[j] : 0u;
            }
            s[k] = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) s[k] += v[k][j];
        }
        u32 inc[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            inc[k] = wave_inclusive_scan(s[k]);
            if (lane == 63) lds[k * 16 + w] = inc[k];
        }
        __syncthreads();
        if (w == 0) {                                             // the 16 wave totals of every array: one more wave scan
#pragma unroll
            for (int k = 0; k < K; k++) {
                const u32 t = lane < 16 ? lds[k * 16 + lane] : 0u;
                const u32 ti = wave_inclusive_scan(t);
                if (lane < 16) lds[k * 16 + lane] = ti - t;
                if (lane == 15) lds[K * 16 + k] = ti;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) {
            u32 ex = carry[k] + lds[k * 16 + w] + inc[k] - s[k];
            u32 *p = a + k * stride + i0;
            if (i0 + 7 < n && ((uintptr_t)p & 15) == 0) {
                uint4 lo, hi;
                lo.x = ex; ex += v[k][0]; lo.y = ex; ex += v[k][1]; lo.z = ex; ex += v[k][2]; lo.w = ex; ex += v[k][3];
                hi.x = ex; ex += v[k][4]; hi.y = ex; ex += v[k][5]; hi.z = ex; ex += v[k][6]; hi.w = ex;
                *(uint4 *)p = lo; *(uint4 *)(p + 4) = hi;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) { if (i0 + j < n) p[j] = ex; ex += v[k][j]; }
            }
            carry[k] += lds[K * 16 + k];
            wide[k] += lds[K * 16 + k];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < K; k++) total_out[k] = wide[k];
}

__global__ __launch_bounds__(1024) void mc3_scan_seg_kernel(u32 *__restrict__ seg_blk, int64_t nblk, u64 *__restrict__ tot)
{
    __shared__ u32 lds[17];
    u64 total;
    scan_small_inplace<1>(seg_blk, 0, nblk, &total, lds);
    if (threadIdx.x == 0) {
        tot[0] = total;                                           // exact even beyond 2^32: the host then reports "surface too large"
        tot[1] = tot[2] = tot[4] = tot[5] = tot[6] = tot[7] = 0ull;
        tot[3] = total > 0xffffffffull ? 1ull : 0ull;             // the 32-bit prefixes have wrapped: nothing downstream may be trusted
    }
}

// ---- list: as mc_list_kernel, but the segment's offset comes from its block's base plus the prefix inside the block;
// seg_aoff (nseg + 1 entries) is written here for the lookups of the faces pass
__global__ __launch_bounds__(MC3_BLK) void mc3_list_kernel(const McGrid g, const u32 *__restrict__ seg_cnt,
                                                           const u32 *__restrict__ seg_blk, const u64 *__restrict__ seg_act,
                                                           int64_t nseg, u32 *__restrict__ seg_aoff, u64 *__restrict__ vox_key,
                                                           u32 cap, u64 *__restrict__ tot)
{
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t seg = (int64_t)blockIdx.x * MC3_BLK + threadIdx.x;
    const u32 cnt = seg < nseg ? seg_cnt[seg] : 0u;
    const u32 inc = wave_inclusive_scan(cnt);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 o = seg_blk[blockIdx.x] + inc - cnt;
    for (int k = 0; k < w; k++) o += wsum[k];
    if (seg >= nseg) return;
    seg_aoff[seg] = o;
    if (seg == nseg - 1) seg_aoff[nseg] = o + cnt;
    if (cnt == 0) return;                                       // empty: its ballot record was never written
    if ((u64)o + cnt > (u64)cap || tot[0] > (u64)cap) {           // the list does not fit (or its 32-bit offsets have wrapped): flagged, nothing written
        if (o <= cap) atomicOr((unsigned long long *)&tot[3], 1ull);
        return;
    }
    int64_t row;
    const int s = divmod_pos(seg, g.segs_per_row, &row);
    const ulonglong2 *q = (const ulonglong2 *)(seg_act + seg * 4);
    const ulonglong2 lo = q[0], hi = q[1];
    const u64 b0 = lo.x, b1 = lo.y, b2 = hi.x, b3 = hi.y;
    const int Xs = s * SEG - SEG_SHIFT - g.xorg;
    u64 any = b0 | b1 | b2 | b3;
    while (any) {
        const int L = __ffsll((long long)any) - 1;
        any &= any - 1;
        const int X0 = Xs + 4 * L;
        if ((b0 >> L) & 1ull) vox_key[o++] = make_key(row, X0, 0);
        if ((b1 >> L) & 1ull) vox_key[o++] = make_key(row, X0 + 1, 0);
        if ((b2 >> L) & 1ull) vox_key[o++] = make_key(row, X0 + 2, 0);
        if ((b3 >> L) & 1ull) vox_key[o++] = make_key(row, X0 + 3, 0);
    }
}

TOMO_API int tomo_mc3_list(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_cnt, const unsigned long long *seg_act,
                           uint32_t *seg_blk, uint32_t *seg_aoff, unsigned long long *vox_key, int64_t cap,
                           unsigned long long *tot, void *stream)
{
    if (Nz < 2 || Ny < 2 || Nx < 2 || !seg_cnt || !seg_act || !seg_blk || !seg_aoff || !vox_key || !tot) return TOMO_E_ARG;
    if (cap <= 0 || cap >= 0x1fffffffll) return TOMO_E_SIZE;        // ids are 4 * position + slot in a signed 32-bit index
    if (Nx >= (1 << KEY_XBITS)) return TOMO_E_SIZE;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    const int64_t nseg = (int64_t)Nz * Ny * g.segs_per_row, nblk = ceil_div64(nseg, MC3_BLK);
    if (nblk > 0x7fffffff) return TOMO_E_SIZE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mc3_segsum_kernel, dim3((unsigned)ceil_div64(nblk, 4)), dim3(256), 0, s, seg_cnt, nseg, seg_blk, nblk);
    hipLaunchKernelGGL(mc3_scan_seg_kernel, dim3(1), dim3(1024), 0, s, seg_blk, nblk, (u64 *)tot);
    hipLaunchKernelGGL(mc3_list_kernel, dim3((unsigned)nblk), dim3(MC3_BLK), 0, s, g, seg_cnt, (const u32 *)seg_blk,
                       (const u64 *)seg_act, nseg, seg_aoff, (u64 *)vox_key, (u32)cap, (u64 *)tot);
    return tomo_status();
}

// ---- eval: everything a cell needs in float64, once.  Per list entry: vox_loc = exclusive prefix INSIDE the block of
// 256 entries of (in-plane vertices | between-plane vertices << 10 | triangles << 20); vox_til = tiling reference << 4 |
// triangles; vox_flags; vox_f3 = the coordinate along the owned x / y / z edge as float32 (z, y, x order; garbage where
// the flag is clear), vox_c3 = the centre vertex (written only when there is one).  blk3[3][nblk] = block sums.
__global__ __launch_bounds__(MC3_BLK) void mc3_eval_kernel(const float *__restrict__ field, const McGrid g,
                                                           const u64 *__restrict__ vox_key, int64_t cap,
                                                           const u64 *__restrict__ tot, int z_offset,
                                                           u32 *__restrict__ vox_loc, int32_t *__restrict__ vox_til,
                                                           uint8_t *__restrict__ vox_flags, uint16_t *__restrict__ vox_used,
                                                           float *__restrict__ vox_f3, float *__restrict__ vox_c3,
                                                           u32 *__restrict__ blk3, int64_t nblk)
{
    __shared__ u32 wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * MC3_BLK + threadIdx.x;
    const u64 na = tot[0];
    const bool live = i < cap && (u64)i < na && na <= (u64)cap;
    u32 packed = 0;
    if (live) {
        Cell c;
        load_cell(field, g, vox_key[i], c);
        int ntri = 0;
        int32_t til = 0;
        u32 used = 0;
        if (c.cell_ok && c.index != 0 && c.index != 255) {
            const McTiling t = mc_cell_tiling(c.v, c.index);
            ntri = t.ntri;
            if (t.centre) c.flags |= 8;
            til = (int32_t)(((intptr_t)t.tris - (intptr_t)LUT_TILING1) * 16 + ntri);
            // the cube edges that carry a vertex (every MC33 tiling uses exactly the bichromatic edges of its cube), bit 12: the
            // centre vertex -- what the faces pass looks up, from the corner signs (a loop over the tiling row would do, but
            // it sends the faces kernel's registers to scratch memory)
            const int ix = c.index;
#define BICH(a, b) ((u32)(((ix >> (a)) ^ (ix >> (b))) & 1))
            used = BICH(0, 1) | BICH(1, 2) << 1 | BICH(2, 3) << 2 | BICH(3, 0) << 3 | BICH(4, 5) << 4 | BICH(5, 6) << 5 | BICH(6, 7) << 6 |
                   BICH(7, 4) << 7 | BICH(0, 4) << 8 | BICH(1, 5) << 9 | BICH(2, 6) << 10 | BICH(3, 7) << 11 | (t.centre ? 1u << 12 : 0u);
#undef BICH
            if (ntri == 0) used = 0;
        }
        const int Zg = c.Z + z_offset;
        float3u f3;
        f3.x = (float)((double)Zg + mc_edge_offset(c.v[0], c.v[4]));
        f3.y = (float)((double)c.Y + mc_edge_offset(c.v[0], c.v[3]));
        f3.z = (float)((double)c.X + mc_edge_offset(c.v[0], c.v[1]));
        *(float3u *)(vox_f3 + 3 * i) = f3;
        if (c.flags & 8) {
            double ox, oy, oz;
            mc_centre_offset(c.v, &ox, &oy, &oz);
            *(float3u *)(vox_c3 + 3 * i) = (float3u){(float)((double)Zg + oz), (float)((double)c.Y + oy), (float)((double)c.X + ox)};
        }
        vox_til[i] = til;
        vox_flags[i] = (uint8_t)c.flags;
        vox_used[i] = (uint16_t)used;
        packed = (u32)__popc(c.flags & 3) | ((u32)__popc(c.flags & 12) << 10) | ((u32)ntri << 20);
    }
    const u32 inc = wave_inclusive_scan(packed);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    u32 ex = inc - packed;
    for (int k = 0; k < w; k++) ex += wsum[k];
    if (i < cap) vox_loc[i] = ex;
    if (threadIdx.x == MC3_BLK - 1) {
        const u32 sum = ex + packed;
        blk3[blockIdx.x] = MC3_LOC_A(sum);
        blk3[nblk + blockIdx.x] = MC3_LOC_B(sum);
        blk3[2 * nblk + blockIdx.x] = MC3_LOC_T(sum);
    }
}

// ---- scan: block sums -> exclusive prefixes, totals and the per-slice tables; mc3_bands_kernel then writes the offsets of the
// sort's segments.  slice_tab: sliceA[Nz + 1] | sliceB[Nz + 1] | offsets[(NB + 1) Nz + 1] | merge3[3]  (uint32), NB =
// TOMO_SORT_NB(Ny): per slice NB bands of tomo_sort_band(Ny) owner rows of in-plane vertices, then the between-plane vertices.
// In a plane the vertices arrive grouped by owner row and a vertex of owner row Y has its key y' in [Y, Y + 1] * mm_y, so the
// sort is local to a row -- a fortiori to a band of rows: planes of more than 1280 rows are cut into bands of 512 (2048^2
// slices hold ~6 700 vertices per plane, beyond what the segmented sort keeps in LDS -- 922 us for 3.7 M vertices on a
// middle rank of BASELINE configs[4]; tomo_common.h).  The between-plane keys z' are in no order: those buckets stay whole.
// merge3 = {start of slab 0's between-plane bucket, start of slab 1's plane, its end}: the two runs uq3_merge_kernel merges.
__global__ __launch_bounds__(1024) void mc3_scan_kernel(const McGrid g, const u32 *__restrict__ seg_aoff,
                                                        const u32 *__restrict__ vox_loc, int64_t cap, u32 *__restrict__ blk3,
                                                        int64_t nblk, u32 *__restrict__ slice_tab, u64 *__restrict__ tot,
                                                        int64_t cap_v, int64_t cap_f)
{
    __shared__ u32 lds[3 * 16 + 3];
    u64 totals[3];
    scan_small_inplace<3>(blk3, nblk, nblk, totals, lds);       // ends with a barrier: the prefixes are visible to every thread
    const u64 na = tot[0];
    const u64 totA64 = totals[0], totB64 = totals[1], totT64 = totals[2];
    const u32 totA = (u32)totA64, totB = (u32)totB64;
    const int Nz = g.Nz;
    u32 *sliceA = slice_tab, *sliceB = slice_tab + (Nz + 1);
    const int64_t segs_per_slice = (int64_t)g.Ny * g.segs_per_row;
    const bool fits = na <= (u64)cap;
    for (int Z = threadIdx.x; Z <= Nz; Z += 1024) {
        u32 a = totA, b = totB;
        if (Z < Nz && fits) {
            const u64 s = seg_aoff[(int64_t)Z * segs_per_slice];         // list position of the slice's first voxel
            if (s < na) {
                const u32 l = vox_loc[s];
                a = blk3[s >> 8] + MC3_LOC_A(l);
                b = blk3[nblk + (s >> 8)] + MC3_LOC_B(l);
            }
        }
        sliceA[Z] = a;
        sliceB[Z] = b;
    }
    u64 ov = 0;
    if (totA64 + totB64 > (u64)cap_v) ov |= 2ull;
    if (totT64 > (u64)cap_f) ov |= 4ull;
    if (!fits) ov |= 1ull;
    if (threadIdx.x == 0) {
        tot[1] = totA64 + totB64;
        tot[2] = totT64;
        if (ov) atomicOr((unsigned long long *)&tot[3], (unsigned long long)ov);
    }
}

// one thread per sort segment (Z, b): b < NB the band of owner rows [b, b + 1) * tomo_sort_band(Ny) of plane Z, b == NB the
// between-plane bucket; on overflow (tot[3]) every segment is empty: the sort touches nothing
__global__ __launch_bounds__(256) void mc3_bands_kernel(const McGrid g, const u32 *__restrict__ seg_aoff,
                                                        const u32 *__restrict__ vox_loc, const u32 *__restrict__ blk3,
                                                        int64_t nblk, u32 *__restrict__ slice_tab, const u64 *__restrict__ tot)
{
    const int Nz = g.Nz, NB = TOMO_SORT_NB(g.Ny);
    const int64_t nseg = (int64_t)(NB + 1) * Nz, i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 *sliceA = slice_tab, *sliceB = slice_tab + (Nz + 1);
    u32 *offsets = slice_tab + 2 * (Nz + 1), *merge3 = offsets + nseg + 1;
    const bool ov = tot[3] != 0;
    const u64 na = tot[0];
    if (i == nseg) {
        offsets[nseg] = ov ? 0u : sliceA[Nz] + sliceB[Nz];
        // slab 0's between-plane bucket and slab 1's plane (uq3_merge_kernel; Nz >= 2 always)
        merge3[0] = ov ? 0u : sliceA[1] + sliceB[0];
        merge3[1] = ov ? 0u : sliceA[1] + sliceB[1];
        merge3[2] = ov ? 0u : sliceA[Nz >= 2 ? 2 : 1] + sliceB[1];
        merge3[3] = 0u;                                           // the ticket word of uq3_sortrank_kernel (mesh.hip)
        return;
    }
    if (i > nseg) return;
    const int Z = (int)(i / (NB + 1)), b = (int)(i - (int64_t)Z * (NB + 1));
    u32 o = 0u;
    if (!ov) {
        if (b == NB) {
            o = sliceA[Z + 1] + sliceB[Z];
        } else {
            // in-plane vertices before the first voxel of row (Z, b * BAND): the list position of that voxel, then its prefix
            const u64 s = seg_aoff[((int64_t)Z * g.Ny + (int64_t)b * tomo_sort_band(g.Ny)) * g.segs_per_row];
            const u32 a = s < na ? blk3[s >> 8] + MC3_LOC_A(vox_loc[s]) : sliceA[Nz];
            o = a + sliceB[Z];
        }
    }
    offsets[i] = o;
}

// ---- vertices: final coordinates (surface_extractor.py:57-65, :82-113 -- the arithmetic of vertex_finalize_kernel),
// partitioned by (slice, in-plane / between), one 16-byte record {z', y', x', id} per vertex + its sort key
struct Mc3Final {
    const double *cum, *adj;
    int64_t ncum, nadj;
    int shift;
    float mm_y, mm_x;
};

__device__ static inline void mc3_put_vertex(float z, float y, float x, u32 id, u32 dest, bool between, const Mc3Final &fin,
                                             float4 *__restrict__ vrec, u32 *__restrict__ keys, u32 *__restrict__ idx)
{
    if (fin.shift) { z -= 1.0f; y -= 1.0f; x -= 1.0f; }
    if (fin.nadj > 0) {
        if (z < 0.0f) z = 0.0f;
        else if ((double)z >= (double)(fin.ncum - 1)) z = (float)fin.cum[fin.ncum - 1];
        else {
            const int64_t lo = (int64_t)floorf(z);
            const float frac = z - (float)lo;
            const int64_t k = lo < fin.nadj - 1 ? lo : fin.nadj - 1;
            z = (float)(fin.cum[lo] + (double)frac * fin.adj[k]);
        }
    }
    y = y * fin.mm_y;
    x = x * fin.mm_x;
    vrec[dest] = make_float4(z, y, x, __uint_as_float(id));
    const u32 u = __float_as_uint(between ? z : y);
    keys[dest] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);         // order-preserving float -> uint (fkey32 of mesh.hip)
    if (idx) idx[dest] = dest;                                       // (the rocPRIM path sorts pairs; the fused kernel carries positions)
}

__global__ __launch_bounds__(MC3_BLK) void mc3_vertices_kernel(const McGrid g, const u64 *__restrict__ vox_key, int64_t cap,
                                                               const u64 *__restrict__ tot, const u32 *__restrict__ vox_loc,
                                                               const uint8_t *__restrict__ vox_flags,
                                                               const float *__restrict__ vox_f3, const float *__restrict__ vox_c3,
                                                               const u32 *__restrict__ blk3, int64_t nblk,
                                                               const u32 *__restrict__ slice_tab, int z_offset,
                                                               const Mc3Final fin, float4 *__restrict__ vrec,
                                                               u32 *__restrict__ keys, u32 *__restrict__ idx)
{
    const int64_t i = (int64_t)blockIdx.x * MC3_BLK + threadIdx.x;
    if (tot[3] != 0 || i >= cap || (u64)i >= tot[0]) return;
    // everything this lane needs from the list is fetched up front (independent, coalesced loads in flight together); only
    // the two slice-table entries depend on one of them
    const int flags = vox_flags[i];
    const u64 key = vox_key[i];
    const u32 l = vox_loc[i];
    const float3u f3 = *(const float3u *)(vox_f3 + 3 * i);
    const u32 bA = blk3[i >> 8], bB = blk3[nblk + (i >> 8)];
    const int64_t row = (int64_t)(key >> (KEY_XBITS + 2));
    const int X = (int)((key >> 2) & ((1u << KEY_XBITS) - 1u));
    int64_t zq;
    const int Y = divmod_pos(row, g.Ny, &zq), Z = (int)zq;
    const u32 *sliceA = slice_tab, *sliceB = slice_tab + (g.Nz + 1);
    const u32 sB = sliceB[Z], sA1 = sliceA[Z + 1];
    if (!flags) return;
    u32 dA = sB + bA + MC3_LOC_A(l);                                 // in-plane vertices of slice Z start at sliceA[Z] + sliceB[Z]
    u32 dB = sA1 + bB + MC3_LOC_B(l);                                // between-plane ones after all in-plane ones of the slice
    const float fZ = (float)(Z + z_offset), fY = (float)Y, fX = (float)X;
    const u32 id = (u32)i * 4u;
    if (flags & 1) mc3_put_vertex(fZ, fY, f3.z, id, dA++, false, fin, vrec, keys, idx);
    if (flags & 2) mc3_put_vertex(fZ, f3.y, fX, id + 1u, dA, false, fin, vrec, keys, idx);
    if (flags & 4) mc3_put_vertex(f3.x, fY, fX, id + 2u, dB++, true, fin, vrec, keys, idx);
    if (flags & 8) {
        const float3u c3 = *(const float3u *)(vox_c3 + 3 * i);
        mc3_put_vertex(c3.x, c3.y, c3.z, id + 3u, dB, true, fin, vrec, keys, idx);
    }
}

// ---- faces: final int64 triples.  table[id] = final vertex index of vertex id = 4 * (owner's list position) + slot.
// A block's triangles are ONE contiguous run of the face list (block base + prefix inside the block): they are staged in
// LDS and leave as coalesced 8-byte stores (three scattered 8-byte stores per triangle cost 4x the bytes in partial-line
// writes: 730 MB of WRITE_SIZE for 170 MB of triangles at 1024^3), in windows of MC3_FWIN triangles.
#define MC3_FWIN 1024
// SLAB: the triangles of a Z-slab rank leave with GLOBAL vertex indices (slab.py, the pass without host round trips): row r of
// this rank's list is r + the sum of the lower ranks' kept rows while r < k = rows - rows on the shared top plane, and
// ids_next[r - k] + the upper rank's offset for the shared rows (what the upper rank's tomo_slab_lookup found); every count
// comes from device memory (tot, the all-gathered summaries) and the mapping is applied where the staged indices leave LDS.
struct Mc3SlabMap { const int64_t *gathered; const int32_t *ids_next; int64_t cap_top, cap_v; int rank; };
template <bool SLAB>
__global__ __launch_bounds__(MC3_BLK) void mc3_faces_kernel(const McGrid g, const u64 *__restrict__ vox_key, int64_t cap,
                                                            u64 *__restrict__ tot, const u64 *__restrict__ seg_act,
                                                            const u32 *__restrict__ seg_aoff, const u32 *__restrict__ vox_loc,
                                                            const int32_t *__restrict__ vox_til, const uint16_t *__restrict__ vox_used,
                                                            const u32 *__restrict__ blk3, int64_t nblk,
                                                            const int32_t *__restrict__ table, int64_t *__restrict__ faces,
                                                            int64_t cap_f, const Mc3SlabMap sm)
{
    __shared__ int32_t sf[3 * MC3_FWIN];
    const int64_t i = (int64_t)blockIdx.x * MC3_BLK + threadIdx.x;
    const u64 na = tot[0];
    if (tot[3] != 0) return;                                     // something did not fit (uniform over the grid)
    const u64 fbase = blk3[2 * nblk + blockIdx.x];
    const u64 fend = (int64_t)blockIdx.x + 1 < nblk ? (u64)blk3[2 * nblk + blockIdx.x + 1] : tot[2];
    const u32 total = (u32)(fend - fbase);                       // triangles of this block
    if (total == 0 || fend > (u64)cap_f) return;                 // (uniform over the block)
    int ntri = 0;
    u32 tloc = 0;
    const signed char *tris = LUT_TILING1;
    // final index of the vertex on each of the 13 edges (scalars on purpose: as an array with run-time conditions around
    // its stores the compiler keeps it -- and everything indexed like it -- in scratch memory, 176 B per lane)
    int32_t id0 = -1, id1 = -1, id2 = -1, id3 = -1, id4 = -1, id5 = -1, id6 = -1, id7 = -1, id8 = -1, id9 = -1, id10 = -1, id11 = -1,
            id12 = -1;
    bool bad = false;
    if (i < cap && (u64)i < na) {
        const int32_t til = vox_til[i];
        ntri = til & 15;
        tris = (const signed char *)((intptr_t)LUT_TILING1 + (intptr_t)(til >> 4));
    }
    if (ntri > 0) {
        tloc = MC3_LOC_T(vox_loc[i]);
        const u64 key = vox_key[i];
        const int64_t row = (int64_t)(key >> (KEY_XBITS + 2));
        const int X = (int)((key >> 2) & ((1u << KEY_XBITS) - 1u));
        const int64_t rowY = (int64_t)g.Ny;
        const u32 used = vox_used[i];
        // owner voxels of the 13 possible vertices: this voxel (edges 0, 3, 8, centre 12), (x+1) edges 1, 9, then the five of
        // mc_emit_kernel: (y+1) 2, 11; (z+1) 4, 7; (y+1, x+1) 10; (z+1, x+1) 5; (y+1, z+1) 6.
        // A lookup that is not needed goes to this voxel's own key: always a valid address.
#define MC3_USED(mask) ((used & (mask)) != 0)
#define MC3_OWNER(n, needed, okey)                                                                                                  \
        const bool need##n = (needed);                                                                                             \
        const u64 k##n = need##n ? (okey) : key;                                                                                   \
        const u32 cc##n = ((u32)(k##n >> 2) & ((1u << KEY_XBITS) - 1u)) + (u32)g.xorg + SEG_SHIFT;                                 \
        const u64 seg##n = (k##n >> (KEY_XBITS + 2)) * (u64)g.segs_per_row + (cc##n >> 8);                                         \
        const u32x2u aa##n = *(const u32x2u *)(seg_aoff + seg##n);      /* (offset, next offset): one 8-byte, 4-byte aligned load */  \
        const u32 a0_##n = aa##n.x, a1_##n = aa##n.y;                                                                              \
        const Rec4 rec##n = load_rec(seg_act, (int64_t)seg##n);       /* garbage for an empty segment: masked below */
        MC3_OWNER(2, MC3_USED((1u << 2) | (1u << 11)), make_key(row + 1, X, 0))
        MC3_OWNER(3, MC3_USED((1u << 4) | (1u << 7)), make_key(row + rowY, X, 0))
        MC3_OWNER(4, MC3_USED(1u << 10), make_key(row + 1, X + 1, 0))
        MC3_OWNER(5, MC3_USED(1u << 5), make_key(row + rowY, X + 1, 0))
        MC3_OWNER(6, MC3_USED(1u << 6), make_key(row + 1 + rowY, X, 0))
#undef MC3_OWNER
        // (x+1, y, z) is the next list entry if it is active at all
        const int64_t inx = (u64)(i + 1) < na ? i + 1 : i;
        const bool have1 = vox_key[inx] == key + 4ull;
        const u32 pos0 = (u32)i, pos1 = (u32)inx;
#define MC3_RANK(n)                                                                                                                 \
        bool have##n;                                                                                                              \
        u32 pos##n;                                                                                                                \
        {                                                                                                                          \
            const int L = (int)((cc##n & 255u) >> 2), k = (int)(cc##n & 3u);                                                       \
            const u64 b0 = rec##n.b[0], b1 = rec##n.b[1], b2 = rec##n.b[2], b3 = rec##n.b[3];                                      \
            const u64 bk = k == 0 ? b0 : (k == 1 ? b1 : (k == 2 ? b2 : b3));                                                      \
            const u64 m = (1ull << L) - 1ull;                                                                                      \
            u32 rank = (u32)(__popcll(b0 & m) + __popcll(b1 & m) + __popcll(b2 & m) + __popcll(b3 & m));                          \
            rank += (k > 0 ? (u32)((b0 >> L) & 1ull) : 0u) + (k > 1 ? (u32)((b1 >> L) & 1ull) : 0u) +                             \
                    (k > 2 ? (u32)((b2 >> L) & 1ull) : 0u);                                                                        \
            have##n = need##n && a1_##n != a0_##n && ((bk >> L) & 1ull);                                                           \
            pos##n = have##n ? a0_##n + rank : (u32)i;                                                                             \
        }
        MC3_RANK(2) MC3_RANK(3) MC3_RANK(4) MC3_RANK(5) MC3_RANK(6)
#undef MC3_RANK
        const bool have0 = true;
        // edge e: owner number o, slot sl
#define MC3_EDGE(e, o, sl)                                                                                                          \
        if (used & (1u << e)) { if (have##o) id##e = table[(int64_t)pos##o * 4 + sl]; else bad = true; }
        MC3_EDGE(0, 0, 0) MC3_EDGE(1, 1, 1) MC3_EDGE(2, 2, 0) MC3_EDGE(3, 0, 1) MC3_EDGE(4, 3, 0) MC3_EDGE(5, 5, 1) MC3_EDGE(6, 6, 0)
        MC3_EDGE(7, 3, 1) MC3_EDGE(8, 0, 2) MC3_EDGE(9, 1, 2) MC3_EDGE(10, 4, 2) MC3_EDGE(11, 2, 2) MC3_EDGE(12, 0, 3)
#undef MC3_EDGE
#undef MC3_USED
    }
    int degen = 0;
    for (u32 w0 = 0; w0 < total; w0 += MC3_FWIN) {
        for (int tI = 0; tI < ntri; tI++) {
            const u32 sl = tloc + (u32)tI - w0;                  // (unsigned: slots below the window wrap to huge values)
            if (sl >= MC3_FWIN) continue;
            int32_t tv[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int ed = tris[3 * tI + j];
                int32_t v = id0;                                // select chain over scalars
                v = ed == 1 ? id1 : v; v = ed == 2 ? id2 : v; v = ed == 3 ? id3 : v; v = ed == 4 ? id4 : v;
                v = ed == 5 ? id5 : v; v = ed == 6 ? id6 : v; v = ed == 7 ? id7 : v; v = ed == 8 ? id8 : v;
                v = ed == 9 ? id9 : v; v = ed == 10 ? id10 : v; v = ed == 11 ? id11 : v; v = ed == 12 ? id12 : v;
                tv[j] = v;
            }
            sf[3 * sl] = tv[2]; sf[3 * sl + 1] = tv[1]; sf[3 * sl + 2] = tv[0];     // np.fliplr(faces) of the skimage wrapper
            degen += (tv[0] == tv[1] || tv[1] == tv[2] || tv[0] == tv[2]) ? 1 : 0;
        }
        __syncthreads();
        const u32 n = 3u * (total - w0 < MC3_FWIN ? total - w0 : (u32)MC3_FWIN);
        int64_t *dst = faces + 3 * (int64_t)(fbase + w0);
        if constexpr (SLAB) {
            int64_t off_me = 0;
            for (int r = 0; r < sm.rank; ++r) off_me += sm.gathered[8 * r];
            const int64_t off_next = off_me + sm.gathered[8 * sm.rank];
            const bool ok = tot[1] <= (u64)sm.cap_v && tot[7] <= tot[1];
            const int64_t k = ok ? (int64_t)(tot[1] - tot[7]) : 0;
            for (u32 e = threadIdx.x; e < n; e += MC3_BLK) {
                const int64_t row = (int64_t)sf[e];
                const int64_t j = row - k;
                dst[e] = row < k ? row + off_me : off_next + ((sm.ids_next && j < sm.cap_top) ? (int64_t)sm.ids_next[j] : 0);
            }
        } else {
            for (u32 e = threadIdx.x; e < n; e += MC3_BLK) dst[e] = (int64_t)sf[e];
        }
        __syncthreads();
    }
    if (degen) atomicAdd((unsigned long long *)&tot[5], (unsigned long long)degen);
    if (bad) atomicAdd((unsigned long long *)&tot[6], 1ull);
}

TOMO_API int tomo_mc3_eval(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                           const unsigned long long *vox_key, int64_t cap, const unsigned long long *tot, int z_offset,
                           uint32_t *vox_loc, int32_t *vox_til, uint8_t *vox_flags, uint16_t *vox_used, float *vox_f3,
                           float *vox_c3, uint32_t *blk3, void *stream)
{
    McGrid g;
    int rc = make_grid(g, field, Nz, Ny, Nx, pitch, xorg, iso);
    if (rc) return rc;
    if (!vox_key || !tot || !vox_loc || !vox_til || !vox_flags || !vox_used || !vox_f3 || !vox_c3 || !blk3 || cap <= 0) return TOMO_E_ARG;
    const int64_t nblk = ceil_div64(cap, MC3_BLK);
    if (nblk > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc3_eval_kernel, dim3((unsigned)nblk), dim3(MC3_BLK), 0, (hipStream_t)stream, field, g, (const u64 *)vox_key,
                       cap, (const u64 *)tot, z_offset, vox_loc, vox_til, vox_flags, vox_used, vox_f3, vox_c3, blk3, nblk);
    return tomo_status();
}

TOMO_API int64_t tomo_mc3_sort_segments(int Nz, int Ny) { return (int64_t)(TOMO_SORT_NB(Ny) + 1) * Nz; }
TOMO_API int64_t tomo_mc3_slice_table_words(int Nz, int Ny) { return 2 * ((int64_t)Nz + 1) + tomo_mc3_sort_segments(Nz, Ny) + 1 + 3 + 8; }

TOMO_API int tomo_mc3_scan(int Nz, int Ny, int Nx, int xorg, const uint32_t *seg_aoff, const uint32_t *vox_loc, int64_t cap,
                           uint32_t *blk3, uint32_t *slice_tab, unsigned long long *tot, int64_t cap_v, int64_t cap_f,
                           void *stream)
{
    if (Nz < 2 || Ny < 2 || Nx < 2 || !seg_aoff || !vox_loc || !blk3 || !slice_tab || !tot || cap <= 0) return TOMO_E_ARG;
    if (cap_v >= 0x7fffffffll || cap_f >= 0x7fffffffll) return TOMO_E_SIZE;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    hipLaunchKernelGGL(mc3_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, seg_aoff, vox_loc, cap, blk3,
                       ceil_div64(cap, MC3_BLK), slice_tab, (u64 *)tot, cap_v, cap_f);
    const int64_t nseg = tomo_mc3_sort_segments(Nz, Ny);
    if (nseg + 1 > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc3_bands_kernel, dim3((unsigned)ceil_div64(nseg + 1, 256)), dim3(256), 0, (hipStream_t)stream, g, seg_aoff,
                       vox_loc, (const u32 *)blk3, ceil_div64(cap, MC3_BLK), slice_tab, (const u64 *)tot);
    return tomo_status();
}

TOMO_API int tomo_mc3_vertices(int Nz, int Ny, int Nx, int xorg, const unsigned long long *vox_key, int64_t cap,
                               const unsigned long long *tot, const uint32_t *vox_loc, const uint8_t *vox_flags,
                               const float *vox_f3, const float *vox_c3, const uint32_t *blk3, const uint32_t *slice_tab,
                               int z_offset, int shift, const double *cum, int64_t ncum, const double *adj, int64_t nadj,
                               float mm_y, float mm_x, float *vrec, uint32_t *keys, uint32_t *idx, void *stream)
{
    if (Nz < 2 || Ny < 2 || Nx < 2 || !vox_key || !tot || !vox_loc || !vox_flags || !vox_f3 || !vox_c3 || !blk3 || !slice_tab ||
        !vrec || !keys || cap <= 0 || (nadj > 0 && (!cum || !adj || ncum != nadj + 1)))
        return TOMO_E_ARG;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    Mc3Final fin; fin.cum = cum; fin.adj = adj; fin.ncum = ncum; fin.nadj = nadj; fin.shift = shift; fin.mm_y = mm_y; fin.mm_x = mm_x;
    const int64_t nblk = ceil_div64(cap, MC3_BLK);
    hipLaunchKernelGGL(mc3_vertices_kernel, dim3((unsigned)nblk), dim3(MC3_BLK), 0, (hipStream_t)stream, g, (const u64 *)vox_key, cap,
                       (const u64 *)tot, vox_loc, vox_flags, vox_f3, vox_c3, blk3, nblk, slice_tab, z_offset, fin, (float4 *)vrec,
                       keys, idx);
    return tomo_status();
}

TOMO_API int tomo_mc3_faces(int Nz, int Ny, int Nx, int xorg, const unsigned long long *vox_key, int64_t cap,
                            unsigned long long *tot, const unsigned long long *seg_act, const uint32_t *seg_aoff,
                            const uint32_t *vox_loc, const int32_t *vox_til, const uint16_t *vox_used, const uint32_t *blk3,
                            const int32_t *table, int64_t *faces, int64_t cap_f, void *stream)
{
    if (Nz < 2 || Ny < 2 || Nx < 2 || !vox_key || !tot || !seg_act || !seg_aoff || !vox_loc || !vox_til || !vox_used || !blk3 || !table || !faces ||
        cap <= 0 || cap_f < 0)
        return TOMO_E_ARG;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    const int64_t nblk = ceil_div64(cap, MC3_BLK);
    hipLaunchKernelGGL(mc3_faces_kernel<false>, dim3((unsigned)nblk), dim3(MC3_BLK), 0, (hipStream_t)stream, g, (const u64 *)vox_key, cap,
                       (u64 *)tot, (const u64 *)seg_act, seg_aoff, vox_loc, vox_til, vox_used, blk3, nblk, table, faces, cap_f,
                       Mc3SlabMap{nullptr, nullptr, 0, 0, 0});
    return tomo_status();
}

TOMO_API int tomo_mc3_faces_slab(int Nz, int Ny, int Nx, int xorg, const unsigned long long *vox_key, int64_t cap,
                                 unsigned long long *tot, const unsigned long long *seg_act, const uint32_t *seg_aoff,
                                 const uint32_t *vox_loc, const int32_t *vox_til, const uint16_t *vox_used, const uint32_t *blk3,
                                 const int32_t *table, int64_t *faces, int64_t cap_f, const int64_t *gathered, int rank, int world,
                                 const int32_t *ids_next, int64_t cap_top, int64_t cap_v, void *stream)
{
    if (Nz < 2 || Ny < 2 || Nx < 2 || !vox_key || !tot || !seg_act || !seg_aoff || !vox_loc || !vox_til || !vox_used || !blk3 || !table || !faces ||
        cap <= 0 || cap_f < 0 || !gathered || rank < 0 || rank >= world || cap_v < 1 || cap_top < 0)
        return TOMO_E_ARG;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = 0; g.xorg = xorg; g.iso = 0.0;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx, xorg);
    const int64_t nblk = ceil_div64(cap, MC3_BLK);
    hipLaunchKernelGGL(mc3_faces_kernel<true>, dim3((unsigned)nblk), dim3(MC3_BLK), 0, (hipStream_t)stream, g, (const u64 *)vox_key, cap,
                       (u64 *)tot, (const u64 *)seg_act, seg_aoff, vox_loc, vox_til, vox_used, blk3, nblk, table, faces, cap_f,
                       Mc3SlabMap{gathered, ids_next, cap_top, cap_v, rank});
    return tomo_status();
}
