// mc.hip -- Lewiner marching cubes on gfx950 as count -> scan -> emit.
//
// Replaces skimage.measure.marching_cubes(volume, level=0.5) as called at surface_extractor.py:55
// (Lewiner MC33, step 1, allow_degenerate=True).  The serial original walks cells z->y->x and numbers
// vertices by first touch through two "face layers"; here
//   * a wave owns one SEGMENT = 256 consecutive voxels of a row (4 per lane), in scan order;
//   * a vertex is identified by the edge it sits on: key = (row << 22 | X << 2 | slot), row = Z*Ny+Y,
//     slot 0/1/2 = x/y/z edge whose lower corner is the voxel, 3 = cell-centre vertex.  An edge vertex
//     exists iff the field changes side of the iso level along the edge (every tiling uses exactly the
//     bichromatic edges of its cube), a centre vertex iff the cell's tiling row contains a 12;
//   * pass 1 counts triangles and vertices per segment (wave reduction, no atomics),
//     a three-kernel exclusive scan turns the counts into output offsets and compacts the list of
//     non-empty segments, pass 2 re-evaluates only those segments and writes vertices and triangles at
//     their final positions: face order == the reference's cell scan order, LUT order inside a cell.
// Memory-bound: pass 1 streams the field once (4 B/voxel; the y+1 / z+1 neighbour rows come from L2/MALL).
#include "tomo_common.h"

#define MC_LUT_QUAL __device__ const
#define MC_FN __device__ static inline
#include "mc_cell.h"

#define SEG 256
#define KEY_XBITS 20

struct McGrid {
    int Nz, Ny, Nx;
    int64_t pitch;
    int xorg;
    int segs_per_row;
    double iso;
};

typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

// Loads the 4 voxels of this lane plus the one to the right, for rows (Z,Y),(Z,Y+1),(Z+1,Y),(Z+1,Y+1).
// Out-of-range neighbours are replaced by clamped coordinates (=> no sign change across the border).
struct Rows4 {
    float a[5], b[5], c[5], d[5];   // a: (Z,Y)  b: (Z,Y+1)  c: (Z+1,Y)  d: (Z+1,Y+1)
};

__device__ static inline void load_row5(const float *__restrict__ row, int X0, int Nx, int lane, float *out)
{
    // row points at padded column 0
    if (X0 + 3 < Nx) {
        float4u v = *(const float4u *)(row + X0);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) { int X = X0 + k < Nx ? X0 + k : Nx - 1; out[k] = X0 < Nx ? row[X] : 0.0f; }
    }
    float nxt = dpp_from_next_f32(out[0]);
    if (lane == 63 || X0 + 4 >= Nx) {
        int X = X0 + 4 < Nx ? X0 + 4 : Nx - 1;
        nxt = X0 < Nx ? row[X] : 0.0f;
    }
    out[4] = nxt;
}

__device__ static inline void load_rows(const float *__restrict__ field, const McGrid &g, int Z, int Y, int X0, int lane,
                                        Rows4 &r)
{
    int Y1 = Y + 1 < g.Ny ? Y + 1 : g.Ny - 1;
    int Z1 = Z + 1 < g.Nz ? Z + 1 : g.Nz - 1;
    const float *pa = field + ((int64_t)Z * g.Ny + Y) * g.pitch + g.xorg;
    const float *pb = field + ((int64_t)Z * g.Ny + Y1) * g.pitch + g.xorg;
    const float *pc = field + ((int64_t)Z1 * g.Ny + Y) * g.pitch + g.xorg;
    const float *pd = field + ((int64_t)Z1 * g.Ny + Y1) * g.pitch + g.xorg;
    load_row5(pa, X0, g.Nx, lane, r.a);
    load_row5(pb, X0, g.Nx, lane, r.b);
    load_row5(pc, X0, g.Nx, lane, r.c);
    load_row5(pd, X0, g.Nx, lane, r.d);
}

// per-voxel evaluation ----------------------------------------------------------------------
struct VoxelEval {
    int ntri;
    int flags;                // bit0 x-edge, bit1 y-edge, bit2 z-edge, bit3 centre
    const signed char *tris;
    double v[8];
};

__device__ static inline void eval_voxel(const Rows4 &r, int k, double iso, bool cell_ok, bool vox_ok, VoxelEval &e)
{
    e.ntri = 0; e.flags = 0; e.tris = LUT_TILING1;
    if (!vox_ok) return;
    e.v[0] = (double)r.a[k] - iso;     e.v[1] = (double)r.a[k + 1] - iso;
    e.v[2] = (double)r.b[k + 1] - iso; e.v[3] = (double)r.b[k] - iso;
    e.v[4] = (double)r.c[k] - iso;     e.v[5] = (double)r.c[k + 1] - iso;
    e.v[6] = (double)r.d[k + 1] - iso; e.v[7] = (double)r.d[k] - iso;
    int index = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) index |= (e.v[i] > 0.0 ? 1 : 0) << i;
    int s0 = index & 1;
    e.flags = (s0 != ((index >> 1) & 1) ? 1 : 0) | (s0 != ((index >> 3) & 1) ? 2 : 0) | (s0 != ((index >> 4) & 1) ? 4 : 0);
    if (cell_ok && index != 0 && index != 255) {
        McTiling t = mc_cell_tiling(e.v, index);
        e.ntri = t.ntri; e.tris = t.tris;
        if (t.centre) e.flags |= 8;
    }
}

// cheap reject: do the 20 values of this lane straddle the iso level at all?
__device__ static inline bool lane_has_crossing(const Rows4 &r, double iso)
{
    int hi = 0, lo = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        hi |= ((double)r.a[k] > iso) | ((double)r.b[k] > iso) | ((double)r.c[k] > iso) | ((double)r.d[k] > iso);
        lo |= !((double)r.a[k] > iso) | !((double)r.b[k] > iso) | !((double)r.c[k] > iso) | !((double)r.d[k] > iso);
    }
    return hi && lo;
}

// ------------------------------------------------------------------------------------------ pass 1
__global__ __launch_bounds__(256) void mc_count_kernel(const float *__restrict__ field, const McGrid g, int64_t nseg,
                                                       u32 *__restrict__ seg_counts)
{
    const int lane = threadIdx.x & 63;
    int64_t seg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    int s = (int)(seg % g.segs_per_row);
    int64_t row = seg / g.segs_per_row;
    int Y = (int)(row % g.Ny), Z = (int)(row / g.Ny);
    int X0 = s * SEG + lane * 4;
    Rows4 r;
    load_rows(field, g, Z, Y, X0, lane, r);
    u32 nt = 0, nv = 0;
    if (lane_has_crossing(r, g.iso)) {
        bool zy_ok = (Y + 1 < g.Ny) && (Z + 1 < g.Nz);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            VoxelEval e;
            int X = X0 + k;
            eval_voxel(r, k, g.iso, zy_ok && (X + 1 < g.Nx), X < g.Nx, e);
            nt += (u32)e.ntri;
            nv += (u32)__popc(e.flags);
        }
    }
    nt = wave_sum(nt);
    nv = wave_sum(nv);
    if (lane == 0) seg_counts[seg] = (nt << 16) | nv;
}

TOMO_API int tomo_mc_count(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                           uint32_t *seg_counts, void *stream)
{
    if (!field || !seg_counts || Nz < 2 || Ny < 2 || Nx < 2 || pitch < Nx + xorg) return TOMO_E_ARG;
    if (Nx >= (1 << KEY_XBITS)) return TOMO_E_SIZE;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = pitch; g.xorg = xorg; g.iso = iso;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx);
    int64_t nseg = (int64_t)Nz * Ny * g.segs_per_row;
    int64_t blocks = ceil_div64(nseg, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_count_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, g, nseg,
                       seg_counts);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ scan
// Exclusive scan of (nvert, ntri, active) over the segments: block sums -> one-block scan -> apply.
#define SCAN_ITEMS 16
#define SCAN_BLOCK (256 * SCAN_ITEMS)

struct Sum3 { u64 v, f, a; };

__global__ __launch_bounds__(256) void scan_reduce_kernel(const u32 *__restrict__ counts, int64_t nseg,
                                                          u64 *__restrict__ bsum)
{
    __shared__ u64 sv[4], sf[4], sa[4];
    int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    u64 v = 0, f = 0, a = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int64_t i = base + k;
        u32 c = i < nseg ? counts[i] : 0u;
        v += c & 0xffffu; f += c >> 16; a += c != 0u;
    }
    v = wave_sum64(v); f = wave_sum64(f); a = wave_sum64(a);
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = v; sf[w] = f; sa[w] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[3 * (int64_t)blockIdx.x + 0] = sv[0] + sv[1] + sv[2] + sv[3];
        bsum[3 * (int64_t)blockIdx.x + 1] = sf[0] + sf[1] + sf[2] + sf[3];
        bsum[3 * (int64_t)blockIdx.x + 2] = sa[0] + sa[1] + sa[2] + sa[3];
    }
}

// single block: exclusive scan of the block sums in place, totals out
__global__ __launch_bounds__(256) void scan_blocks_kernel(u64 *__restrict__ bsum, int64_t nblocks, u64 *__restrict__ totals)
{
    __shared__ u64 carry[3];
    __shared__ u64 wsum[4][3];
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t start = 0; start < nblocks; start += 256) {
        int64_t i = start + threadIdx.x;
        u64 x[3];
#pragma unroll
        for (int c = 0; c < 3; c++) x[c] = i < nblocks ? bsum[3 * i + c] : 0ull;
        u64 inc[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            u64 v = x[c];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                u64 o = __shfl_up(v, d, 64);
                if (lane >= d) v += o;
            }
            inc[c] = v;
            if (lane == 63) wsum[w][c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; c++) {
            u64 off = carry[c];
            for (int j = 0; j < w; j++) off += wsum[j][c];
            if (i < nblocks) bsum[3 * i + c] = off + inc[c] - x[c];
        }
        __syncthreads();
        if (threadIdx.x < 3) carry[threadIdx.x] += wsum[0][threadIdx.x] + wsum[1][threadIdx.x] + wsum[2][threadIdx.x] + wsum[3][threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x < 3) totals[threadIdx.x] = carry[threadIdx.x];
    if (threadIdx.x == 3) totals[3] = 0;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const u32 *__restrict__ counts, int64_t nseg,
                                                         const u64 *__restrict__ bsum, u32 *__restrict__ seg_voff,
                                                         u32 *__restrict__ seg_foff, u32 *__restrict__ active_ids)
{
    __shared__ u64 wv[4], wf[4], wa[4];
    int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    u32 c[SCAN_ITEMS];
    u64 v = 0, f = 0, a = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int64_t i = base + k;
        c[k] = i < nseg ? counts[i] : 0u;
        v += c[k] & 0xffffu; f += c[k] >> 16; a += c[k] != 0u;
    }
    // exclusive scan of the per-thread sums across the block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    u64 iv = v, jf = f, ia = a;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 o1 = __shfl_up(iv, d, 64), o2 = __shfl_up(jf, d, 64), o3 = __shfl_up(ia, d, 64);
        if (lane >= d) { iv += o1; jf += o2; ia += o3; }
    }
    if (lane == 63) { wv[w] = iv; wf[w] = jf; wa[w] = ia; }
    __syncthreads();
    u64 ov = bsum[3 * (int64_t)blockIdx.x + 0] + iv - v, of = bsum[3 * (int64_t)blockIdx.x + 1] + jf - f,
        oa = bsum[3 * (int64_t)blockIdx.x + 2] + ia - a;
    for (int j = 0; j < w; j++) { ov += wv[j]; of += wf[j]; oa += wa[j]; }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int64_t i = base + k;
        if (i < nseg) {
            seg_voff[i] = (u32)ov; seg_foff[i] = (u32)of;
            if (c[k]) active_ids[oa] = (u32)i;
        }
        ov += c[k] & 0xffffu; of += c[k] >> 16; oa += c[k] != 0u;
    }
    // the thread that owns the last segment writes the end sentinels
    if (base <= nseg - 1 && nseg - 1 < base + SCAN_ITEMS) { seg_voff[nseg] = (u32)ov; seg_foff[nseg] = (u32)of; }
}

TOMO_API int64_t tomo_mc_scan_workspace_bytes(int64_t nseg)
{
    return (ceil_div64(nseg, SCAN_BLOCK) * 3 + 8) * (int64_t)sizeof(u64);
}

TOMO_API int tomo_mc_scan(const uint32_t *seg_counts, int64_t nseg, uint32_t *seg_voff, uint32_t *seg_foff,
                          uint32_t *active_ids, unsigned long long *totals, void *workspace, int64_t workspace_bytes,
                          void *stream)
{
    if (!seg_counts || !seg_voff || !seg_foff || !active_ids || !totals || !workspace || nseg <= 0) return TOMO_E_ARG;
    if (nseg >= 0xffffffffll) return TOMO_E_SIZE;
    if (workspace_bytes < tomo_mc_scan_workspace_bytes(nseg)) return TOMO_E_WORKSPACE;
    int64_t nblocks = ceil_div64(nseg, SCAN_BLOCK);
    u64 *bsum = (u64 *)workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, seg_counts, nseg, bsum);
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(256), 0, s, bsum, nblocks, (u64 *)totals);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, seg_counts, nseg, (const u64 *)bsum,
                       seg_voff, seg_foff, active_ids);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ pass 2
__device__ static inline u64 make_key(int64_t row, int X, int slot)
{
    return ((u64)row << (KEY_XBITS + 2)) | ((u64)(u32)X << 2) | (u64)slot;
}

__global__ __launch_bounds__(256) void mc_emit_kernel(const float *__restrict__ field, const McGrid g,
                                                      const u32 *__restrict__ seg_voff, const u32 *__restrict__ seg_foff,
                                                      const u32 *__restrict__ active_ids, int64_t n_active,
                                                      u64 *__restrict__ vkey, float *__restrict__ vpos,
                                                      u64 *__restrict__ fkey)
{
    const int lane = threadIdx.x & 63;
    int64_t ai = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ai >= n_active) return;
    int64_t seg = active_ids[ai];
    int s = (int)(seg % g.segs_per_row);
    int64_t row = seg / g.segs_per_row;
    int Y = (int)(row % g.Ny), Z = (int)(row / g.Ny);
    int X0 = s * SEG + lane * 4;
    Rows4 r;
    load_rows(field, g, Z, Y, X0, lane, r);
    const bool zy_ok = (Y + 1 < g.Ny) && (Z + 1 < g.Nz);
    const bool crossing = lane_has_crossing(r, g.iso);
    // first sweep: this lane's totals
    u32 nt = 0, nv = 0;
    if (crossing) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            VoxelEval e;
            int X = X0 + k;
            eval_voxel(r, k, g.iso, zy_ok && (X + 1 < g.Nx), X < g.Nx, e);
            nt += (u32)e.ntri; nv += (u32)__popc(e.flags);
        }
    }
    u32 voff = seg_voff[seg] + wave_inclusive_scan(nv) - nv;
    u32 foff = seg_foff[seg] + wave_inclusive_scan(nt) - nt;
    if (!crossing) return;
    // second sweep: write
    const int64_t rowY = (int64_t)g.Ny;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        VoxelEval e;
        int X = X0 + k;
        eval_voxel(r, k, g.iso, zy_ok && (X + 1 < g.Nx), X < g.Nx, e);
        if (e.flags & 1) {
            vkey[voff] = make_key(row, X, 0);
            float *p = vpos + 3 * (int64_t)voff;
            p[0] = (float)Z; p[1] = (float)Y; p[2] = (float)((double)X + mc_edge_offset(e.v[0], e.v[1]));
            voff++;
        }
        if (e.flags & 2) {
            vkey[voff] = make_key(row, X, 1);
            float *p = vpos + 3 * (int64_t)voff;
            p[0] = (float)Z; p[1] = (float)((double)Y + mc_edge_offset(e.v[0], e.v[3])); p[2] = (float)X;
            voff++;
        }
        if (e.flags & 4) {
            vkey[voff] = make_key(row, X, 2);
            float *p = vpos + 3 * (int64_t)voff;
            p[0] = (float)((double)Z + mc_edge_offset(e.v[0], e.v[4])); p[1] = (float)Y; p[2] = (float)X;
            voff++;
        }
        if (e.flags & 8) {
            double ox, oy, oz;
            mc_centre_offset(e.v, &ox, &oy, &oz);
            vkey[voff] = make_key(row, X, 3);
            float *p = vpos + 3 * (int64_t)voff;
            p[0] = (float)((double)Z + oz); p[1] = (float)((double)Y + oy); p[2] = (float)((double)X + ox);
            voff++;
        }
        for (int tI = 0; tI < e.ntri; tI++) {
            u64 kk[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                int ed = e.tris[3 * tI + j];
                // owner voxel of the edge and its slot
                int dx = (ed == 1 || ed == 5 || ed == 9 || ed == 10) ? 1 : 0;
                int dy = (ed == 2 || ed == 6 || ed == 10 || ed == 11) ? 1 : 0;
                int dz = (ed >= 4 && ed <= 7) ? 1 : 0;
                int slot = ed == 12 ? 3 : (ed >= 8 ? 2 : (ed & 1));
                kk[j] = make_key(row + dy + dz * rowY, X + dx, slot);
            }
            u64 *fp = fkey + 3 * (int64_t)foff;
            fp[0] = kk[2]; fp[1] = kk[1]; fp[2] = kk[0];   // np.fliplr(faces) of the wrapper
            foff++;
        }
    }
}

TOMO_API int tomo_mc_emit(const float *field, int Nz, int Ny, int Nx, int64_t pitch, int xorg, double iso,
                          const uint32_t *seg_voff, const uint32_t *seg_foff, const uint32_t *active_ids,
                          int64_t n_active, unsigned long long *vkey, float *vpos, unsigned long long *fkey,
                          void *stream)
{
    if (!field || !seg_voff || !seg_foff || !active_ids || !vkey || !vpos || !fkey || Nz < 2 || Ny < 2 || Nx < 2)
        return TOMO_E_ARG;
    if (n_active <= 0) return TOMO_OK;
    McGrid g; g.Nz = Nz; g.Ny = Ny; g.Nx = Nx; g.pitch = pitch; g.xorg = xorg; g.iso = iso;
    g.segs_per_row = (int)tomo_mc_segments_per_row(Nx);
    int64_t blocks = ceil_div64(n_active, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(mc_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, field, g, seg_voff,
                       seg_foff, active_ids, n_active, (u64 *)vkey, vpos, (u64 *)fkey);
    return tomo_status();
}
