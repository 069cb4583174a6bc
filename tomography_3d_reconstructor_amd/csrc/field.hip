// field.hip -- the scalar field ("SDF") kernel: bit-packed volume -> float32 field that marching cubes
// reads.  Replaces surface_extractor.py:43-53 (pad, astype(float64), scipy gaussian_filter sigma=0.5)
// plus the float32 cast of skimage/measure/_marching_cubes_lewiner.py:292 in ONE streaming pass:
// the reference's 8 B/voxel float64 intermediates are never materialised.
//
// Arithmetic contract (bit-exact with SciPy 1.7.1, pinned by tests/golden): three separable 5-tap
// float64 passes, axis 0 (z), 1 (y), 2 (x), mode 'reflect', each output
//     t = x[0]*w0;  t += (x[-2] + x[+2])*w2;  t += (x[-1] + x[+1])*w1;
// (symmetric branch of NI_Correlate1D), no FMA contraction (-ffp-contract=off), then (float)t.
//
// Kernel structure (memory-bound: 1 bit in, 4 B out per padded voxel) -- "tile sparse":
//   * the field row is cut into TILES of 32 float columns (one 128-byte line) x 16 rows; float column c
//     of a row holds the value of extended bit c + 4 (the extended bit volume carries every border rule, so
//     there is not a single boundary branch here: pad columns, reflected columns and the unused columns of
//     the pitch are ordinary outputs);
//   * a block owns 4 consecutive slices x 16 rows x (up to 64) tiles.  It stages its 8 x 20 input rows of
//     bits in LDS once (each bit is fetched 2x instead of 5x), reduces them to per-(slice, word) OR / AND,
//     and classifies every tile by its 5 x 20 x 36-bit neighbourhood: all zero -> the field is 0, all one ->
//     the field is the interior constant (computed with the same operation sequence), else MIXED;
//   * constant tiles (94 % of an ellipsoid volume, every tile away from the surface of any volume) are pure
//     aligned float4 store loops: 1 KiB per wave instruction, whole 128-byte lines only;
//   * the block's mixed tiles are pooled and computed six at a time per wave, ten lanes per tile (eight
//     output lanes of four columns + one halo lane on each side): pass 1 (z) has binary input, so it is an
//     18-entry LDS LUT indexed by a 4-voxel SWAR code word; pass 2 (y) marches down the rows with a 5-row
//     window in registers; pass 3 (x) takes the neighbours' pass-2 values by DPP wave shifts.  No barrier
//     and no LDS exchange inside the row loop, full lanes for the float64 work;
//   * the marching-cubes sign records ([field > 0.5], one bit per voxel) are a by-product: bytes of mixed
//     tiles through LDS, bytes of constant tiles from the tile class, one 512-byte store per wave.
#include <stdlib.h>
#include "tomo_common.h"

#define FW0 0x1.92b965ef5aaefp-1    // exp(-2 x^2)/sum for x = 0, +-1, +-2 as produced by the pinned
#define FW1 0x1.b405b9842b206p-4    // oracle environment (SciPy 1.7.1 / NumPy 1.26.4)
#define FW2 0x1.14aebe6a24088p-12

#define FT_ROWS 16                  // rows of a tile = rows of a block
#define FT_SROWS (FT_ROWS + 4)      // staged rows
#define FT_ZG 4                     // slices of a block (2 / 3 / 5 / 6 / 8 measured slower: DESIGN 4.1)
#define FT_SLOTS (FT_ZG + 4)        // staged slices
#define FT_THREADS 256
#define FT_MAXT 64                  // tiles of a block along x (wider rows: several blocks)
#define FT_SBMAX 72                 // max bytes per (slice, row, word k) line of the LDS sign area (>= FT_MAXT + 7)
#define FT_BATCH 24                 // staging loads in flight per thread

struct FieldParams {
    int Nz, Ny, NT;                 // field slices, rows, tiles per row (pitch / 32)
    int EZ, EY, EWX32;              // extended bit volume: slices, rows, 32-bit words per row
    int64_t pitch;
    int nxc, ntr, nzg;              // blocks per row, tile rows, slice groups
    int tp;                         // tile positions (tile index + 7) per block along x: a multiple of 8, <= FT_MAXT
    int nz, ny, nx, pad, SW32;      // (FROM_BITS) the plain bit volume the input is derived from: dims, 32-bit words per row
    u64 *signs;                     // sign records [Nz][S][NyP][4] (may be null)
    unsigned char *gcls;            // class of every group of 16 rows of records [Nz][NyP / 16][S]: 0 / 1 = all bits 0 / 1
                                    // (records not written), 2 = records written
    int S, NyP;
    int SB;                         // bytes per (slice, row, word k) line of the LDS sign area: tiles + 7, rounded up to 8
    // sparse fill (all null = dense): comb[Z][tile row][tile] = 3 where the tile is within reach of the surface (bit 0: a 1,
    // bit 1: a 0 within 3 voxels); list[0 .. *count) = linear ids of the blocks that have anything to do
    const unsigned char *comb;
    const u32 *list, *count;
#ifdef FT_PROFILE
    unsigned long long *prof;
#endif
};

__device__ static inline double tap5(double a, double b, double c, double d, double e)
{   // a..e = x[-2], x[-1], x[0], x[+1], x[+2]
    double t = c * FW0;
    t += (a + e) * FW2;
    t += (b + d) * FW1;
    return t;
}

// dynamic LDS: u32 s_bits[FT_SLOTS][FT_SROWS][WS] | u32 s_ror[FT_SLOTS][WS] | u32 s_rand[FT_SLOTS][WS] |
//              (8-byte aligned) u8 s_sign[FT_ZG][FT_ROWS][4][SB]
extern __shared__ __attribute__((aligned(16))) u32 s_dyn[];

#ifdef FT_PROFILE
// tools/fieldprof.py: per-block phase time stamps (s_memtime) of field_tile_kernel -- a profiling build only
static unsigned long long *g_ft_prof = nullptr;
TOMO_API void tomo_field_profile_buffer(unsigned long long *buf) { g_ft_prof = buf; }
#define FT_STAMP(k) do { if (p.prof && threadIdx.x == 0) p.prof[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FT_STAMP(k) do { } while (0)
#endif

// One 32-bit word of the EXTENDED bit volume (slice ez, row ey, word gw) from the plain bit volume -- what extend_kernel
// (bits.hip) would have stored there: reflect of the padded array along z and y, a funnel shift by 4 + pad bits along x,
// and the four reflected columns X = -2, -1, Nx, Nx+1.  Done inside the staging loop of the kernel (FROM_BITS).
__device__ static inline int reflect_near(int i, int n)
{   // scipy 'reflect' for indices at most 2 outside [0, n) (n = 1: everything reflects onto 0)
    return n < 2 ? 0 : (i < 0 ? -i - 1 : (i >= n ? 2 * n - 1 - i : i));
}

typedef float f4v __attribute__((ext_vector_type(4)));
#define ST4_NT(ptr, a, b, c, d) __builtin_nontemporal_store((f4v){a, b, c, d}, (f4v *)(ptr))
#define ST4_PL(ptr, a, b, c, d) (*(float4 *)(ptr) = make_float4(a, b, c, d))
// measured on the 1024^3 ellipsoid (bench, kernel ms): plain/plain 0.810, const nontemporal 0.796, both nontemporal 0.850
#define ST4C ST4_NT     // constant tiles: whole-line streams nobody reads back soon
#define ST4M ST4_PL     // mixed tiles: marching cubes reads exactly these lines next
template <bool FROM_BITS>
__global__ __launch_bounds__(FT_THREADS) void field_tile_kernel(const u32 *__restrict__ ext32, float *__restrict__ field,
                                                                const FieldParams p)
{
    __shared__ double s_lut[18];
    __shared__ unsigned char s_cls[FT_ZG][FT_SBMAX];
    __shared__ unsigned short s_nlist[FT_ZG * FT_MAXT]; // (sparse fill) constant tiles within reach of the surface
    __shared__ int s_nnear;
    __shared__ unsigned short s_list[FT_ZG * FT_MAXT];
    __shared__ int s_nmixed;
    __shared__ int s_zy[FT_SLOTS + FT_SROWS];       // (FROM_BITS) source slice / row of the staged slots / rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = FT_THREADS / 64;
    if (tid < 18) {
        int c = tid / 9, s2 = (tid / 3) % 3, s1 = tid % 3;
        double t = (double)c * FW0;
        t += (double)s2 * FW2;
        t += (double)s1 * FW1;
        s_lut[tid] = t;
    }
    if (tid == 0) { s_nmixed = 0; s_nnear = 0; }
    FT_STAMP(0);
    // ---- which part of the field is this block's
    if (p.list != nullptr && blockIdx.x >= *p.count) return;
    // consecutive blocks = consecutive tile rows of one slice group (other orders measured slower: DESIGN 4.5)
    const unsigned lin = p.list != nullptr ? p.list[blockIdx.x] : blockIdx.x;
    const int bx = (int)(lin % (unsigned)p.nxc);
    const int tr = (int)((lin / (unsigned)p.nxc) % (unsigned)p.ntr);
    const int zg = (int)(lin / ((unsigned)p.nxc * (unsigned)p.ntr));
    // tiles j0 .. j0+nt-1; chunk boundaries fall on marching-cubes segment boundaries ((j + 7) % 8 == 0) and the
    // chunks of a wide row are equally long (p.tp positions each)
    const int j0 = bx == 0 ? 0 : p.tp * bx - 7;
    const int jend = p.tp * bx + p.tp - 7 < p.NT ? p.tp * bx + p.tp - 7 : p.NT;
    const int nt = jend - j0;
    const int joff = (j0 + 7) & 7;                  // byte position of tile j0 in its segment's record words
    const int WS = nt + 1;                          // staged words per row: ext words j0 .. j0+nt
    const int Y0 = tr * FT_ROWS, Z0 = zg * FT_ZG, SB = p.SB;
    u32 *const s_bits = s_dyn;
    u32 *const s_ror = s_bits + FT_SLOTS * FT_SROWS * WS, *const s_rand = s_ror + FT_SLOTS * WS;
    unsigned char *const s_sign = (unsigned char *)(s_dyn + (((FT_SLOTS * FT_SROWS + 2 * FT_SLOTS) * WS + 1) & ~1));
    const int nrows = Y0 + FT_ROWS <= p.Ny ? FT_ROWS : p.Ny - Y0;
    // ---- sign records of the block: per (slice, segment) 16 rows x 4 words = one 512-byte store per wave.
    //      Byte b of a word = tile 8 (segment) - 7 + b: 0x00 / 0xff for a constant tile, the LDS byte for a mixed one.
    auto write_signs = [&]()
    {
        const int nsegl = (nt + joff + 7) >> 3;
        const int s0 = (j0 + 7) >> 3;
        const int r = lane >> 2, k = lane & 3;
        for (int it = wave; it < FT_ZG * nsegl; it += nwaves) {
            const int z = it / nsegl, sl = it - z * nsegl;
            const int Z = Z0 + z, s = s0 + sl;
            if (Z >= p.Nz || s >= p.S) continue;
            u64 cm = 0ull, mm = 0ull, vm = 0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const int jl = 8 * sl + b - joff;
                const int c = (jl >= 0 && jl < nt) ? s_cls[z][jl] : 3;
                if (c == 1) cm |= 0xffull << (8 * b);
                if (c == 2) mm |= 0xffull << (8 * b);
                if (c != 3) vm |= 0xffull << (8 * b);
            }
            // a group whose tiles are all constant and equal needs no records: marching cubes reads its class instead
            const int gc = mm ? 2 : (cm == 0ull ? 0 : (cm == vm ? 1 : 2));
            if (lane == 0) p.gcls[((int64_t)Z * (p.NyP >> 4) + tr) * p.S + s] = (unsigned char)gc;
            if (gc != 2) continue;
            const u64 v = cm | (*(const u64 *)(s_sign + ((z * FT_ROWS + r) * 4 + k) * SB + 8 * sl) & mm);
            if (r < nrows) p.signs[((((int64_t)Z * p.S + s) * p.NyP) + Y0 + r) * 4 + k] = v;
        }
    };

    // ---- stage the block's input bits [slot][row][word] (slot = ext slice Z0 + slot, row = ext row Y0 + row, word = ext
    //      word j0 + widx); everything outside the extended volume reads as zero
    if (FROM_BITS) {
        // Straight from the plain bit volume.  A thread keeps ONE word column and walks down the (slot, row) pairs, so
        // everything that depends on x -- which source words exist, the reflected columns -- is computed once; per row
        // there is the reflect of (z, y), two unconditional loads (all of a batch in flight together) and a funnel shift.
        const int RG = FT_THREADS / WS;                       // rows handled per sweep of the block
        const int rg = tid / WS, widx = tid - rg * WS;
        const int gw = j0 + widx;
        const bool colv = rg < RG && gw < p.EWX32;
        const bool hv = gw < p.SW32, lv = gw >= 1 && gw - 1 < p.SW32;
        const int sh = 4 + p.pad;                             // ext bit e = data x + 4 + pad
        const int Nxp = p.nx + 2 * p.pad;
        // reflected columns X = -2, -1, Nx, Nx+1 (ext bit X + 4): bit `fsrc` of the 64-bit pair (word gw : word gw-1) is
        // OR-ed in at position `fpos` (the funnel shift leaves zeros there: left of the data and in its tail)
        int fpos[4], fsrc[4];
        u32 fen[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int X = q < 2 ? q - 2 : Nxp + q - 2;
            const int e = X + 4;
            const int x = reflect_near(X, Nxp) - p.pad;        // data column the bit is copied from (outside: the pad ring, 0)
            const bool on = (e >> 5) == gw && x >= 0 && x < p.nx;
            fpos[q] = e & 31;
            fsrc[q] = on ? x - 32 * (gw - 1) : 0;              // 0 .. 63 (e - x <= 7)
            fen[q] = on ? 1u : 0u;
        }
        // source slice / row of every staged slot / row (-1: outside the volume), once per block
        if (tid < FT_SLOTS) {
            const int z = reflect_near(Z0 + tid - 2, p.nz + 2 * p.pad) - p.pad;
            s_zy[tid] = (Z0 + tid < p.EZ && z >= 0 && z < p.nz) ? z : -1;
        } else if (tid < FT_SLOTS + FT_SROWS) {
            const int row = tid - FT_SLOTS;
            const int y = reflect_near(Y0 + row - 2, p.ny + 2 * p.pad) - p.pad;
            s_zy[tid] = (Y0 + row < p.EY && y >= 0 && y < p.ny) ? y : -1;
        }
        __syncthreads();
        const int nrs = FT_SLOTS * FT_SROWS;
        for (int r0 = rg; r0 < nrs; r0 += RG * FT_BATCH) {
            u32 vh[FT_BATCH], vl[FT_BATCH];
            bool rv[FT_BATCH];
#pragma unroll
            for (int b = 0; b < FT_BATCH; b++) {
                const int r = r0 + b * RG;
                const int slot = r / FT_SROWS, row = r - slot * FT_SROWS;
                const int z = r < nrs ? s_zy[slot] : -1, y = r < nrs ? s_zy[FT_SLOTS + row] : -1;
                rv[b] = colv && z >= 0 && y >= 0;
                const int64_t base = rv[b] ? ((int64_t)z * p.ny + y) * p.SW32 : 0;
                vh[b] = ext32[base + ((rv[b] && hv) ? gw : 0)];
                vl[b] = ext32[base + ((rv[b] && lv) ? gw - 1 : 0)];
            }
#pragma unroll
            for (int b = 0; b < FT_BATCH; b++) {
                const int r = r0 + b * RG;
                const u32 hi = (rv[b] && hv) ? vh[b] : 0u, lo = (rv[b] && lv) ? vl[b] : 0u;
                const u64 pair = ((u64)hi << 32) | lo;
                u32 v = (hi << sh) | (lo >> (32 - sh));
#pragma unroll
                for (int q = 0; q < 4; q++) v |= ((u32)(pair >> fsrc[q]) & fen[q]) << fpos[q];
                if (rg < RG && r < nrs) s_bits[r * WS + widx] = v;
            }
        }
    } else {
        const int total = FT_SLOTS * FT_SROWS * WS;
        int widx = tid % WS, rs = tid / WS;             // rs = slot * FT_SROWS + row
        const int dw = FT_THREADS % WS, drs = FT_THREADS / WS;
        for (int base = 0; base < total; base += FT_BATCH * FT_THREADS) {
            u32 v[FT_BATCH];
            int wi = widx, r = rs;
#pragma unroll
            for (int b = 0; b < FT_BATCH; b++) {
                const int slot = r / FT_SROWS, row = r - slot * FT_SROWS;
                const int ez = Z0 + slot, ey = Y0 + row, gw = j0 + wi;
                const bool ok = (base + b * FT_THREADS + tid) < total && ez < p.EZ && ey < p.EY && gw >= 0 && gw < p.EWX32;
                v[b] = ok ? ext32[((int64_t)ez * p.EY + ey) * p.EWX32 + gw] : 0u;
                wi += dw; r += drs;
                if (wi >= WS) { wi -= WS; r++; }
            }
#pragma unroll
            for (int b = 0; b < FT_BATCH; b++) {
                const int idx = base + b * FT_THREADS + tid;
                if (idx < total) s_bits[idx] = v[b];
            }
            widx = wi; rs = r;
        }
    }
    __syncthreads();
    FT_STAMP(1);
    // ---- per (slot, word): OR / AND over the 20 staged rows
    for (int it = tid; it < FT_SLOTS * WS; it += FT_THREADS) {
        const int slot = it / WS, w = it - slot * WS;
        const u32 *bp = s_bits + slot * FT_SROWS * WS + w;
        u32 o = 0u, a = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < FT_SROWS; r++) { u32 v = bp[r * WS]; o |= v; a &= v; }
        s_ror[it] = o; s_rand[it] = a;
    }
    __syncthreads();
    // ---- tile classes: 0 all zero, 1 all one, 2 mixed, 3 no such slice.  Tile j reads ext bits [32j+2, 32j+38)
    //      = bits 2..31 of staged word jl and bits 0..5 of staged word jl+1 (jl = j - j0), 5 slices, 20 rows.
    for (int it = tid; it < FT_ZG * nt; it += FT_THREADS) {
        const int z = it / nt, jl = it - z * nt;
        u32 o = 0u, a = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const int q = (z + k) * WS + jl;
            o |= (s_ror[q] & 0xfffffffcu) | (s_ror[q + 1] & 0x3fu);
            a &= (s_rand[q] | 0x3u) & (s_rand[q + 1] | ~0x3fu);
        }
        int c = o == 0u ? 0 : (a == 0xffffffffu ? 1 : 2);
        if (Z0 + z >= p.Nz) c = 3;
        s_cls[z][jl] = (unsigned char)c;
        if (p.comb != nullptr && c < 2 && p.comb[((int64_t)(Z0 + z) * p.ntr + tr) * p.NT + j0 + jl] == 3)
            s_nlist[atomicAdd(&s_nnear, 1)] = (unsigned short)((z << 8) | jl);      // constant, but within reach of the surface
        if (c == 2) s_list[atomicAdd(&s_nmixed, 1)] = (unsigned short)((z << 8) | jl);
    }
    __syncthreads();
    FT_STAMP(2);

    // constants of the all-ones interior, produced by the same operation sequence
    const double p1one = s_lut[17];
    const double c2 = tap5(p1one, p1one, p1one, p1one, p1one);
    const double c3 = tap5(c2, c2, c2, c2, c2);
    const float c3f = (float)c3;

    // ---- constant tiles: whole-line float4 stores.  The block's part of a slice (16 rows x nt tiles) is swept as a
    //      flat, row-major list of 128-byte lines, 8 lines (1 KiB) per wave instruction, the four waves interleaved:
    //      the block writes each slice as one sequential stream (for rows of <= 57 tiles the region is contiguous).
    if (p.comb != nullptr) {
        // sparse fill: only the constant tiles within reach of the surface (a short list), one tile = two wave stores
        const int ncl = s_nnear;
        for (int ti = wave; ti < ncl; ti += nwaves) {
            const u32 ent = (u32)s_nlist[ti];
            const int z = (int)(ent >> 8), jl = (int)(ent & 0xffu);
            const float kf = s_cls[z][jl] ? c3f : 0.0f;
            float *tb = field + ((int64_t)(Z0 + z) * p.Ny + Y0) * p.pitch + 32 * (j0 + jl) + 4 * (lane & 7);
#pragma unroll
            for (int rr = 0; rr < FT_ROWS; rr += 8) {
                const int row = rr + (lane >> 3);
                if (row < nrows) ST4M(tb + (int64_t)row * p.pitch, kf, kf, kf, kf);
            }
        }
    } else {
        const int nlines = nrows * nt;
        const float inv_nt = 1.0f / (float)nt;
        for (int z = 0; z < FT_ZG && Z0 + z < p.Nz; z++) {
            float *zbase = field + ((int64_t)(Z0 + z) * p.Ny + Y0) * p.pitch + 32 * j0 + 4 * (lane & 7);
            for (int line = 8 * wave + (lane >> 3); line < nlines; line += 8 * nwaves) {
                const int row = (int)(((float)line + 0.5f) * inv_nt);      // exact: line < 1024, nt <= 64
                const int jl = line - row * nt;
                const int c = s_cls[z][jl];
                if (c < 2) {
                    const float kf = c ? c3f : 0.0f;
                    ST4C(zbase + (int64_t)row * p.pitch + 32 * jl, kf, kf, kf, kf);
                }
            }
        }
    }

    // ---- mixed tiles: six per wave pass, ten lanes per tile (lanes 1..8 own the tile's 32 columns, lanes 0 and 9
    //      compute the pass-2 values of the halo columns)
    const int nmixed = s_nmixed;
    FT_STAMP(3);
#ifdef FT_PROFILE
    if (p.prof && tid == 0) p.prof[(size_t)blockIdx.x * 8 + 7] = (unsigned long long)nmixed | ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32);
#endif
    const int SS = FT_SROWS * WS;                   // LDS slot stride
    for (int ch = wave; ch * 6 < nmixed; ch += nwaves) {
        const int g = lane / 10, l = lane - 10 * g;
        const int ti = ch * 6 + g;
        const bool act = lane < 60 && ti < nmixed;
        const u32 ent = act ? (u32)s_list[ti] : 0u;
        const int z = (int)(ent >> 8), jl = (int)(ent & 0xffu);
        // this lane's four columns start at float column 32 (j0+jl) - 4 + 4 l = ext bit 32 (j0+jl) + 4 l
        const u32 *bp = s_bits + z * SS + jl + (l >= 8 ? 1 : 0);
        const int sh = (4 * l) & 31;
        const bool st = act && l >= 1 && l <= 8;
        float *orow = field + ((int64_t)(Z0 + z) * p.Ny + Y0) * p.pitch + 32 * (j0 + jl) + 4 * (l - 1);
        unsigned char *sg = s_sign + ((z * FT_ROWS) * 4 + (l - 1)) * SB + jl + joff;   // lanes l = 1..4 write word k = l-1
        const int bsh = 10 * g + 1;

#define SPREAD(n) ((((n) >> sh) & 0xFu) * 0x00204081u & 0x01010101u)
        // pass-1 code word of staged row yy: 4 bytes, byte k = 9*c + 3*s2 + s1 of column k
#define CODES(yy, dst)                                                                                  \
        {                                                                                               \
            const u32 *rp = bp + (yy) * WS;                                                             \
            u32 n0 = rp[0], n1 = rp[SS], n2 = rp[2 * SS], n3 = rp[3 * SS], n4 = rp[4 * SS];             \
            dst = SPREAD(n2) * 9u + (SPREAD(n0) + SPREAD(n4)) * 3u + (SPREAD(n1) + SPREAD(n3));         \
        }
#define LOOKUP(dst, c)                                    \
        {                                                 \
            dst[0] = s_lut[(c) & 0xffu];                  \
            dst[1] = s_lut[((c) >> 8) & 0xffu];           \
            dst[2] = s_lut[((c) >> 16) & 0xffu];          \
            dst[3] = s_lut[(c) >> 24];                    \
        }
        // one output row r: window rows (A,B,C,D,E) = staged rows r .. r+4
#define STEP(A, B, C, D, E)                                                                             \
        {                                                                                               \
            u32 cE;                                                                                     \
            CODES(r + 4, cE);                                                                           \
            LOOKUP(E, cE);                                                                              \
            const double q0 = tap5(A[0], B[0], C[0], D[0], E[0]);                                       \
            const double q1 = tap5(A[1], B[1], C[1], D[1], E[1]);                                       \
            const double q2 = tap5(A[2], B[2], C[2], D[2], E[2]);                                       \
            const double q3 = tap5(A[3], B[3], C[3], D[3], E[3]);                                       \
            const double L2 = dpp_from_prev_f64(q2), L3 = dpp_from_prev_f64(q3);                        \
            const double R0 = dpp_from_next_f64(q0), R1 = dpp_from_next_f64(q1);                        \
            const float o0 = (float)tap5(L2, L3, q0, q1, q2);                                           \
            const float o1 = (float)tap5(L3, q0, q1, q2, q3);                                           \
            const float o2 = (float)tap5(q0, q1, q2, q3, R0);                                           \
            const float o3 = (float)tap5(q1, q2, q3, R0, R1);                                           \
            if (st && r < nrows) ST4M(orow, o0, o1, o2, o3);                                            \
            orow += p.pitch;                                                                            \
            if (do_signs) {                                                                             \
                const u64 b0 = __ballot(o0 > 0.5f), b1 = __ballot(o1 > 0.5f);                           \
                const u64 b2 = __ballot(o2 > 0.5f), b3 = __ballot(o3 > 0.5f);                           \
                const u64 bk = l == 1 ? b0 : (l == 2 ? b1 : (l == 3 ? b2 : b3));                        \
                if (act && l >= 1 && l <= 4) sg[r * 4 * SB] = (unsigned char)(bk >> bsh);            \
            }                                                                                           \
        }
        const bool do_signs = p.signs != nullptr;
        double wa[4], wb[4], wc[4], wd[4], we[4];
        {
            u32 c;
            CODES(0, c); LOOKUP(wa, c);
            CODES(1, c); LOOKUP(wb, c);
            CODES(2, c); LOOKUP(wc, c);
            CODES(3, c); LOOKUP(wd, c);
        }
        int r = 0;
        while (r < FT_ROWS) {
            STEP(wa, wb, wc, wd, we); if (++r >= FT_ROWS) break;
            STEP(wb, wc, wd, we, wa); if (++r >= FT_ROWS) break;
            STEP(wc, wd, we, wa, wb); if (++r >= FT_ROWS) break;
            STEP(wd, we, wa, wb, wc); if (++r >= FT_ROWS) break;
            STEP(we, wa, wb, wc, wd); ++r;
        }
#undef STEP
#undef LOOKUP
#undef CODES
#undef SPREAD
    }
    FT_STAMP(4);
    if (p.signs == nullptr) return;
    __syncthreads();
    FT_STAMP(5);
    write_signs();
    FT_STAMP(6);
}

// manifold=False: the field is the raw 0/1 volume (surface_extractor.py:46 without the Gaussian).
__global__ __launch_bounds__(256) void field_raw_kernel(const u64 *__restrict__ ext, float *__restrict__ field, int Nz,
                                                        int Ny, int Nx, int pad, int EY, int EWX, int64_t pitch, int xorg)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)Nz * Ny * Nx;
    if (i >= total) return;
    int X = (int)(i % Nx);
    int64_t r = i / Nx;
    int Y = (int)(r % Ny), Z = (int)(r / Ny);
    int e = X + 4;
    u64 w = ext[((int64_t)(Z + 2) * EY + (Y + 2)) * EWX + (e >> 6)];
    field[((int64_t)Z * Ny + Y) * pitch + xorg + X] = (float)((w >> (e & 63)) & 1ull);
}

static size_t fill_params(FieldParams &p, int nz, int ny, int nx, int pad, unsigned long long *signs, unsigned char *gcls,
                          int tp_force = 0)
{
    p.Nz = nz + 2 * pad; p.Ny = ny + 2 * pad;
    const int Nx = nx + 2 * pad;
    p.EZ = (int)tomo_ext_slices(nz, pad);
    p.EY = (int)tomo_ext_rows(ny, pad);
    p.EWX32 = 2 * (int)tomo_ext_words_per_row(nx, pad);
    p.pitch = tomo_field_pitch(nx, pad);
    p.NT = (int)(p.pitch / 32);
    p.nxc = (p.NT + 7 + FT_MAXT - 1) / FT_MAXT;
    p.tp = ((p.NT + 7 + p.nxc - 1) / p.nxc + 7) / 8 * 8;          // balanced chunks, still a multiple of 8 and <= FT_MAXT
    if (tp_force) {                                                // sparse fill: short chunks, so that most blocks see no surface
        p.tp = tp_force;
        p.nxc = (p.NT + 7 + p.tp - 1) / p.tp;
    }
    p.ntr = (p.Ny + FT_ROWS - 1) / FT_ROWS;
    p.nzg = (p.Nz + FT_ZG - 1) / FT_ZG;
    p.S = (int)tomo_mc_segments_per_row(Nx, tomo_field_xorg(pad));
    p.NyP = (int)tomo_sign_rows(p.Ny);
    p.nz = nz; p.ny = ny; p.nx = nx; p.pad = pad;
    p.SW32 = 2 * (int)tomo_words_per_row(nx);
    p.signs = (u64 *)signs;
    p.gcls = gcls;
    p.comb = nullptr; p.list = nullptr; p.count = nullptr;
#ifdef FT_PROFILE
    p.prof = g_ft_prof;
#endif
    int ntmax = p.NT < p.tp ? p.NT : p.tp;           // tiles of the widest block
    int WS = ntmax + 1;
    size_t words = (((size_t)(FT_SLOTS * FT_SROWS + 2 * FT_SLOTS) * WS + 1) & ~(size_t)1);
    p.SB = (ntmax + 7 + 7) & ~7;
    return words * sizeof(u32) + (signs ? (size_t)FT_ZG * FT_ROWS * 4 * p.SB : 0);
}

// ------------------------------------------------------------------------------------------ sparse fill
// The float field is an intermediate: marching cubes reads it only at the corners of ACTIVE cells (and of the voxels that
// own a vertex), i.e. within one voxel of a sign change.  A constant tile whose input is uniform within 3 voxels (2 for
// the Gaussian's radius + 1 for the cell) in every direction can therefore stay unwritten.  span[Z][tile row][tile] says,
// for ONE slice, whether the tile's reach (x and y extended by 3, positions outside the padded array ignored, the pad
// ring = 0) holds a 1 (bit 0) and / or a 0 (bit 1); the field kernel ORs seven slices of it.
#define FSPAN_Z 4                // slices of a block (one wave each)
__global__ __launch_bounds__(64 * FSPAN_Z) void field_span_kernel(const u32 *__restrict__ b32, unsigned char *__restrict__ span,
                                                                  int nz, int ny, int nx, int pad, int SW32, int Nz, int Ny,
                                                                  int Nx, int NT, int ntr)
{
    // one block per (tile row tr, FSPAN_Z slices), one wave per slice: OR / AND of every 32-bit data word column over the
    // data rows of the tile row's reach (one lane per column, the rows' loads in flight together), then one lane per tile
    // looks at its 38 columns
    extern __shared__ u32 s_oa[];                   // [FSPAN_Z][2][C]: OR, AND of data word w = c - 1, c = 0 .. C-1
    const int C = NT + 2;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    u32 *s_o = s_oa + wv * 2 * C, *s_a = s_o + C;
    const int tr = (int)(blockIdx.x % (unsigned)ntr), Z = (int)(blockIdx.x / (unsigned)ntr) * FSPAN_Z + wv;
    const int z = Z - pad;
    const bool zdata = z >= 0 && z < nz;
    // rows of the reach inside the padded array, and the data rows among them
    const int Ya = tr * FT_ROWS - 3 < 0 ? 0 : tr * FT_ROWS - 3, Yb = tr * FT_ROWS + FT_ROWS + 3 < Ny ? tr * FT_ROWS + FT_ROWS + 3 : Ny;
    const int ya = Ya - pad < 0 ? 0 : Ya - pad, yb = Yb - pad < ny ? Yb - pad : ny;      // data rows [ya, yb)
    const bool ring_row = zdata && (Ya < pad || Yb > ny + pad);                           // a pad row within reach
    if (Z < Nz) {
        for (int c = lane; c < C; c += 64) {
            const int w = c - 1;
            u32 o = 0u, a = 0xffffffffu;
            if (zdata && w >= 0 && w < SW32) {
                const u32 *col = b32 + ((int64_t)z * ny) * SW32 + w;
                for (int y0 = ya; y0 < yb; y0 += 12) {
                    u32 v[12];
#pragma unroll
                    for (int k = 0; k < 12; k++) v[k] = y0 + k < yb ? col[(int64_t)(y0 + k) * SW32] : 0u;
#pragma unroll
                    for (int k = 0; k < 12; k++) if (y0 + k < yb) { o |= v[k]; a &= v[k]; }
                }
            } else {
                a = 0u;                              // no such data: handled through the valid / data masks below
            }
            s_o[c] = o; s_a[c] = a;
        }
    }
    __syncthreads();
    if (Z >= Nz) return;
    for (int j = lane; j < NT; j += 64) {
        const int X0 = 32 * j - 3;                              // padded column of window bit 0 (38 columns)
        const int lo_i = X0 < 0 ? -X0 : 0, hi_i = Nx - X0 < 38 ? Nx - X0 : 38;
        u32 f = 0u;
        if (hi_i > lo_i && Yb > Ya) {
            const u64 valid = ((1ull << hi_i) - 1ull) & ~((1ull << lo_i) - 1ull);
            // data columns among the window: x = X0 - pad + i in [0, nx)
            const int xs = X0 - pad;
            const int dlo = -xs > 0 ? -xs : 0, dhi = nx - xs < 38 ? nx - xs : 38;
            const u64 dmask = dhi > dlo ? (((1ull << dhi) - 1ull) & ~((1ull << dlo) - 1ull)) : 0ull;
            u64 WO = 0ull, WA = 0ull;
            if (zdata && yb > ya) {
                const int w0 = xs >> 5, sh = xs & 31;            // arithmetic shift: floor; w0 >= -1
                const int c0 = w0 + 1;                           // LDS index of data word w0
                const u32 o0 = s_o[c0], o1 = c0 + 1 < C ? s_o[c0 + 1] : 0u, o2 = c0 + 2 < C ? s_o[c0 + 2] : 0u;
                const u32 a0 = s_a[c0], a1 = c0 + 1 < C ? s_a[c0 + 1] : 0u, a2 = c0 + 2 < C ? s_a[c0 + 2] : 0u;
                const u64 lo_o = ((u64)o1 << 32) | o0, lo_a = ((u64)a1 << 32) | a0;
                WO = sh ? ((lo_o >> sh) | ((u64)o2 << (64 - sh))) : lo_o;
                WA = sh ? ((lo_a >> sh) | ((u64)a2 << (64 - sh))) : lo_a;
            }
            const bool has_one = (WO & dmask) != 0ull;
            // a zero: a pad slice / no data rows at all, a pad row or pad column within reach, or a zero data bit
            const bool has_zero = !zdata || yb <= ya || ring_row || (valid & ~dmask) != 0ull || ((~WA) & dmask) != 0ull;
            f = (has_one ? 1u : 0u) | (has_zero ? 2u : 0u);
        }
        span[((int64_t)Z * ntr + tr) * NT + j] = (unsigned char)f;
    }
}

// comb = OR of the span bytes of slices Z-3 .. Z+3: 3 <=> the tile is within reach of the surface
__global__ __launch_bounds__(256) void field_comb_kernel(const unsigned char *__restrict__ span, unsigned char *__restrict__ comb,
                                                         int Nz, int64_t per_slice)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)Nz * per_slice) return;
    const int Z = (int)(i / per_slice);
    u32 c = 0u;
#pragma unroll
    for (int k = -3; k <= 3; k++)
        if (Z + k >= 0 && Z + k < Nz) c |= span[i + (int64_t)k * per_slice];
    comb[i] = (unsigned char)c;
}

// One thread per block of the field kernel's (sparse) grid: 4 slices x one marching-cubes segment (8 tiles).  A block with
// a tile within reach of the surface -- or with constant tiles of both kinds, whose records must be written -- goes on
// the work list; every other block is done here: its groups are constant, only their class is recorded.
__global__ __launch_bounds__(256) void field_worklist_kernel(const unsigned char *__restrict__ comb, FieldParams p,
                                                             u32 *__restrict__ list, u32 *__restrict__ count)
{
    const int64_t lin = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lin >= (int64_t)p.nxc * p.ntr * p.nzg) return;
    const int bx = (int)(lin % p.nxc), tr = (int)((lin / p.nxc) % p.ntr), zg = (int)(lin / ((int64_t)p.nxc * p.ntr));
    const int j0 = bx == 0 ? 0 : p.tp * bx - 7;
    const int jend = p.tp * bx + p.tp - 7 < p.NT ? p.tp * bx + p.tp - 7 : p.NT;
    // per slice and segment of the block: which kinds of constant tiles it has (1: ones, 2: zeros); 3 in reach => near
    const int joff = (j0 + 7) & 7, nsegl = (jend - j0 + joff + 7) >> 3, s0 = (j0 + 7) >> 3;
    bool near = false;
    for (int z = 0; z < FT_ZG && !near; z++) {
        const int Z = zg * FT_ZG + z;
        if (Z >= p.Nz) continue;
        const unsigned char *row = comb + ((int64_t)Z * p.ntr + tr) * p.NT;
        for (int sl = 0; sl < nsegl; sl++) {
            u32 kinds = 0u;
            for (int b = 0; b < 8; b++) {
                const int j = j0 + 8 * sl + b - joff;
                if (j < j0 || j >= jend) continue;
                const u32 c = row[j];
                near |= c == 3u;
                kinds |= c == 1u ? 1u : 2u;
            }
            near |= kinds == 3u;
        }
    }
    if (near) { list[atomicAdd(count, 1u)] = (u32)lin; return; }
    if (p.gcls == nullptr) return;
    for (int z = 0; z < FT_ZG; z++) {
        const int Z = zg * FT_ZG + z;
        if (Z >= p.Nz) continue;
        const unsigned char *row = comb + ((int64_t)Z * p.ntr + tr) * p.NT;
        for (int sl = 0; sl < nsegl; sl++) {
            if (s0 + sl >= p.S) continue;
            int j = j0 + 8 * sl - joff;
            if (j < j0) j = j0;
            p.gcls[((int64_t)Z * (p.NyP >> 4) + tr) * p.S + s0 + sl] = (unsigned char)(row[j] == 1u ? 1 : 0);
        }
    }
}

static int ft_sparse_tp()
{   // tile positions per block of the sparse grid (a multiple of 8 = whole marching-cubes segments)
    static const int v = getenv("TOMO_SPARSE_TP") ? atoi(getenv("TOMO_SPARSE_TP")) : 16;
    return v >= 8 && v <= FT_MAXT && v % 8 == 0 ? v : 16;
}
#define FT_SPARSE_TP ft_sparse_tp()
TOMO_API int64_t tomo_field_span_bytes(int nz, int ny, int nx, int pad)
{   // span + comb maps, the work list and its counter
    FieldParams p;
    (void)fill_params(p, nz, ny, nx, pad, nullptr, nullptr, FT_SPARSE_TP);
    const int64_t tiles = (int64_t)p.Nz * p.ntr * p.NT, blocks = (int64_t)p.nxc * p.ntr * p.nzg;
    return 2 * ((tiles + 63) / 64 * 64) + 4 * blocks + 256;
}

// tomo_field_fill_bits that leaves out what marching cubes cannot read (see above): `field` is only valid within one
// voxel of the iso-surface's cells afterwards.  span_ws: tomo_field_span_bytes(...) bytes of device scratch.
TOMO_API int tomo_field_fill_bits_sparse(const uint64_t *bits, float *field, int nz, int ny, int nx, int pad,
                                         unsigned long long *signs, uint8_t *gcls, uint8_t *span_ws, void *stream)
{
    if (!bits || !field || !span_ws || !signs || !gcls || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1))
        return TOMO_E_ARG;
    if (((uintptr_t)field & 127u) != 0 || ((uintptr_t)span_ws & 3u) != 0) return TOMO_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    FieldParams p;
    size_t lds = fill_params(p, nz, ny, nx, pad, signs, gcls, FT_SPARSE_TP);
    const int64_t blocks = (int64_t)p.nxc * p.ntr * p.nzg, sblocks = (int64_t)p.ntr * ((p.Nz + FSPAN_Z - 1) / FSPAN_Z);
    const int64_t per_slice = (int64_t)p.ntr * p.NT, tiles = per_slice * p.Nz, tiles_al = (tiles + 63) / 64 * 64;
    if (blocks > 0x7fffffff || sblocks > 0x7fffffff || ceil_div64(tiles, 256) > 0x7fffffff) return TOMO_E_SIZE;
    unsigned char *span = span_ws, *comb = span_ws + tiles_al;
    u32 *count = (u32 *)(span_ws + 2 * tiles_al), *list = count + 16;
    if (hipMemsetAsync(count, 0, sizeof(u32), s) != hipSuccess) return TOMO_E_LAUNCH;
    hipLaunchKernelGGL(field_span_kernel, dim3((unsigned)sblocks), dim3(64 * FSPAN_Z), (size_t)FSPAN_Z * 2 * (p.NT + 2) * sizeof(u32),
                       s, (const u32 *)bits, span, nz, ny, nx, pad, p.SW32, p.Nz, p.Ny, nx + 2 * pad, p.NT, p.ntr);
    hipLaunchKernelGGL(field_comb_kernel, dim3((unsigned)ceil_div64(tiles, 256)), dim3(256), 0, s, (const unsigned char *)span,
                       comb, p.Nz, per_slice);
    hipLaunchKernelGGL(field_worklist_kernel, dim3((unsigned)ceil_div64(blocks, 256)), dim3(256), 0, s, (const unsigned char *)comb,
                       p, list, count);
    p.comb = comb; p.list = list; p.count = count;
    if (getenv("TOMO_EXP_NEAR_DENSE")) p.comb = nullptr;      // experiment: the listed blocks write ALL their tiles
    hipLaunchKernelGGL(field_tile_kernel<true>, dim3((unsigned)blocks), dim3(FT_THREADS), lds, s, (const u32 *)bits, field, p);
    return tomo_status();
}

TOMO_API int tomo_field_signs_fused(int nx)
{   // does tomo_field_fill(gaussian = 1) write the sign records itself for this row width?  (always, since ABI 2)
    (void)nx;
    return 1;
}

TOMO_API int64_t tomo_sign_buffer_words(int Nz, int Ny, int Nx, int xorg)
{   // uint64 words of the sign records [Nz][S][NyP][4]
    int64_t S = tomo_mc_segments_per_row(Nx, xorg), NyP = tomo_sign_rows(Ny);
    return (int64_t)Nz * S * NyP * 4 + 8;
}

TOMO_API int tomo_field_fill(const uint64_t *ext, float *field, int nz, int ny, int nx, int pad, int gaussian,
                             unsigned long long *signs, uint8_t *gcls, void *stream)
{
    if (!ext || !field || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1) || ((signs != nullptr) != (gcls != nullptr)))
        return TOMO_E_ARG;
    if (((uintptr_t)field & 127u) != 0) return TOMO_E_ARG;
    FieldParams p;
    size_t lds = fill_params(p, nz, ny, nx, pad, gaussian ? signs : nullptr, gaussian ? gcls : nullptr);
    hipStream_t s = (hipStream_t)stream;
    if (!gaussian) {
        const int Nx = nx + 2 * pad;
        int64_t total = (int64_t)p.Nz * p.Ny * Nx;
        int64_t blocks = ceil_div64(total, 256);
        if (blocks > 0x7fffffff) return TOMO_E_SIZE;
        hipLaunchKernelGGL(field_raw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const u64 *)ext, field, p.Nz, p.Ny,
                           Nx, pad, p.EY, p.EWX32 / 2, p.pitch, tomo_field_xorg(pad));
        return tomo_status();
    }
    int64_t blocks = (int64_t)p.nxc * p.ntr * p.nzg;
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(field_tile_kernel<false>, dim3((unsigned)blocks), dim3(FT_THREADS), lds, s, (const u32 *)ext, field, p);
    return tomo_status();
}

// The same field straight from the plain bit volume (uint64 (nz, ny, ceil(nx / 64)) words): the extended volume is
// formed on the fly while the block stages its input, so tomo_extend_bits and its 1/8 B per voxel round trip drop out.
TOMO_API int tomo_field_fill_bits(const uint64_t *bits, float *field, int nz, int ny, int nx, int pad,
                                  unsigned long long *signs, uint8_t *gcls, void *stream)
{
    if (!bits || !field || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1) || ((signs != nullptr) != (gcls != nullptr)))
        return TOMO_E_ARG;
    if (((uintptr_t)field & 127u) != 0) return TOMO_E_ARG;
    FieldParams p;
    size_t lds = fill_params(p, nz, ny, nx, pad, signs, gcls);
    int64_t blocks = (int64_t)p.nxc * p.ntr * p.nzg;
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(field_tile_kernel<true>, dim3((unsigned)blocks), dim3(FT_THREADS), lds, (hipStream_t)stream,
                       (const u32 *)bits, field, p);
    return tomo_status();
}
