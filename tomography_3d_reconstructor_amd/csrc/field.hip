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
// Kernel structure (memory-bound: 1 bit in, 4 B out per padded voxel):
//   * pass 1 (z): its input is binary, so the result is one of 18 doubles -> an 18-entry LDS LUT
//     indexed by (centre bit, sum of the +-2 bits, sum of the +-1 bits), four voxels per lookup word;
//   * pass 2 (y): each lane owns 4 consecutive x columns and marches down y keeping a 5-row window
//     of pass-1 values in registers (each pass-1 value is looked up once, used 5 times);
//   * pass 3 (x): needs pass-2 values of the two columns left and right: wave-level DPP shifts, and
//     one LDS slot per wave edge (double buffered, one barrier per row);
//   * stores: one aligned float4 per lane per row (1 KiB contiguous per wave);
//   * wave-uniform fast path: if the 5x5 neighbourhood of a wave's 256 columns is all-0 / all-1 the
//     result is a constant (computed with the same operation sequence) and no float64 work is done.
// All border handling (zero pad ring, scipy 'reflect') is already materialised in the extended bit
// volume (bits.hip: extend_kernel), so the hot loop has no boundary branches except the row ends.
#include "tomo_common.h"

#define FW0 0x1.92b965ef5aaefp-1    // exp(-2 x^2)/sum for x = 0, +-1, +-2 as produced by the pinned
#define FW1 0x1.b405b9842b206p-4    // oracle environment (SciPy 1.7.1 / NumPy 1.26.4)
#define FW2 0x1.14aebe6a24088p-12

struct FieldParams {
    int nz, ny, nx, pad;
    int Nz, Ny, Nx;
    int EY, EWX32;          // ext rows per slice, ext 32-bit words per row
    int64_t pitch;
    int xorg;
    int nq;                 // data lanes = ceil(nx / 4)
    int nlanes;             // lanes that compute pass 2 (nq, +1 when nx % 4 != 0)
    int r;                  // nx % 4
    int rows_per_block;
    int W;                  // 32-bit ext words staged per row and block
    u64 *signs;             // sign records [Z][S][NyP][4] (may be null)
    int S;                  // marching-cubes segments per row
    int NyP;                // rows per (Z, segment) block of records: Ny rounded up to 16
    u32 *sflags;            // per (Z, segment, row chunk): 1 = records final, 0 = lane-major words to convert
    const float *fieldp;    // (convert kernel) the field, to read the right pad column
};

__device__ static inline double tap5(double a, double b, double c, double d, double e)
{   // a..e = x[-2], x[-1], x[0], x[+1], x[+2]
    double t = c * FW0;
    t += (a + e) * FW2;
    t += (b + d) * FW1;
    return t;
}

#define FIELD_MAX_STAGE_ROWS 36   // rows_per_block (<= 32) + 4
#define FIELD_NW 10               // 32-bit words that can hold a wave's 256 columns + 2 halo bits on each side
// LDS: [5 slices][rows_per_block + 4 rows][W words] of the extended bit volume (the block's whole input),
// followed by the per-(slice, word) OR / AND over the rows (2 x 5 x W words).
extern __shared__ u32 s_bits[];

// The two pad columns X = 0 and X = Nx-1 of the block's rows, from the values collected in LDS.  Isolated
// 4-byte stores into otherwise untouched 128-byte lines are ~9x as expensive as whole-line stores on this HBM
// (read-modify-write), so for the common aligned shape (nx % 32 == 0) the whole line around each pad column is
// written (zeros elsewhere: those columns are outside the field): 8 rows x 128 B per store instruction.
__device__ static inline void store_pad_columns(float *obase, const FieldParams &p, float (*s_pad)[32], bool w_first,
                                                bool w_last, int nrows_out, int lane)
{
    if ((p.nx & 31) == 0) {
        const int q = lane & 7;
        for (int j = 0; j < nrows_out; j += 8) {
            const int row = j + (lane >> 3);
            if (row < nrows_out) {
                float *o = obase + (int64_t)row * p.pitch;
                if (w_first) *(float4 *)(o + 4 * q) = make_float4(0.f, 0.f, 0.f, q == 7 ? s_pad[0][row] : 0.f);
                if (w_last) *(float4 *)(o + 32 + p.nx + 4 * q) = make_float4(q == 0 ? s_pad[1][row] : 0.f, 0.f, 0.f, 0.f);
            }
        }
    } else if (lane < nrows_out) {
        float *o = obase + (int64_t)lane * p.pitch;
        if (w_first) o[p.xorg] = s_pad[0][lane];
        if (w_last) o[p.xorg + p.Nx - 1] = s_pad[1][lane];
    }
}

// Sign records of the two segments that hold only a pad column: segment 0 (column 31 = X 0 is its lane 63,
// element 3) and, when the right pad column starts a new segment, segment wave + 2 (its lane 0, element 0).
__device__ static inline void store_pad_signs(const FieldParams &p, float (*s_pad)[32], bool left, bool right, int Z,
                                              int wave, int Y0, int nrows_out, int lane)
{
    if (lane >= nrows_out) return;
    if (left) {
        ulonglong2 *o = (ulonglong2 *)(p.signs + ((((int64_t)Z * p.S + 0) * p.NyP) + Y0 + lane) * 4);
        o[0] = make_ulonglong2(0ull, 0ull);
        o[1] = make_ulonglong2(0ull, s_pad[0][lane] > 0.5f ? (1ull << 63) : 0ull);
    }
    if (right && wave + 2 < p.S) {
        ulonglong2 *o = (ulonglong2 *)(p.signs + ((((int64_t)Z * p.S + wave + 2) * p.NyP) + Y0 + lane) * 4);
        o[0] = make_ulonglong2(s_pad[1][lane] > 0.5f ? 1ull : 0ull, 0ull);
        o[1] = make_ulonglong2(0ull, 0ull);
    }
}

__device__ static inline float p_field_pad(const FieldParams &p, int Z, int Y)
{
    return p.fieldp[((int64_t)Z * p.Ny + Y) * p.pitch + p.xorg + p.Nx - 1];
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void field_gauss_kernel(const u32 *__restrict__ ext32, float *__restrict__ field,
                                                           const FieldParams p)
{
    __shared__ double s_lut[18];
    __shared__ double s_halo[2][16][4];
    __shared__ float s_pad[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6, T = blockDim.x;
    if (tid < 18) {
        int c = tid / 9, s2 = (tid / 3) % 3, s1 = tid % 3;
        double t = (double)c * FW0;
        t += (double)s2 * FW2;
        t += (double)s1 * FW1;
        s_lut[tid] = t;
    }
    const bool multi = gridDim.x > 1;
    const int tstart = multi ? (int)blockIdx.x * (T - 2) - 1 : 0;
    const int t = tstart + tid;                     // global lane (4-column group) index
    const bool valid = t >= 0 && t < p.nlanes;
    const bool can_out = valid && (!multi || (tid > 0 && tid < T - 1));
    const int Z = blockIdx.z;
    const int Y0 = (int)blockIdx.y * p.rows_per_block;
    const int Y1 = Y0 + p.rows_per_block < p.Ny ? Y0 + p.rows_per_block : p.Ny;
    const int nrows = Y1 - Y0 + 4;                  // ext rows Y0 .. Y1+3
    const int e0 = 4 * t + 4;                       // ext bit of this lane's first element
    const int col0 = 4 * t + 32;                    // its float column in the field row (128-byte aligned per wave)
    const int sh = e0 & 31;
    const int wbase = tstart < 0 ? 0 : (4 * tstart + 2) >> 5;   // word of the block's first halo bit
    const int W = p.W;
    const int RS = (p.rows_per_block + 4) * W;      // LDS slice stride
    u32 *const s_or = s_bits + 5 * RS, *const s_and = s_or + 5 * W;
    // ---- stage the block's input bits: 5 slices x nrows x W words; thread j < 5*W owns (slice k, word w)
    if (tid < 5 * W) {
        const int64_t sstride = (int64_t)p.EY * p.EWX32;
        const int k = tid / W, w = tid - k * W;
        const bool inb = wbase + w < p.EWX32;
        const u32 *sp = ext32 + (int64_t)Z * sstride + (int64_t)Y0 * p.EWX32 + wbase + (int64_t)k * sstride + w;
        u32 *dp = s_bits + k * RS + w;
        u32 vor = 0u, vand = 0xffffffffu;
#pragma unroll
        for (int h = 0; h < 2; h++) {               // two batches: all loads of a batch are in flight together
            u32 v[FIELD_MAX_STAGE_ROWS / 2];
#pragma unroll
            for (int j = 0; j < FIELD_MAX_STAGE_ROWS / 2; j++) {
                int yy = h * (FIELD_MAX_STAGE_ROWS / 2) + j;
                v[j] = (inb && yy < nrows) ? sp[(int64_t)yy * p.EWX32] : 0u;
            }
#pragma unroll
            for (int j = 0; j < FIELD_MAX_STAGE_ROWS / 2; j++) {
                int yy = h * (FIELD_MAX_STAGE_ROWS / 2) + j;
                if (yy < nrows) { dp[yy * W] = v[j]; vor |= v[j]; vand &= v[j]; }
            }
        }
        s_or[tid] = vor; s_and[tid] = vand;
    }
    __syncthreads();
    const int wl = valid ? (e0 >> 5) - wbase : 0;   // this lane's word inside the staged rows

    // element k of this lane is padded column X = 4t + k + pad
    const int Xfirst = 4 * t + p.pad;
    const bool full4 = can_out && t < p.nq && (Xfirst + 3 < p.Nx);
    const bool first_lane = (t == 0);
    const bool last_r0 = (p.r == 0) && (t == p.nq - 1);
    const int t0w = tstart + wave * 64;
    const bool w_first = (0 >= t0w) && (0 < t0w + 64);                                 // wave holds lane t = 0
    const bool w_last = (p.r == 0) && (p.nq - 1 >= t0w) && (p.nq - 1 < t0w + 64);      // wave holds the last_r0 lane

    // constants of the all-ones interior, produced by the same operation sequence
    const double p1one = s_lut[17];
    const double c2 = tap5(p1one, p1one, p1one, p1one, p1one);
    const double c3 = tap5(c2, c2, c2, c2, c2);
    const float c3f = (float)c3;

    // ---- is the wave's whole neighbourhood (its 256 columns + 2 on each side, all staged rows, 5 slices) constant?
    int wconst;     // 0 general, 1 all zero, 2 all one
    {
        const int ebeg = 4 * t0w + 2 > 0 ? 4 * t0w + 2 : 0, eend = 4 * (t0w + 64) + 6;    // ext bits [ebeg, eend)
        const int wf = (ebeg >> 5) - wbase, nw = ((eend - 1) >> 5) - wbase - wf + 1;      // nw <= FIELD_NW
        u32 accor = 0u, accand = 0xffffffffu;
        if (lane < 5 * FIELD_NW) {
            const int k = lane / FIELD_NW, j = lane - k * FIELD_NW;
            if (j < nw && wf + j < W) {
                const int b0 = (wf + j + wbase) << 5;              // first ext bit of this word
                u32 m = 0xffffffffu;
                if (b0 < ebeg) m &= 0xffffffffu << (ebeg - b0);
                if (b0 + 32 > eend) m &= 0xffffffffu >> (b0 + 32 - eend);
                accor = s_or[k * W + wf + j] & m;
                accand = s_and[k * W + wf + j] | ~m;
            }
        }
        const bool allz = __all(accor == 0u), allo = __all(accand == 0xffffffffu);
        wconst = allz ? 1 : (allo ? 2 : 0);
    }
    float *const obase = field + ((int64_t)Z * p.Ny + Y0) * p.pitch;
    // sign records (marching-cubes pass 1 input): wave w of a single-block row is exactly MC segment w + 1
    const bool do_signs = p.signs != nullptr && !multi;
    u64 *const srec = do_signs ? p.signs + ((((int64_t)Z * p.S + (wave + 1)) * p.NyP) + Y0) * 4 : nullptr;
    const int padlane = p.nq - t0w;                 // lane whose element 0 is the right pad column (when nx % 4 == 0)
    if (do_signs && lane == 0) p.sflags[((int64_t)Z * p.S + (wave + 1)) * gridDim.y + blockIdx.y] = wconst ? 1u : 0u;
    if (wconst) {   // publish the (constant) pass-2 halo values for both row parities, once
        const double k2 = wconst == 1 ? 0.0 : c2;
        if (lane < 8) s_halo[lane >> 2][wave][lane & 3] = k2;
    }
    __syncthreads();
    if (wconst) {
        // constant field over the wave's columns for every row of the block: a pure store loop, then the wave
        // ends (terminated waves no longer take part in the block's barriers)
        const float kf = wconst == 1 ? 0.0f : c3f;
        const float4 k4 = make_float4(kf, kf, kf, kf);
        float *o = obase + col0;
        if (full4) {
            for (int Y = Y0; Y < Y1; Y++, o += p.pitch) *(float4 *)o = k4;
        } else if (can_out) {
            for (int Y = Y0; Y < Y1; Y++, o += p.pitch) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (Xfirst + k < p.Nx) o[k] = kf;
            }
        }
        if (p.pad && (w_first || w_last)) {                        // pad columns: their bits are zero => wconst == 1
            if (lane < 32) { if (w_first) s_pad[0][lane] = 0.0f; if (w_last) s_pad[1][lane] = 0.0f; }
            store_pad_columns(obase, p, s_pad, w_first, w_last, Y1 - Y0, lane);
        }
        if (do_signs) {
            if (wconst == 2)                               // all-zero waves: the caller zeroed the buffer
                for (int j = 0; j < Y1 - Y0; j += 16) {    // 16 rows x 32 B = one 512-byte store
                    int nr = Y1 - Y0 - j < 16 ? Y1 - Y0 - j : 16;
                    if (lane < 4 * nr) srec[(int64_t)j * 4 + lane] = ~0ull;
                }
            store_pad_signs(p, s_pad, w_first, w_last && padlane == 64, Z, wave, Y0, Y1 - Y0, lane);
        }
        return;
    }

    // pass-1 code word of staged row yy: 4 bytes, byte k = 9*c + 3*s2 + s1 of column k
#define SPREAD(n) ((((n) >> sh) & 0xFu) * 0x00204081u & 0x01010101u)
#define CODES(yy, dst)                                                                          \
    {                                                                                           \
        const u32 *rp = s_bits + (yy) * W + wl;                                                 \
        u32 n0 = rp[0], n1 = rp[RS], n2 = rp[2 * RS], n3 = rp[3 * RS], n4 = rp[4 * RS];         \
        dst = valid ? (SPREAD(n2) * 9u + (SPREAD(n0) + SPREAD(n4)) * 3u + (SPREAD(n1) + SPREAD(n3))) : 0u; \
    }
#define LOOKUP(dst, c)                                    \
    {                                                     \
        dst[0] = s_lut[(c) & 0xffu];                      \
        dst[1] = s_lut[((c) >> 8) & 0xffu];               \
        dst[2] = s_lut[((c) >> 16) & 0xffu];              \
        dst[3] = s_lut[(c) >> 24];                        \
    }

    double wa[4], wb[4], wc[4], wd[4], we[4];
    u32 ca, cb, cc, cd, ce;
    CODES(0, ca); LOOKUP(wa, ca);
    CODES(1, cb); LOOKUP(wb, cb);
    CODES(2, cc); LOOKUP(wc, cc);
    CODES(3, cd); LOOKUP(wd, cd);
    int buf = 0;

    // one row: window rows (A,B,C,D,E) = Y-2..Y+2; E enters from staged row (Y - Y0) + 4
#define STEP(A, B, C, D, E, cA, cB, cC, cD, cE)                                                        \
    {                                                                                                  \
        CODES(Y - Y0 + 4, cE);                                                                         \
        const bool z0 = (cA | cB | cC | cD | cE) == 0u;                                                \
        const bool o1 = (cA & cB & cC & cD & cE) == 0x11111111u;                                       \
        const bool wz = __all(z0), wo = __all(o1);                                                     \
        const bool wu = wz || wo;                                                                      \
        const double K = wz ? 0.0 : c2;                                                                \
        double q0, q1, q2, q3;                                                                         \
        if (wu) { q0 = q1 = q2 = q3 = K; E[0] = E[1] = E[2] = E[3] = wz ? 0.0 : p1one; }               \
        else {                                                                                         \
            LOOKUP(E, cE);                                                                             \
            q0 = tap5(A[0], B[0], C[0], D[0], E[0]);                                                   \
            q1 = tap5(A[1], B[1], C[1], D[1], E[1]);                                                   \
            q2 = tap5(A[2], B[2], C[2], D[2], E[2]);                                                   \
            q3 = tap5(A[3], B[3], C[3], D[3], E[3]);                                                   \
        }                                                                                              \
        if (lane == 0) { s_halo[buf][wave][0] = q0; s_halo[buf][wave][1] = q1; }                       \
        if (lane == 63) { s_halo[buf][wave][2] = q2; s_halo[buf][wave][3] = q3; }                      \
        __syncthreads();                                                                               \
        /* pass-2 values of the two columns left / right of this lane's four */                        \
        double L2 = K, L3 = K, R0 = K, R1 = K;                                                         \
        if (lane == 0) {                                                                               \
            if (wave > 0) { L2 = s_halo[buf][wave - 1][2]; L3 = s_halo[buf][wave - 1][3]; }            \
            else { L2 = 0.0; L3 = 0.0; }                                                               \
        }                                                                                              \
        if (lane == 63) {                                                                              \
            if (wave < nwaves - 1) { R0 = s_halo[buf][wave + 1][0]; R1 = s_halo[buf][wave + 1][1]; }   \
            else { R0 = 0.0; R1 = 0.0; }                                                               \
        }                                                                                              \
        buf ^= 1;                                                                                      \
        if (first_lane) { L2 = p.pad ? 0.0 : q1; L3 = p.pad ? 0.0 : q0; }                              \
        if (last_r0) { R0 = p.pad ? 0.0 : q3; R1 = p.pad ? 0.0 : q2; }                                 \
        float o0, o1f, o2, o3;                                                                         \
        const bool uni = wu && __all(L2 == K && L3 == K && R0 == K && R1 == K);                        \
        if (uni) { o0 = o1f = o2 = o3 = wz ? 0.0f : c3f; }                                             \
        else {                                                                                         \
            double l2 = dpp_from_prev_f64(q2), l3 = dpp_from_prev_f64(q3);                             \
            double r0 = dpp_from_next_f64(q0), r1 = dpp_from_next_f64(q1);                             \
            if (lane != 0) { L2 = l2; L3 = l3; }                                                       \
            if (lane != 63) { R0 = r0; R1 = r1; }                                                      \
            if (first_lane) { L2 = p.pad ? 0.0 : q1; L3 = p.pad ? 0.0 : q0; }                          \
            if (last_r0) { R0 = p.pad ? 0.0 : q3; R1 = p.pad ? 0.0 : q2; }                             \
            o0 = (float)tap5(L2, L3, q0, q1, q2);                                                      \
            o1f = (float)tap5(L3, q0, q1, q2, q3);                                                     \
            o2 = (float)tap5(q0, q1, q2, q3, R0);                                                      \
            o3 = (float)tap5(q1, q2, q3, R0, R1);                                                      \
        }                                                                                              \
        if (do_signs) {   /* 4 sign bits of this lane and row; 8 rows per word, stored lane-major (256 B per wave) */ \
            u32 n4 = ((u32)(0x3F000000 - __float_as_int(o0)) >> 31) | (((u32)(0x3F000000 - __float_as_int(o1f)) >> 31) << 1) | \
                     (((u32)(0x3F000000 - __float_as_int(o2)) >> 31) << 2) | (((u32)(0x3F000000 - __float_as_int(o3)) >> 31) << 3); \
            nibacc |= n4 << (((Y - Y0) & 7) * 4);                                                      \
            if (((Y - Y0) & 7) == 7 || Y == Y1 - 1) {                                                  \
                ((u32 *)srec)[(int64_t)((Y - Y0) >> 3) * 64 + lane] = nibacc;                          \
                nibacc = 0;                                                                            \
            }                                                                                          \
        }                                                                                              \
        if (full4) {                                                                                   \
            *(float4 *)(orow + col0) = make_float4(o0, o1f, o2, o3);                                     \
        } else if (can_out) {                                                                          \
            if (Xfirst + 0 < p.Nx) orow[col0 + 0] = o0;                                                  \
            if (Xfirst + 1 < p.Nx) orow[col0 + 1] = o1f;                                                 \
            if (Xfirst + 2 < p.Nx) orow[col0 + 2] = o2;                                                  \
            if (Xfirst + 3 < p.Nx) orow[col0 + 3] = o3;                                                  \
        }                                                                                              \
        if (p.pad && can_out) {   /* pad columns X = 0 / Nx-1 (pass-2 value there is 0): kept in LDS, stored once */ \
            if (first_lane) { double s = 0.0; s += (q0 + q1) * FW2; s += q0 * FW1; s_pad[0][Y - Y0] = (float)s; } \
            if (last_r0) { double s = 0.0; s += (q2 + q3) * FW2; s += q3 * FW1; s_pad[1][Y - Y0] = (float)s; }   \
        }                                                                                              \
        orow += p.pitch;                                                                               \
    }

    // sign bits of this lane: 4 per row, 8 rows per register, parked in LDS per 8 rows and turned into records
    // after the row loop (keeps ballots / SGPR pressure out of the hot loop)
    u32 nibacc = 0;
    float *orow = obase;
    int Y = Y0;
    while (Y < Y1) {
        STEP(wa, wb, wc, wd, we, ca, cb, cc, cd, ce); if (++Y >= Y1) break;
        STEP(wb, wc, wd, we, wa, cb, cc, cd, ce, ca); if (++Y >= Y1) break;
        STEP(wc, wd, we, wa, wb, cc, cd, ce, ca, cb); if (++Y >= Y1) break;
        STEP(wd, we, wa, wb, wc, cd, ce, ca, cb, cc); if (++Y >= Y1) break;
        STEP(we, wa, wb, wc, wd, ce, ca, cb, cc, cd); ++Y;
    }
    if (p.pad && (w_first || w_last)) {
        store_pad_columns(obase, p, s_pad, w_first, w_last, Y1 - Y0, lane);
        if (do_signs) store_pad_signs(p, s_pad, w_first, w_last && padlane == 64, Z, wave, Y0, Y1 - Y0, lane);
    }
#undef STEP
#undef LOOKUP
#undef CODES
#undef SPREAD
}

// The non-constant waves of the field kernel leave their sign bits LANE-MAJOR in the record area (one 32-bit word
// per lane and 8 rows: bit 4r+k = element k of row r); this turns them into the ballot records the classify pass
// reads (bit L of word k of row r), in place.  One wave per flagged (Z, segment, chunk of rows): ~1 KB each.
__global__ __launch_bounds__(256) void field_signs_convert_kernel(FieldParams p, int nchunks, int nwaves, int64_t ntasks)
{
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // (Z, wave, chunk), chunk fastest
    if (task >= ntasks) return;
    const int chunk = (int)(task % nchunks);
    const int64_t tz = task / nchunks;
    const int wave = (int)(tz % nwaves), s = wave + 1;
    const int Z = (int)(tz / nwaves);
    if (p.sflags[((int64_t)Z * p.S + s) * nchunks + chunk] != 0u) return;
    const int Y0 = chunk * p.rows_per_block;
    const int nr = (Y0 + p.rows_per_block < p.Ny ? p.rows_per_block : p.Ny - Y0);
    u64 *srec = p.signs + ((((int64_t)Z * p.S + s) * p.NyP) + Y0) * 4;
    u32 w[4];
#pragma unroll
    for (int g = 0; g < 4; g++) w[g] = (g * 8 < nr) ? ((const u32 *)srec)[g * 64 + lane] : 0u;
    // right pad column = element 0 of lane `padlane` of the last wave: its value was computed separately
    const int t0w = wave * 64, padlane = p.nq - t0w;
    const bool patch = p.pad && p.r == 0 && padlane >= 0 && padlane < 64;
    u64 rec = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) {
#pragma unroll
        for (int rr = 0; rr < 8; rr++) {
            const int r = g * 8 + rr;
            u32 n4 = (w[g] >> (rr * 4)) & 15u;
            if (patch && lane == padlane && r < nr) {
                float pv = p_field_pad(p, Z, Y0 + r);
                n4 = (n4 & ~1u) | (pv > 0.5f ? 1u : 0u);
            }
            u64 b0 = __ballot(n4 & 1u), b1 = __ballot(n4 & 2u), b2 = __ballot(n4 & 4u), b3 = __ballot(n4 & 8u);
            const int r16 = r & 15;
            if ((lane >> 2) == r16) rec = (lane & 3) == 0 ? b0 : ((lane & 3) == 1 ? b1 : ((lane & 3) == 2 ? b2 : b3));
            if (r < nr && (r16 == 15 || r == nr - 1)) {
                if (lane < 4 * (r16 + 1)) srec[(int64_t)(r - r16) * 4 + lane] = rec;
            }
        }
    }
    if (lane == 0) p.sflags[((int64_t)Z * p.S + s) * nchunks + chunk] = 1u;   // converted: a second call is a no-op
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
    int e = X + 4 - pad;
    u64 w = ext[((int64_t)(Z + 2) * EY + (Y + 2)) * EWX + (e >> 6)];
    field[((int64_t)Z * Ny + Y) * pitch + xorg + X] = (float)((w >> (e & 63)) & 1ull);
}

static void fill_params(FieldParams &p, int nz, int ny, int nx, int pad, unsigned long long *signs)
{
    p.nz = nz; p.ny = ny; p.nx = nx; p.pad = pad;
    p.Nz = nz + 2 * pad; p.Ny = ny + 2 * pad; p.Nx = nx + 2 * pad;
    p.EY = (int)tomo_ext_rows(ny, pad);
    p.EWX32 = 2 * (int)tomo_ext_words_per_row(nx, pad);
    p.pitch = tomo_field_pitch(nx, pad);
    p.xorg = tomo_field_xorg(pad);
    p.nq = (nx + 3) / 4;
    p.r = nx % 4;
    p.nlanes = p.nq + (p.r ? 1 : 0);
    int threads = (p.nlanes + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    p.rows_per_block = threads <= 256 ? 32 : 16;
    p.W = (4 * threads + 4 + 31) / 32 + 1;
    if (p.W > p.EWX32) p.W = p.EWX32;
    p.S = (int)tomo_mc_segments_per_row(p.Nx, p.xorg);
    p.NyP = (int)tomo_sign_rows(p.Ny);
    p.signs = (u64 *)signs;
    p.fieldp = nullptr;
    p.sflags = signs ? (u32 *)((u64 *)signs + (int64_t)p.Nz * p.S * p.NyP * 4) : nullptr;
}

TOMO_API int tomo_field_signs_fused(int nx)
{   // does tomo_field_fill(gaussian = 1) write the sign records itself for this row width?  (one block per row)
    int nq = (nx + 3) / 4, nlanes = nq + ((nx % 4) ? 1 : 0);
    return (nlanes + 63) / 64 * 64 <= 1024 ? 1 : 0;
}

TOMO_API int64_t tomo_sign_buffer_words(int Nz, int Ny, int Nx, int xorg)
{   // uint64 words: the sign records [Nz][S][NyP][4] followed by the field kernel's per-chunk flags
    int64_t S = tomo_mc_segments_per_row(Nx, xorg), NyP = tomo_sign_rows(Ny);
    int64_t chunks = (Ny + 15) / 16;
    return (int64_t)Nz * S * NyP * 4 + ((int64_t)Nz * S * chunks + 1) / 2 + 8;
}

// Second step of the fused sign records: convert the lane-major words of the non-constant waves (see the kernel).
TOMO_API int tomo_field_signs_finish(const float *field, int nz, int ny, int nx, int pad, unsigned long long *signs,
                                     void *stream)
{
    if (!field || !signs || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1)) return TOMO_E_ARG;
    if (!tomo_field_signs_fused(nx)) return TOMO_E_ARG;
    FieldParams p;
    fill_params(p, nz, ny, nx, pad, signs);
    p.fieldp = field;
    int nchunks = (int)ceil_div64(p.Ny, p.rows_per_block);
    int nwaves = (p.nlanes + 63) / 64;
    int64_t ntasks = (int64_t)p.Nz * nwaves * nchunks;
    int64_t blocks = ceil_div64(ntasks, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    hipLaunchKernelGGL(field_signs_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, nchunks,
                       nwaves, ntasks);
    return tomo_status();
}

TOMO_API int tomo_field_fill(const uint64_t *ext, float *field, int nz, int ny, int nx, int pad, int gaussian,
                             unsigned long long *signs, void *stream)
{
    if (!ext || !field || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1)) return TOMO_E_ARG;
    FieldParams p;
    fill_params(p, nz, ny, nx, pad, (gaussian && tomo_field_signs_fused(nx)) ? signs : nullptr);
    hipStream_t s = (hipStream_t)stream;
    if (!gaussian) {
        int64_t total = (int64_t)p.Nz * p.Ny * p.Nx;
        int64_t blocks = ceil_div64(total, 256);
        if (blocks > 0x7fffffff) return TOMO_E_SIZE;
        hipLaunchKernelGGL(field_raw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const u64 *)ext, field, p.Nz, p.Ny,
                           p.Nx, pad, p.EY, p.EWX32 / 2, p.pitch, p.xorg);
        return tomo_status();
    }
    if (p.Nz > 65535) return TOMO_E_SIZE;
    int threads = (p.nlanes + 63) / 64 * 64;
    unsigned gx = 1;
    if (threads > 1024) {
        threads = 1024;
        gx = (unsigned)ceil_div64(p.nlanes, threads - 2);
    }
    size_t lds = ((size_t)5 * (p.rows_per_block + 4) * p.W + 10 * p.W) * sizeof(u32);
    dim3 grid(gx, (unsigned)ceil_div64(p.Ny, p.rows_per_block), (unsigned)p.Nz);
    if (threads <= 256)
        hipLaunchKernelGGL(field_gauss_kernel<256>, grid, dim3(threads), lds, s, (const u32 *)ext, field, p);
    else
        hipLaunchKernelGGL(field_gauss_kernel<1024>, grid, dim3(threads), lds, s, (const u32 *)ext, field, p);
    return tomo_status();
}
