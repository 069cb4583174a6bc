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
};

__device__ static inline double tap5(double a, double b, double c, double d, double e)
{   // a..e = x[-2], x[-1], x[0], x[+1], x[+2]
    double t = c * FW0;
    t += (a + e) * FW2;
    t += (b + d) * FW1;
    return t;
}

template <int MAXT>
__global__ __launch_bounds__(MAXT) void field_gauss_kernel(const u32 *__restrict__ ext32, float *__restrict__ field,
                                                           const FieldParams p)
{
    __shared__ double s_lut[18];
    __shared__ double s_halo[2][16][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6, T = blockDim.x;
    if (tid < 18) {
        int c = tid / 9, s2 = (tid / 3) % 3, s1 = tid % 3;
        double t = (double)c * FW0;
        t += (double)s2 * FW2;
        t += (double)s1 * FW1;
        s_lut[tid] = t;
    }
    const bool multi = gridDim.x > 1;
    const int t = multi ? (int)blockIdx.x * (T - 2) + tid - 1 : tid;   // global lane (4-column group) index
    const bool valid = t >= 0 && t < p.nlanes;
    const bool can_out = valid && (!multi || (tid > 0 && tid < T - 1));
    const int Z = blockIdx.z;
    const int Y0 = (int)blockIdx.y * p.rows_per_block;
    const int Y1 = Y0 + p.rows_per_block < p.Ny ? Y0 + p.rows_per_block : p.Ny;
    const int e0 = 4 * t + 4;                      // ext bit == field column of this lane's first element
    const int sh = e0 & 31;
    const int64_t sstride = (int64_t)p.EY * p.EWX32;
    const u32 *base = ext32 + (int64_t)Z * sstride + (valid ? (e0 >> 5) : 0);
    __syncthreads();

    // constants of the all-ones interior, produced by the same operation sequence
    const double p1one = s_lut[17];
    const double c2 = tap5(p1one, p1one, p1one, p1one, p1one);
    const double c3 = tap5(c2, c2, c2, c2, c2);
    const float c3f = (float)c3;

    // element k of this lane is padded column X = 4t + k + pad
    const int Xfirst = 4 * t + p.pad;
    const bool full4 = can_out && t < p.nq && (Xfirst + 3 < p.Nx);
    const bool first_lane = (t == 0);
    const bool last_r0 = (p.r == 0) && (t == p.nq - 1);

    // pass-1 code word of ext row ey: 4 bytes, byte k = 9*c + 3*s2 + s1 of column k
#define LOAD_NIBBLES(ey, n0, n1, n2, n3, n4)                                             \
    {                                                                                    \
        const u32 *rp = base + (int64_t)(ey) * p.EWX32;                                  \
        n0 = rp[0]; n1 = rp[sstride]; n2 = rp[2 * sstride]; n3 = rp[3 * sstride]; n4 = rp[4 * sstride]; \
    }
#define SPREAD(n) ((((n) >> sh) & 0xFu) * 0x00204081u & 0x01010101u)
#define CODES(n0, n1, n2, n3, n4) (valid ? (SPREAD(n2) * 9u + (SPREAD(n0) + SPREAD(n4)) * 3u + (SPREAD(n1) + SPREAD(n3))) : 0u)
#define LOOKUP(dst, c)                                    \
    {                                                     \
        dst[0] = s_lut[(c) & 0xffu];                      \
        dst[1] = s_lut[((c) >> 8) & 0xffu];               \
        dst[2] = s_lut[((c) >> 16) & 0xffu];              \
        dst[3] = s_lut[(c) >> 24];                        \
    }

    double wa[4], wb[4], wc[4], wd[4], we[4];
    u32 ca, cb, cc, cd, ce;
    {
        u32 n0, n1, n2, n3, n4;
        LOAD_NIBBLES(Y0 + 0, n0, n1, n2, n3, n4); ca = CODES(n0, n1, n2, n3, n4); LOOKUP(wa, ca);
        LOAD_NIBBLES(Y0 + 1, n0, n1, n2, n3, n4); cb = CODES(n0, n1, n2, n3, n4); LOOKUP(wb, cb);
        LOAD_NIBBLES(Y0 + 2, n0, n1, n2, n3, n4); cc = CODES(n0, n1, n2, n3, n4); LOOKUP(wc, cc);
        LOAD_NIBBLES(Y0 + 3, n0, n1, n2, n3, n4); cd = CODES(n0, n1, n2, n3, n4); LOOKUP(wd, cd);
    }
    // prefetched raw words of the row that enters the window next
    u32 m0, m1, m2, m3, m4;
    LOAD_NIBBLES(Y0 + 4, m0, m1, m2, m3, m4);
    int buf = 0;

    // one row: window rows (A,B,C,D,E) = Y-2..Y+2, E is filled from the prefetched words
#define STEP(A, B, C, D, E, cA, cB, cC, cD, cE)                                                        \
    {                                                                                                  \
        cE = CODES(m0, m1, m2, m3, m4);                                                                \
        {   /* prefetch the next row (clamped: the last prefetch is never used) */                     \
            int eyn = Y + 5 < p.EY ? Y + 5 : p.EY - 1;                                                 \
            LOAD_NIBBLES(eyn, m0, m1, m2, m3, m4);                                                     \
        }                                                                                              \
        const bool z0 = (cA | cB | cC | cD | cE) == 0u;                                                \
        const bool o1 = (cA & cB & cC & cD & cE) == 0x11111111u;                                       \
        const bool wz = __all(z0), wo = __all(o1);                                                     \
        double q0, q1, q2, q3;                                                                         \
        if (wz) { q0 = q1 = q2 = q3 = 0.0; E[0] = E[1] = E[2] = E[3] = 0.0; }                          \
        else if (wo) { q0 = q1 = q2 = q3 = c2; E[0] = E[1] = E[2] = E[3] = p1one; }                    \
        else {                                                                                         \
            LOOKUP(E, cE);                                                                             \
            q0 = tap5(A[0], B[0], C[0], D[0], E[0]);                                                   \
            q1 = tap5(A[1], B[1], C[1], D[1], E[1]);                                                   \
            q2 = tap5(A[2], B[2], C[2], D[2], E[2]);                                                   \
            q3 = tap5(A[3], B[3], C[3], D[3], E[3]);                                                   \
        }                                                                                              \
        if (lane == 0) { s_halo[buf][wave][0] = q0; s_halo[buf][wave][1] = q1; }                       \
        if (lane == 63) { s_halo[buf][wave][2] = q2; s_halo[buf][wave][3] = q3; }                      \
        double L2 = dpp_from_prev_f64(q2), L3 = dpp_from_prev_f64(q3);                                 \
        double R0 = dpp_from_next_f64(q0), R1 = dpp_from_next_f64(q1);                                 \
        __syncthreads();                                                                               \
        if (lane == 0 && wave > 0) { L2 = s_halo[buf][wave - 1][2]; L3 = s_halo[buf][wave - 1][3]; }   \
        if (lane == 63 && wave < nwaves - 1) { R0 = s_halo[buf][wave + 1][0]; R1 = s_halo[buf][wave + 1][1]; } \
        buf ^= 1;                                                                                      \
        if (first_lane) { L2 = p.pad ? 0.0 : q1; L3 = p.pad ? 0.0 : q0; }                              \
        if (last_r0) { R0 = p.pad ? 0.0 : q3; R1 = p.pad ? 0.0 : q2; }                                 \
        float o0, o1f, o2, o3;                                                                         \
        const double K = wz ? 0.0 : c2;                                                                \
        const bool uni = (wz || wo) && __all(L2 == K && L3 == K && R0 == K && R1 == K);                \
        if (uni) { o0 = o1f = o2 = o3 = wz ? 0.0f : c3f; }                                             \
        else {                                                                                         \
            o0 = (float)tap5(L2, L3, q0, q1, q2);                                                      \
            o1f = (float)tap5(L3, q0, q1, q2, q3);                                                     \
            o2 = (float)tap5(q0, q1, q2, q3, R0);                                                      \
            o3 = (float)tap5(q1, q2, q3, R0, R1);                                                      \
        }                                                                                              \
        float *orow = field + ((int64_t)Z * p.Ny + Y) * p.pitch;                                      \
        if (full4) {                                                                                   \
            *(float4 *)(orow + e0) = make_float4(o0, o1f, o2, o3);                                     \
        } else if (can_out) {                                                                          \
            if (Xfirst + 0 < p.Nx) orow[e0 + 0] = o0;                                                  \
            if (Xfirst + 1 < p.Nx) orow[e0 + 1] = o1f;                                                 \
            if (Xfirst + 2 < p.Nx) orow[e0 + 2] = o2;                                                  \
            if (Xfirst + 3 < p.Nx) orow[e0 + 3] = o3;                                                  \
        }                                                                                              \
        if (p.pad && can_out) {   /* the two pad columns X = 0 and X = Nx-1 (pass-2 value there is 0) */ \
            if (first_lane) {                                                                          \
                double s = 0.0; s += (q0 + q1) * FW2; s += q0 * FW1;                                   \
                orow[p.xorg] = (float)s;                                                               \
            }                                                                                          \
            if (last_r0) {                                                                             \
                double s = 0.0; s += (q2 + q3) * FW2; s += q3 * FW1;                                   \
                orow[p.xorg + p.Nx - 1] = (float)s;                                                    \
            }                                                                                          \
        }                                                                                              \
    }

    int Y = Y0;
    while (Y < Y1) {
        STEP(wa, wb, wc, wd, we, ca, cb, cc, cd, ce); if (++Y >= Y1) break;
        STEP(wb, wc, wd, we, wa, cb, cc, cd, ce, ca); if (++Y >= Y1) break;
        STEP(wc, wd, we, wa, wb, cc, cd, ce, ca, cb); if (++Y >= Y1) break;
        STEP(wd, we, wa, wb, wc, cd, ce, ca, cb, cc); if (++Y >= Y1) break;
        STEP(we, wa, wb, wc, wd, ce, ca, cb, cc, cd); ++Y;
    }
#undef STEP
#undef LOOKUP
#undef CODES
#undef SPREAD
#undef LOAD_NIBBLES
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

TOMO_API int tomo_field_fill(const uint64_t *ext, float *field, int nz, int ny, int nx, int pad, int gaussian,
                             void *stream)
{
    if (!ext || !field || nz <= 0 || ny <= 0 || nx <= 0 || (pad != 0 && pad != 1)) return TOMO_E_ARG;
    FieldParams p;
    p.nz = nz; p.ny = ny; p.nx = nx; p.pad = pad;
    p.Nz = nz + 2 * pad; p.Ny = ny + 2 * pad; p.Nx = nx + 2 * pad;
    p.EY = (int)tomo_ext_rows(ny, pad);
    p.EWX32 = 2 * (int)tomo_ext_words_per_row(nx, pad);
    p.pitch = tomo_field_pitch(nx, pad);
    p.xorg = tomo_field_xorg(pad);
    p.nq = (nx + 3) / 4;
    p.r = nx % 4;
    p.nlanes = p.nq + (p.r ? 1 : 0);
    p.rows_per_block = 32;
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
    dim3 grid(gx, (unsigned)ceil_div64(p.Ny, p.rows_per_block), (unsigned)p.Nz);
    if (threads <= 256)
        hipLaunchKernelGGL(field_gauss_kernel<256>, grid, dim3(threads), 0, s, (const u32 *)ext, field, p);
    else
        hipLaunchKernelGGL(field_gauss_kernel<1024>, grid, dim3(threads), 0, s, (const u32 *)ext, field, p);
    return tomo_status();
}
