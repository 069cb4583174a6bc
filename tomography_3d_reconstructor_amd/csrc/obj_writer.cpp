// obj_writer.cpp -- host-side OBJ export of the final mesh (the consumer right after the hot path):
// obj_exporter.py:17-38 writes "v %.6f %.6f %.6f" / "f a+1 b+1 c+1" with a Python loop per vertex and face
// (minutes for the 3.5 M vertices / 7 M faces of a 1024^3 volume).  Same bytes here, formatted in parallel chunks.
//
// "%.6f" of a float is the exact decimal expansion of the binary value rounded half-to-even at the sixth decimal
// (what both CPython's format() and glibc's printf produce).  For |x| < 2^40 that is done exactly with 128-bit integer
// arithmetic (mantissa * 10^6, shifted, ties to even); anything else goes through snprintf.
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <thread>
#include <vector>
#include "../../include/tomo_hip.h"

#define TOMO_API extern "C" __attribute__((visibility("default")))

static inline char *put_u64(char *p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

// appends "%.6f" of x
static char *put_f6(char *p, double x)
{
    if (!(fabs(x) < 1099511627776.0)) {                 // inf, nan, |x| >= 2^40: rare, let libc do it (python spells nan without sign)
        if (x != x) { memcpy(p, "nan", 3); return p + 3; }
        if (isinf(x)) { if (x < 0) *p++ = '-'; memcpy(p, "inf", 3); return p + 3; }
        return p + snprintf(p, 400, "%.6f", x);
    }
    const bool neg = signbit(x);
    int e;
    const double fr = frexp(fabs(x), &e);                // |x| = fr * 2^e, fr in [0.5, 1) or 0
    const uint64_t m = (uint64_t)ldexp(fr, 53);          // 53-bit integer mantissa, exact
    int sh = 53 - e;                                     // |x| = m / 2^sh, sh >= 13 because |x| < 2^40
    unsigned __int128 n = (unsigned __int128)m * 1000000u;   // < 2^73
    uint64_t k;
    if (m == 0) k = 0;
    else if (sh > 120) k = 0;                            // far below half a unit of the sixth decimal
    else {
        const unsigned __int128 q = n >> sh, r = n - (q << sh), half = (unsigned __int128)1 << (sh - 1);
        k = (uint64_t)q;
        if (r > half || (r == half && (k & 1))) k++;
    }
    if (neg) *p++ = '-';                                 // python and printf both print "-0.000000" for tiny negatives and -0.0
    p = put_u64(p, k / 1000000u);
    *p++ = '.';
    uint32_t f = (uint32_t)(k % 1000000u);
    for (int i = 5; i >= 0; i--) { p[i] = (char)('0' + f % 10); f /= 10; }
    return p + 6;
}

template <typename T>
static void format_vertices(const T *v, int64_t a, int64_t b, std::string &out)
{
    out.resize((size_t)(b - a) * 3 * 48 + 16);
    char *p = &out[0];
    for (int64_t i = a; i < b; i++) {
        *p++ = 'v';
        for (int c = 0; c < 3; c++) { *p++ = ' '; p = put_f6(p, (double)v[3 * i + c]); }
        *p++ = '\n';
        if ((size_t)(p - &out[0]) + 1300 > out.size()) {                        // snprintf fallback lines can be long
            size_t used = (size_t)(p - &out[0]);
            out.resize(out.size() * 2 + 4096);
            p = &out[0] + used;
        }
    }
    out.resize((size_t)(p - &out[0]));
}

static void format_faces(const int64_t *f, int64_t a, int64_t b, std::string &out)
{
    out.resize((size_t)(b - a) * 66 + 16);
    char *p = &out[0];
    for (int64_t i = a; i < b; i++) {
        *p++ = 'f';
        for (int c = 0; c < 3; c++) {
            *p++ = ' ';
            int64_t x = f[3 * i + c] + 1;
            if (x < 0) { *p++ = '-'; p = put_u64(p, (uint64_t)(-x)); } else p = put_u64(p, (uint64_t)x);
        }
        *p++ = '\n';
    }
    out.resize((size_t)(p - &out[0]));
}

// vertices: nv x 3 (float32 if vertex_is_double == 0, else float64), faces: nf x 3 int64 (0-based), host memory.
// Returns 0, or -errno when the file cannot be written.
TOMO_API int tomo_obj_write(const char *path, const void *vertices, int vertex_is_double, int64_t nv, const int64_t *faces,
                            int64_t nf, int nthreads)
{
    if (!path || nv < 0 || nf < 0 || (nv > 0 && !vertices) || (nf > 0 && !faces)) return TOMO_E_ARG;
    FILE *fp = fopen(path, "w");
    if (!fp) return -errno;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    bool ok = fprintf(fp, "# Tomography reconstruction model\n# %lld vertices, %lld faces\n\n", (long long)nv, (long long)nf) > 0;
    const int64_t CH = 1 << 18;                                   // rows per work item
    for (int phase = 0; phase < 2 && ok; phase++) {
        const int64_t n = phase == 0 ? nv : nf;
        if (phase == 1) ok = fputc('\n', fp) != EOF;
        for (int64_t base = 0; base < n && ok; base += CH * nthreads) {
            int used = 0;
            std::vector<std::string> bufs((size_t)nthreads);
            std::vector<std::thread> th;
            for (int t = 0; t < nthreads; t++) {
                int64_t a = base + (int64_t)t * CH, b = a + CH < n ? a + CH : n;
                if (a >= n) break;
                used++;
                if (phase == 0) {
                    if (vertex_is_double) th.emplace_back(format_vertices<double>, (const double *)vertices, a, b, std::ref(bufs[t]));
                    else th.emplace_back(format_vertices<float>, (const float *)vertices, a, b, std::ref(bufs[t]));
                } else th.emplace_back(format_faces, faces, a, b, std::ref(bufs[t]));
            }
            for (auto &x : th) x.join();
            for (int t = 0; t < used && ok; t++) ok = fwrite(bufs[t].data(), 1, bufs[t].size(), fp) == bufs[t].size();
        }
    }
    int err = ok ? 0 : (errno ? -errno : -EIO);
    if (fclose(fp) != 0 && !err) err = errno ? -errno : -EIO;
    return err;
}
