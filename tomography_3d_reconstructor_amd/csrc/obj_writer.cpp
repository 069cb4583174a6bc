// obj_writer.cpp -- host-side OBJ export of the final mesh (the consumer right after the hot path):
// obj_exporter.py:17-38 writes "v %.6f %.6f %.6f" / "f a+1 b+1 c+1" with a Python loop per vertex and face
// (minutes for the 3.5 M vertices / 7 M faces of a 1024^3 volume).  Same bytes here, formatted in parallel chunks.
//
// "%.6f" of a float is the exact decimal expansion of the binary value rounded half-to-even at the sixth decimal
// (what both CPython's format() and glibc's printf produce).  For |x| < 2^40 that is done exactly with 128-bit integer
// arithmetic (mantissa * 10^6, shifted, ties to even); anything else goes through snprintf.
#include <errno.h>
#include <fcntl.h>
#include <unistd.h>
#include <new>
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

// Formats rows [0, n) of one block (kind 0: float32 vertices, 1: float64 vertices, 2: int64 faces) on `nthreads` threads
// into `chunks` (in row order).
static void format_block(int kind, const void *rows, int64_t n, int nthreads, std::vector<std::string> &chunks)
{
    const int64_t CH = 1 << 18;                                   // rows per work item
    const int64_t nch = (n + CH - 1) / CH;
    chunks.assign((size_t)nch, std::string());
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    for (int64_t base = 0; base < nch; base += nthreads) {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads && base + t < nch; t++) {
            const int64_t a = (base + t) * CH, b = a + CH < n ? a + CH : n;
            std::string &out = chunks[(size_t)(base + t)];
            if (kind == 0) th.emplace_back(format_vertices<float>, (const float *)rows, a, b, std::ref(out));
            else if (kind == 1) th.emplace_back(format_vertices<double>, (const double *)rows, a, b, std::ref(out));
            else th.emplace_back(format_faces, (const int64_t *)rows, a, b, std::ref(out));
        }
        for (auto &x : th) x.join();
    }
}

// vertices: nv x 3 (float32 if vertex_is_double == 0, else float64), faces: nf x 3 int64 (0-based), host memory.
// Returns 0, or -errno when the file cannot be written.
TOMO_API int tomo_obj_write(const char *path, const void *vertices, int vertex_is_double, int64_t nv, const int64_t *faces,
                            int64_t nf, int nthreads)
{
    if (!path || nv < 0 || nf < 0 || (nv > 0 && !vertices) || (nf > 0 && !faces)) return TOMO_E_ARG;
    FILE *fp = fopen(path, "w");
    if (!fp) return -errno;
    bool ok = fprintf(fp, "# Tomography reconstruction model\n# %lld vertices, %lld faces\n\n", (long long)nv, (long long)nf) > 0;
    const int64_t SLICE = (int64_t)64 << 18;                      // rows formatted before they are written (bounds the memory)
    for (int phase = 0; phase < 2 && ok; phase++) {
        const int64_t n = phase == 0 ? nv : nf;
        if (phase == 1) ok = fputc('\n', fp) != EOF;
        for (int64_t base = 0; base < n && ok; base += SLICE) {
            const int64_t m = base + SLICE < n ? SLICE : n - base;
            std::vector<std::string> chunks;
            if (phase == 0 && vertex_is_double) format_block(1, (const double *)vertices + 3 * base, m, nthreads, chunks);
            else if (phase == 0) format_block(0, (const float *)vertices + 3 * base, m, nthreads, chunks);
            else format_block(2, faces + 3 * base, m, nthreads, chunks);
            for (size_t c = 0; c < chunks.size() && ok; c++) ok = fwrite(chunks[c].data(), 1, chunks[c].size(), fp) == chunks[c].size();
        }
    }
    int err = ok ? 0 : (errno ? -errno : -EIO);
    if (fclose(fp) != 0 && !err) err = errno ? -errno : -EIO;
    return err;
}

// ---- one OBJ file written by several processes (a Z-slab job: every rank holds a run of the vertex list and of the
// face list).  A rank formats its block in memory (tomo_obj_block_format -> handle + size), the sizes are exchanged, and
// every rank writes its bytes at its offset of the shared file (tomo_obj_block_pwrite).  Same formatter as above, so
// the file is byte for byte the one tomo_obj_write produces from the gathered mesh.
struct ObjBlock { std::vector<std::string> chunks; int64_t bytes; };

TOMO_API int tomo_obj_block_format(int kind, const void *h_rows, int64_t n, int nthreads, void **h_block, int64_t *h_nbytes)
{
    if (!h_block || !h_nbytes || kind < 0 || kind > 2 || n < 0 || (n > 0 && !h_rows)) return TOMO_E_ARG;
    ObjBlock *b = new (std::nothrow) ObjBlock();
    if (!b) return -ENOMEM;
    try {
        format_block(kind, h_rows, n, nthreads, b->chunks);
    } catch (...) {
        delete b;
        return -ENOMEM;
    }
    b->bytes = 0;
    for (auto &c : b->chunks) b->bytes += (int64_t)c.size();
    *h_block = b;
    *h_nbytes = b->bytes;
    return TOMO_OK;
}

// Writes the block's bytes at `offset` of an EXISTING file (it is not truncated).  Returns 0 or -errno.
TOMO_API int tomo_obj_block_pwrite(const char *path, int64_t offset, const void *h_block)
{
    if (!path || !h_block || offset < 0) return TOMO_E_ARG;
    const ObjBlock *b = (const ObjBlock *)h_block;
    int fd = open(path, O_WRONLY);
    if (fd < 0) return -errno;
    int err = 0;
    int64_t off = offset;
    for (size_t c = 0; c < b->chunks.size() && !err; c++) {
        const char *p = b->chunks[c].data();
        size_t left = b->chunks[c].size();
        while (left) {
            ssize_t w = pwrite(fd, p, left, (off_t)off);
            if (w < 0) { if (errno == EINTR) continue; err = -errno; break; }
            p += w; left -= (size_t)w; off += w;
        }
    }
    if (close(fd) != 0 && !err) err = -errno;
    return err;
}

TOMO_API void tomo_obj_block_free(void *h_block) { delete (ObjBlock *)h_block; }
