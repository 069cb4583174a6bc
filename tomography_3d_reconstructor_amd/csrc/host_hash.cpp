// host_hash.cpp -- full-content, position-dependent 128-bit checksum of a HOST buffer on several threads.
//
// Used by the device-side volume cache (_devcache.py): a cached device copy of a host array the caller can still
// write to is only trusted after EVERY byte of the array has been compared with what was uploaded (the reference reads
// the array it is handed: voxel_processor.py:84, surface_extractor.py:43-46) -- a sampled checksum misses in-place
// edits.  Not cryptographic; built so that any single-bit change, any swap and any shift of content changes the value:
// 64-bit words are mixed with a key that advances with the position, multiplied 32x32->64 (both halves) and accumulated in 8 lanes per
// stripe with the raw word added to the neighbouring lane; lanes are scrambled per 1 KiB block, blocks of 1 MiB are
// digested independently (that is what the threads split) and the digests are folded IN ORDER.
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/tomo_hip.h"

#define TOMO_API extern "C" __attribute__((visibility("default")))

namespace {

const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull;
const int64_t CHUNK = 1 << 20;   // bytes per independently digested chunk
const int LANES = 8;

inline uint64_t rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t avalanche(uint64_t h)
{
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}

// digest of one chunk (n bytes, n <= CHUNK); `index` makes equal chunks at different places differ

void chunk_digest(const uint8_t *p, int64_t n, uint64_t index, uint64_t out[2])
{
    uint64_t acc[LANES], key[LANES];
    for (int j = 0; j < LANES; j++) {
        acc[j] = P1 * (uint64_t)(j + 1) + index * P2;
        key[j] = avalanche(P3 * (uint64_t)(j + 1) ^ 0x5851F42D4C957F2Dull);
    }
    int64_t nstripes = n / (8 * LANES);
    const uint8_t *q = p;
    for (int64_t s = 0; s < nstripes; s++, q += 8 * LANES) {
        uint64_t w[LANES];
        memcpy(w, q, sizeof w);
        for (int j = 0; j < LANES; j++) {
            uint64_t d = w[j] ^ key[j];
            acc[j ^ 1] += w[j];
            acc[j] += (d & 0xFFFFFFFFull) * (d >> 32);
            key[j] += P1;                                     // the key moves with the position: swapped words do not cancel
        }
        if ((s & 15) == 15)                                   // every 1 KiB: scramble the lanes
            for (int j = 0; j < LANES; j++) acc[j] = (acc[j] ^ (acc[j] >> 47) ^ key[j]) * 0x9E3779B1ull;
    }
    uint64_t tail[LANES] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t rest = n - nstripes * 8 * LANES;
    if (rest) {
        memcpy(tail, q, (size_t)rest);
        for (int j = 0; j < LANES; j++) {
            uint64_t d = tail[j] ^ key[j];
            acc[j ^ 1] += tail[j];
            acc[j] += (d & 0xFFFFFFFFull) * (d >> 32);
        }
    }
    uint64_t h0 = (uint64_t)n * P1, h1 = (uint64_t)n * P2 + index;
    for (int j = 0; j < LANES; j++) {
        h0 = rotl(h0 ^ avalanche(acc[j]), 27) * P1 + P3;
        h1 = rotl(h1 + avalanche(acc[j] + key[j]), 31) * P2 + P1;
    }
    out[0] = avalanche(h0);
    out[1] = avalanche(h1);
}

}  // namespace

// h_out[0..1] = checksum of h_data[0 .. nbytes).  nthreads <= 0: one thread.  Returns 0 or TOMO_E_ARG.
TOMO_API int tomo_host_checksum(const void *h_data, int64_t nbytes, int nthreads, uint64_t *h_out)
{
    if (!h_out || nbytes < 0 || (nbytes > 0 && !h_data)) return TOMO_E_ARG;
    const uint8_t *p = (const uint8_t *)h_data;
    int64_t nchunks = (nbytes + CHUNK - 1) / CHUNK;
    std::vector<uint64_t> dig((size_t)nchunks * 2);
    if (nthreads < 1) nthreads = 1;
    if ((int64_t)nthreads > nchunks) nthreads = (int)(nchunks > 0 ? nchunks : 1);
    auto work = [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; c++) {
            int64_t off = c * CHUNK, n = nbytes - off < CHUNK ? nbytes - off : CHUNK;
            chunk_digest(p + off, n, (uint64_t)c, &dig[(size_t)c * 2]);
        }
    };
    if (nthreads == 1) {
        work(0, nchunks);
    } else {
        std::vector<std::thread> th;
        int64_t per = (nchunks + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; t++) {
            int64_t c0 = t * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
            if (c0 < c1) th.emplace_back(work, c0, c1);
        }
        for (auto &t : th) t.join();
    }
    uint64_t h0 = (uint64_t)nbytes ^ P3, h1 = (uint64_t)nbytes * P1;
    for (int64_t c = 0; c < nchunks; c++) {
        h0 = rotl(h0, 29) * P1 + dig[(size_t)c * 2];
        h1 = (rotl(h1, 23) ^ dig[(size_t)c * 2 + 1]) * P2;
    }
    h_out[0] = avalanche(h0);
    h_out[1] = avalanche(h1 ^ h0);
    return TOMO_OK;
}

// Bring the pages of a freshly allocated HOST buffer in on `nthreads` threads (one byte written per 4 KiB page; the buffer's
// content is unspecified before and after -- meant for np.empty arrays that a download is about to fill).  A fresh
// 1 GiB array costs ~65 ms of page faults when ONE thread touches it first (whoever does: memcpy, the DMA engine's pinning
// pass, np.ones); the faults of different pages run in parallel.  Returns 0 or TOMO_E_ARG.
TOMO_API int tomo_host_touch(void *h_data, int64_t nbytes, int nthreads)
{
    if (nbytes < 0 || (nbytes > 0 && !h_data)) return TOMO_E_ARG;
    if (nbytes == 0) return TOMO_OK;
    volatile uint8_t *p = (volatile uint8_t *)h_data;
    const int64_t PAGE = 4096;
    const int64_t npages = (nbytes + PAGE - 1) / PAGE;
    if (nthreads < 1) nthreads = 1;
    if ((int64_t)nthreads > (npages + 255) / 256) nthreads = (int)((npages + 255) / 256);
    auto work = [&](int64_t a, int64_t b) {
        for (int64_t g = a; g < b; g++) p[g * PAGE] = 0;
    };
    if (nthreads <= 1) {
        work(0, npages);
    } else {
        std::vector<std::thread> th;
        const int64_t per = (npages + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; t++) {
            const int64_t a = t * per, b = a + per < npages ? a + per : npages;
            if (a < b) th.emplace_back(work, a, b);
        }
        for (auto &t : th) t.join();
    }
    p[nbytes - 1] = 0;
    return TOMO_OK;
}

// np.stack on several threads: h_dst[i * bytes_each ...] = the bytes of h_src[i], i = 0 .. n-1 (n separate host buffers of
// bytes_each bytes -- the reference stacks its mask images with np.stack, voxel_processor.py:46, one thread, 85 ms for 1024
// masks of 1 MiB on the MI355X host).  The destination may be fresh memory: every thread brings the pages of its own share in.
TOMO_API int tomo_host_gather(const void *const *h_src, int64_t n, int64_t bytes_each, void *h_dst, int nthreads)
{
    if (n < 0 || bytes_each < 0 || (n > 0 && bytes_each > 0 && (!h_src || !h_dst))) return TOMO_E_ARG;
    if (n == 0 || bytes_each == 0) return TOMO_OK;
    for (int64_t i = 0; i < n; i++) if (!h_src[i]) return TOMO_E_ARG;
    uint8_t *dst = (uint8_t *)h_dst;
    // the work is cut into pieces of <= 1 MiB, dealt out contiguously: a thread writes one contiguous run of the destination
    const int64_t PIECE = 1 << 20;
    const int64_t per_src = (bytes_each + PIECE - 1) / PIECE, pieces = n * per_src;
    if (nthreads < 1) nthreads = 1;
    if ((int64_t)nthreads > pieces) nthreads = (int)pieces;
    auto work = [&](int64_t a, int64_t b) {
        for (int64_t q = a; q < b; q++) {
            const int64_t i = q / per_src, off = (q - i * per_src) * PIECE;
            const int64_t len = bytes_each - off < PIECE ? bytes_each - off : PIECE;
            memcpy(dst + i * bytes_each + off, (const uint8_t *)h_src[i] + off, (size_t)len);
        }
    };
    if (nthreads == 1) {
        work(0, pieces);
    } else {
        std::vector<std::thread> th;
        const int64_t per = (pieces + nthreads - 1) / nthreads;
        for (int t = 0; t < nthreads; t++) {
            const int64_t a = t * per, b = a + per < pieces ? a + per : pieces;
            if (a < b) th.emplace_back(work, a, b);
        }
        for (auto &t : th) t.join();
    }
    return TOMO_OK;
}
