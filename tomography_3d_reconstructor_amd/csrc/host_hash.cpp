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
#if defined(__x86_64__)
#include <immintrin.h>
#endif
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

#if defined(__x86_64__)
// the stripes of a chunk on AVX2: the 8 lanes are two vectors of four 64-bit words; the same wrapping arithmetic as the scalar
// loop below (adds commute), so the digest does not depend on which one ran.  -> stripes done
__attribute__((target("avx2"))) int64_t chunk_stripes_avx2(const uint8_t *q, int64_t nstripes, uint64_t acc[LANES], uint64_t key[LANES])
{
    __m256i a0 = _mm256_loadu_si256((const __m256i *)&acc[0]), a1 = _mm256_loadu_si256((const __m256i *)&acc[4]);
    __m256i k0 = _mm256_loadu_si256((const __m256i *)&key[0]), k1 = _mm256_loadu_si256((const __m256i *)&key[4]);
    const __m256i p1 = _mm256_set1_epi64x((long long)P1), c = _mm256_set1_epi64x(0x9E3779B1ll);
    for (int64_t s = 0; s < nstripes; s++, q += 8 * LANES) {
        const __m256i w0 = _mm256_loadu_si256((const __m256i *)q), w1 = _mm256_loadu_si256((const __m256i *)(q + 32));
        const __m256i d0 = _mm256_xor_si256(w0, k0), d1 = _mm256_xor_si256(w1, k1);
        a0 = _mm256_add_epi64(a0, _mm256_permute4x64_epi64(w0, 0xB1));              // acc[j ^ 1] += w[j]
        a1 = _mm256_add_epi64(a1, _mm256_permute4x64_epi64(w1, 0xB1));
        a0 = _mm256_add_epi64(a0, _mm256_mul_epu32(d0, _mm256_srli_epi64(d0, 32)));  // acc[j] += lo32(d) * hi32(d)
        a1 = _mm256_add_epi64(a1, _mm256_mul_epu32(d1, _mm256_srli_epi64(d1, 32)));
        k0 = _mm256_add_epi64(k0, p1);
        k1 = _mm256_add_epi64(k1, p1);
        if ((s & 15) == 15) {                                                        // acc = (acc ^ (acc >> 47) ^ key) * 0x9E3779B1
            __m256i t0 = _mm256_xor_si256(_mm256_xor_si256(a0, _mm256_srli_epi64(a0, 47)), k0);
            __m256i t1 = _mm256_xor_si256(_mm256_xor_si256(a1, _mm256_srli_epi64(a1, 47)), k1);
            a0 = _mm256_add_epi64(_mm256_mul_epu32(t0, c), _mm256_slli_epi64(_mm256_mul_epu32(_mm256_srli_epi64(t0, 32), c), 32));
            a1 = _mm256_add_epi64(_mm256_mul_epu32(t1, c), _mm256_slli_epi64(_mm256_mul_epu32(_mm256_srli_epi64(t1, 32), c), 32));
        }
    }
    _mm256_storeu_si256((__m256i *)&acc[0], a0); _mm256_storeu_si256((__m256i *)&acc[4], a1);
    _mm256_storeu_si256((__m256i *)&key[0], k0); _mm256_storeu_si256((__m256i *)&key[4], k1);
    return nstripes;
}
bool have_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
#endif

void chunk_digest(const uint8_t *p, int64_t n, uint64_t index, uint64_t out[2], int impl)
{
    uint64_t acc[LANES], key[LANES];
    for (int j = 0; j < LANES; j++) {
        acc[j] = P1 * (uint64_t)(j + 1) + index * P2;
        key[j] = avalanche(P3 * (uint64_t)(j + 1) ^ 0x5851F42D4C957F2Dull);
    }
    int64_t nstripes = n / (8 * LANES);
    const uint8_t *q = p;
    int64_t s = 0;
#if defined(__x86_64__)
    if (impl != 1 && have_avx2()) { s = chunk_stripes_avx2(q, nstripes, acc, key); q += s * 8 * LANES; }
#endif
    for (; s < nstripes; s++, q += 8 * LANES) {
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
// impl: 0 = the fastest the CPU offers (AVX2), 1 = the portable loop; same digest (tests compare them)
TOMO_API int tomo_host_checksum_impl(const void *h_data, int64_t nbytes, int nthreads, int impl, uint64_t *h_out);
TOMO_API int tomo_host_checksum_part(const void *h_part, int64_t nbytes, int64_t first_chunk, int nthreads, int impl, uint64_t *h_dig);
TOMO_API int tomo_host_checksum_fold(const uint64_t *h_dig, int64_t nchunks, int64_t nbytes, uint64_t *h_out);
TOMO_API int tomo_host_checksum(const void *h_data, int64_t nbytes, int nthreads, uint64_t *h_out)
{
    return tomo_host_checksum_impl(h_data, nbytes, nthreads, 0, h_out);
}
// The checksum in two steps, for a caller that receives the buffer piece by piece (a download in 128 MiB pieces: piece k is
// digested while piece k + 1 is on the bus): the digests of the 1 MiB chunks of a PART that starts at a chunk boundary
// (first_chunk = its offset / 1 MiB; every part but the last a whole number of chunks) -> h_dig[2 per chunk]; then the fold of all
// chunk digests of the buffer, in order.  tomo_host_checksum = one part + the fold.
TOMO_API int64_t tomo_host_checksum_chunk_bytes(void) { return CHUNK; }

TOMO_API int tomo_host_checksum_part(const void *h_part, int64_t nbytes, int64_t first_chunk, int nthreads, int impl, uint64_t *h_dig)
{
    if (nbytes < 0 || first_chunk < 0 || (nbytes > 0 && (!h_part || !h_dig))) return TOMO_E_ARG;
    const uint8_t *p = (const uint8_t *)h_part;
    const int64_t nchunks = (nbytes + CHUNK - 1) / CHUNK;
    if (nthreads < 1) nthreads = 1;
    if ((int64_t)nthreads > nchunks) nthreads = (int)(nchunks > 0 ? nchunks : 1);
    auto work = [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; c++) {
            int64_t off = c * CHUNK, n = nbytes - off < CHUNK ? nbytes - off : CHUNK;
            chunk_digest(p + off, n, (uint64_t)(first_chunk + c), &h_dig[(size_t)c * 2], impl);
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
    return TOMO_OK;
}

TOMO_API int tomo_host_checksum_fold(const uint64_t *h_dig, int64_t nchunks, int64_t nbytes, uint64_t *h_out)
{
    if (!h_out || nbytes < 0 || nchunks != (nbytes + CHUNK - 1) / CHUNK || (nchunks > 0 && !h_dig)) return TOMO_E_ARG;
    uint64_t h0 = (uint64_t)nbytes ^ P3, h1 = (uint64_t)nbytes * P1;
    for (int64_t c = 0; c < nchunks; c++) {
        h0 = rotl(h0, 29) * P1 + h_dig[(size_t)c * 2];
        h1 = (rotl(h1, 23) ^ h_dig[(size_t)c * 2 + 1]) * P2;
    }
    h_out[0] = avalanche(h0);
    h_out[1] = avalanche(h1 ^ h0);
    return TOMO_OK;
}

TOMO_API int tomo_host_checksum_impl(const void *h_data, int64_t nbytes, int nthreads, int impl, uint64_t *h_out)
{
    if (!h_out || nbytes < 0 || (nbytes > 0 && !h_data)) return TOMO_E_ARG;
    const int64_t nchunks = (nbytes + CHUNK - 1) / CHUNK;
    std::vector<uint64_t> dig((size_t)nchunks * 2 + 2);
    int rc = tomo_host_checksum_part(h_data, nbytes, 0, nthreads, impl, dig.data());
    if (rc != TOMO_OK) return rc;
    return tomo_host_checksum_fold(dig.data(), nchunks, nbytes, h_out);
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

// ---------------------------------------------------------------------------------------------------------------------
// SHA-256 (FIPS 180-4) with a caller-held, RELOCATABLE state.  hashlib cannot hand a running hash to another process; the
// Z-slab job needs exactly that: the digest of the WHOLE vertex / face list -- the bytes a single-GPU run returns, what
// the golden fixtures hold (the reference numbers its vertices globally: surface_extractor.py:115-126) -- without gathering
// the lists: rank 0 hashes its rows, the 112-byte state travels to rank 1, which continues with its rows, ...
// (slab.SlabJob.mesh_sha256).  State layout (little endian): u32 h[8] | u64 total bytes | u32 bytes buffered | u32 0 |
// u8 buffer[64].  A portable implementation plus the x86 SHA extensions where the CPU has them (same results).
namespace {

struct Sha256State {
    uint32_t h[8];
    uint64_t nbytes;
    uint32_t buflen, pad;
    uint8_t buf[64];
};
static_assert(sizeof(Sha256State) == 112, "state layout is part of the ABI");

const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

inline uint32_t rotr32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

void sha256_blocks_portable(uint32_t h[8], const uint8_t *p, int64_t nblocks)
{
    for (; nblocks > 0; nblocks--, p += 64) {
        uint32_t w[64];
        for (int i = 0; i < 16; i++)
            w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | (uint32_t)p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = rotr32(w[i - 15], 7) ^ rotr32(w[i - 15], 18) ^ (w[i - 15] >> 3);
            const uint32_t s1 = rotr32(w[i - 2], 17) ^ rotr32(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25), ch = (e & f) ^ (~e & g);
            const uint32_t t1 = hh + S1 + ch + SHA_K[i] + w[i];
            const uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22), maj = (a & b) ^ (a & c) ^ (b & c);
            const uint32_t t2 = S0 + maj;
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
}

#if defined(__x86_64__)
__attribute__((target("sha,sse4.1,ssse3"))) void sha256_blocks_shani(uint32_t h[8], const uint8_t *p, int64_t nblocks)
{
    const __m128i MASK = _mm_set_epi64x(0x0c0d0e0f08090a0bLL, 0x0405060700010203LL);
    __m128i tmp = _mm_loadu_si128((const __m128i *)&h[0]);          // DCBA
    __m128i st1 = _mm_loadu_si128((const __m128i *)&h[4]);          // HGFE
    tmp = _mm_shuffle_epi32(tmp, 0xB1);                             // CDAB
    st1 = _mm_shuffle_epi32(st1, 0x1B);                             // EFGH
    __m128i st0 = _mm_alignr_epi8(tmp, st1, 8);                     // ABEF
    st1 = _mm_blend_epi16(st1, tmp, 0xF0);                          // CDGH
    for (; nblocks > 0; nblocks--, p += 64) {
        const __m128i save0 = st0, save1 = st1;
        __m128i m[4], msg;
        for (int i = 0; i < 4; i++) m[i] = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)(p + 16 * i)), MASK);
        for (int r = 0; r < 16; r++) {                              // 16 groups of 4 rounds
            __m128i cur = m[r & 3];
            msg = _mm_add_epi32(cur, _mm_loadu_si128((const __m128i *)&SHA_K[4 * r]));
            st1 = _mm_sha256rnds2_epu32(st1, st0, msg);
            msg = _mm_shuffle_epi32(msg, 0x0E);
            st0 = _mm_sha256rnds2_epu32(st0, st1, msg);
            if (r < 12) {                                           // schedule w[4 (r + 4) .. 4 (r + 4) + 3] into m[r & 3]
                __m128i x = _mm_sha256msg1_epu32(m[r & 3], m[(r + 1) & 3]);
                x = _mm_add_epi32(x, _mm_alignr_epi8(m[(r + 3) & 3], m[(r + 2) & 3], 4));
                m[r & 3] = _mm_sha256msg2_epu32(x, m[(r + 3) & 3]);
            }
        }
        st0 = _mm_add_epi32(st0, save0);
        st1 = _mm_add_epi32(st1, save1);
    }
    tmp = _mm_shuffle_epi32(st0, 0x1B);                             // FEBA
    st1 = _mm_shuffle_epi32(st1, 0xB1);                             // DCHG
    st0 = _mm_blend_epi16(tmp, st1, 0xF0);                          // DCBA
    st1 = _mm_alignr_epi8(st1, tmp, 8);                             // HGFE
    _mm_storeu_si128((__m128i *)&h[0], st0);
    _mm_storeu_si128((__m128i *)&h[4], st1);
}
bool have_shani()
{
    static const bool yes = __builtin_cpu_supports("sha") && __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3");
    return yes;
}
#endif

void sha256_blocks(uint32_t h[8], const uint8_t *p, int64_t nblocks, int impl)
{
#if defined(__x86_64__)
    if (impl != 1 && have_shani()) { sha256_blocks_shani(h, p, nblocks); return; }
#endif
    sha256_blocks_portable(h, p, nblocks);
}

}  // namespace

// state: 112 bytes the caller holds (any alignment; may be copied, stored, sent to another process between calls).
TOMO_API int tomo_host_sha256_init(void *h_state)
{
    if (!h_state) return TOMO_E_ARG;
    Sha256State s;
    memset(&s, 0, sizeof s);
    const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s.h, iv, sizeof iv);
    memcpy(h_state, &s, sizeof s);
    return TOMO_OK;
}

// impl: 0 = the fastest the CPU offers, 1 = the portable code (tests compare the two)
TOMO_API int tomo_host_sha256_update(void *h_state, const void *h_data, int64_t nbytes, int impl)
{
    if (!h_state || nbytes < 0 || (nbytes > 0 && !h_data)) return TOMO_E_ARG;
    Sha256State s;
    memcpy(&s, h_state, sizeof s);
    if (s.buflen >= 64) return TOMO_E_ARG;
    const uint8_t *p = (const uint8_t *)h_data;
    int64_t n = nbytes;
    s.nbytes += (uint64_t)nbytes;
    if (s.buflen) {
        const int64_t take = n < 64 - (int64_t)s.buflen ? n : 64 - (int64_t)s.buflen;
        memcpy(s.buf + s.buflen, p, (size_t)take);
        s.buflen += (uint32_t)take; p += take; n -= take;
        if (s.buflen == 64) { sha256_blocks(s.h, s.buf, 1, impl); s.buflen = 0; }
    }
    if (n >= 64) {
        sha256_blocks(s.h, p, n / 64, impl);
        p += n / 64 * 64; n %= 64;
    }
    if (n) { memcpy(s.buf, p, (size_t)n); s.buflen = (uint32_t)n; }
    memcpy(h_state, &s, sizeof s);
    return TOMO_OK;
}

// the digest of everything hashed so far; the state is left as it is (more data may follow)
TOMO_API int tomo_host_sha256_digest(const void *h_state, uint8_t *h_digest32)
{
    if (!h_state || !h_digest32) return TOMO_E_ARG;
    Sha256State s;
    memcpy(&s, h_state, sizeof s);
    if (s.buflen >= 64) return TOMO_E_ARG;
    uint8_t tail[128];
    memset(tail, 0, sizeof tail);
    memcpy(tail, s.buf, s.buflen);
    tail[s.buflen] = 0x80;
    const int blocks = s.buflen + 9 <= 64 ? 1 : 2;
    const uint64_t bits = s.nbytes * 8;
    for (int i = 0; i < 8; i++) tail[blocks * 64 - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_blocks(s.h, tail, blocks, 1);
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 4; b++) h_digest32[4 * i + b] = (uint8_t)(s.h[i] >> (24 - 8 * b));
    return TOMO_OK;
}
