// volume.hip -- the callers either side of the hot path that work on the resident bit volume:
//   * per-slice voxel counts and the bounding box (volume_calculator.py:23-35, 37-57, 59-94): the reference does
//     np.sum(voxel_data[z]) per slice and np.where(voxel_data) (three int64 index arrays, 24 B per set voxel);
//     here both are one streaming pass over 1 bit/voxel;
//   * `img >= threshold` of the mask loader (image_loader.py:108) fused with the bit packing, so the grey
//     stack that was uploaded is never written back as 1 B/voxel booleans.
#include "tomo_common.h"

// ------------------------------------------------------------------------------------------ per-slice counts
// grid (chunks, nz): every block sums its part of slice z and adds it to counts[z] (zeroed by the call)
__global__ __launch_bounds__(256) void slice_popcount_kernel(const u64 *__restrict__ bits, int64_t words_per_slice,
                                                             unsigned long long *__restrict__ counts)
{
    const u64 *sl = bits + (int64_t)blockIdx.y * words_per_slice;
    u64 acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words_per_slice; i += (int64_t)gridDim.x * blockDim.x)
        acc += (u64)__popcll(sl[i]);
    acc = wave_sum64(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(&counts[blockIdx.y], (unsigned long long)acc);
}

TOMO_API int tomo_slice_popcounts(const uint64_t *bits, int nz, int ny, int nx, unsigned long long *counts, void *stream)
{
    if (!bits || !counts || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    if (nz > 65535) return TOMO_E_SIZE;
    int64_t wps = (int64_t)ny * tomo_words_per_row(nx);
    if (hipMemsetAsync(counts, 0, (size_t)nz * sizeof(unsigned long long), (hipStream_t)stream) != hipSuccess)
        return TOMO_E_LAUNCH;
    int64_t chunks = ceil_div64(wps, 256 * 16);
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(slice_popcount_kernel, dim3((unsigned)chunks, (unsigned)nz), dim3(256), 0, (hipStream_t)stream,
                       (const u64 *)bits, wps, counts);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ bounding box
// box = {zmin, zmax, ymin, ymax, xmin, xmax} as int32; an empty volume leaves {INT_MAX, -1, INT_MAX, -1, INT_MAX, -1}
__global__ void bbox_init_kernel(int *box)
{
    if (threadIdx.x < 6) box[threadIdx.x] = (threadIdx.x & 1) ? -1 : 0x7fffffff;
}

__device__ static inline int wave_min_i(int v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o < v ? o : v; }
    return v;
}
__device__ static inline int wave_max_i(int v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { int o = __shfl_xor(v, d, 64); v = o > v ? o : v; }
    return v;
}

__global__ __launch_bounds__(256) void bbox_kernel(const u64 *__restrict__ bits, int64_t nwords, int ny, int wx,
                                                   int *__restrict__ box)
{
    int zmin = 0x7fffffff, zmax = -1, ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 w = bits[i];
        if (!w) continue;
        const int64_t row = i / wx;
        const int wi = (int)(i - row * wx);
        const int z = (int)(row / ny), y = (int)(row - (int64_t)z * ny);
        const int xl = wi * 64 + (__ffsll((long long)w) - 1), xh = wi * 64 + 63 - __clzll((long long)w);
        zmin = z < zmin ? z : zmin; zmax = z > zmax ? z : zmax;
        ymin = y < ymin ? y : ymin; ymax = y > ymax ? y : ymax;
        xmin = xl < xmin ? xl : xmin; xmax = xh > xmax ? xh : xmax;
    }
    zmin = wave_min_i(zmin); ymin = wave_min_i(ymin); xmin = wave_min_i(xmin);
    zmax = wave_max_i(zmax); ymax = wave_max_i(ymax); xmax = wave_max_i(xmax);
    if ((threadIdx.x & 63) == 0 && zmax >= 0) {
        atomicMin(&box[0], zmin); atomicMax(&box[1], zmax);
        atomicMin(&box[2], ymin); atomicMax(&box[3], ymax);
        atomicMin(&box[4], xmin); atomicMax(&box[5], xmax);
    }
}

TOMO_API int tomo_bbox(const uint64_t *bits, int nz, int ny, int nx, int32_t *box, void *stream)
{
    if (!bits || !box || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    const int wx = (int)tomo_words_per_row(nx);
    const int64_t nwords = (int64_t)nz * ny * wx;
    hipLaunchKernelGGL(bbox_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, box);
    int64_t blocks = ceil_div64(nwords, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(bbox_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u64 *)bits, nwords, ny,
                       wx, box);
    return tomo_status();
}

// ------------------------------------------------------------------------------------------ threshold + pack
// bits = (grey >= threshold): same lane layout as pack16_kernel / pack_kernel of bits.hip.
__device__ static inline u32 ge_nibble(u32 w, u32 k2)
{   // bit i of the result = (byte i of w >= t), k2 = (256 - t) * 0x00010001: the compare is the carry out of byte + (256 - t)
    const u32 ce = (((w & 0x00ff00ffu) + k2) >> 8) & 0x00010001u;          // bytes 0, 2
    const u32 co = ((((w >> 8) & 0x00ff00ffu) + k2) >> 8) & 0x00010001u;   // bytes 1, 3
    return (ce & 1u) | ((co & 1u) << 1) | ((ce >> 16) << 2) | ((co >> 16) << 3);
}

__global__ __launch_bounds__(256) void pack16_threshold_kernel(const uint8_t *__restrict__ grey, u64 *__restrict__ bits,
                                                               int64_t rows, int nx, int wx, int groups, u32 k2)
{
    const int lane = threadIdx.x & 63;
    int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= rows * groups) return;
    int64_t row = wid / groups;
    int g = (int)(wid - row * groups);
    int x = g * 1024 + lane * 16;
    u32 piece = 0;
    if (x < nx) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_nontemporal_load((const u4 *)(grey + row * (int64_t)nx + x));   // read once, keep it out of L2
        piece = ge_nibble(v.x, k2) | (ge_nibble(v.y, k2) << 4) | (ge_nibble(v.z, k2) << 8) | (ge_nibble(v.w, k2) << 12);
    }
    u64 w = (u64)piece << (16 * (lane & 3));
    w |= __shfl_xor(w, 1, 64);
    w |= __shfl_xor(w, 2, 64);
    int word = g * 16 + (lane >> 2);
    if ((lane & 3) == 0 && word < wx) bits[row * (int64_t)wx + word] = w;
}

__global__ __launch_bounds__(256) void pack_threshold_kernel(const uint8_t *__restrict__ grey, u64 *__restrict__ bits,
                                                             int64_t rows, int nx, int wx, int groups, int threshold)
{
    const int lane = threadIdx.x & 63;
    int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wid >= rows * groups) return;
    int64_t row = wid / groups;
    int g = (int)(wid - row * groups);
    const uint8_t *src = grey + row * (int64_t)nx;
    u64 mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int x = (g * 16 + k) * 64 + lane;
        int b = x < nx ? (int)src[x] : -1;
        u64 m = __ballot(x < nx && b >= threshold);
        if (lane == k) mine = m;
    }
    int w = g * 16 + lane;
    if (lane < 16 && w < wx) bits[row * (int64_t)wx + w] = mine;
}

TOMO_API int tomo_pack_threshold(const uint8_t *grey, uint64_t *bits, int nz, int ny, int nx, int threshold, void *stream)
{
    if (!grey || !bits || nz <= 0 || ny <= 0 || nx <= 0) return TOMO_E_ARG;
    int wx = (int)tomo_words_per_row(nx);
    int groups = (wx + 15) / 16;
    int64_t rows = (int64_t)nz * ny;
    int64_t blocks = ceil_div64(rows * groups, 4);
    if (blocks > 0x7fffffff) return TOMO_E_SIZE;
    if (threshold > 255) {                                   // uint8 >= t is never true
        if (hipMemsetAsync(bits, 0, (size_t)rows * wx * sizeof(u64), (hipStream_t)stream) != hipSuccess) return TOMO_E_LAUNCH;
        return TOMO_OK;
    }
    if (threshold < 0) threshold = 0;
    if (nx % 16 == 0 && (((uintptr_t)grey) & 15) == 0)
        hipLaunchKernelGGL(pack16_threshold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grey,
                           (u64 *)bits, rows, nx, wx, groups, (u32)(256 - threshold) * 0x00010001u);
    else
        hipLaunchKernelGGL(pack_threshold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grey,
                           (u64 *)bits, rows, nx, wx, groups, threshold);
    return tomo_status();
}
