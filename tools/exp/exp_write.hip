// experiment: the streaming-write ceiling (what bounds field_tile_kernel): plain / nontemporal 16-byte stores, one-shot blocks
// of U x 4 KiB or a persistent grid-stride loop
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int NT>
__global__ __launch_bounds__(256) void k_write_chunk(f4 *__restrict__ out, int64_t n16, float v)
{
    const int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x;
    const f4 val = {v, v, v, v};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + u * 256;
        if (i < n16) { if (NT) __builtin_nontemporal_store(val, out + i); else out[i] = val; }
    }
}

template <int NT>
__global__ __launch_bounds__(256) void k_write_persist(f4 *__restrict__ out, int64_t n16, float v)
{
    const f4 val = {v, v, v, v};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(val, out + i); else out[i] = val;
    }
}

// each wave writes whole 4 KiB "pages" (64 lanes x 16 B x 4 rows)
template <int NT>
__global__ __launch_bounds__(256) void k_write_wavepage(f4 *__restrict__ out, int64_t n16, float v)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const f4 val = {v, v, v, v};
    const int64_t base = wave * 256 + lane;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * 64;
        if (i < n16) { if (NT) __builtin_nontemporal_store(val, out + i); else out[i] = val; }
    }
}

extern "C" int exp_write(void *out, int64_t bytes, int mode, int U, int nt, int blocks, void *stream)
{
    const int64_t n16 = bytes / 16;
    hipStream_t s = (hipStream_t)stream;
    f4 *o = (f4 *)out;
    if (mode == 0) {
        const unsigned g = (unsigned)((n16 + 256 * U - 1) / (256 * U));
#define C(UU, NN) hipLaunchKernelGGL((k_write_chunk<UU, NN>), dim3(g), dim3(256), 0, s, o, n16, 1.0f)
        if (U == 1) { if (nt) C(1, 1); else C(1, 0); }
        else if (U == 2) { if (nt) C(2, 1); else C(2, 0); }
        else if (U == 4) { if (nt) C(4, 1); else C(4, 0); }
        else { if (nt) C(8, 1); else C(8, 0); }
    } else if (mode == 1) {
        if (nt) hipLaunchKernelGGL((k_write_persist<1>), dim3(blocks), dim3(256), 0, s, o, n16, 1.0f);
        else hipLaunchKernelGGL((k_write_persist<0>), dim3(blocks), dim3(256), 0, s, o, n16, 1.0f);
    } else {
        const unsigned g = (unsigned)((n16 + 1023) / 1024);
        if (nt) hipLaunchKernelGGL((k_write_wavepage<1>), dim3(g), dim3(256), 0, s, o, n16, 1.0f);
        else hipLaunchKernelGGL((k_write_wavepage<0>), dim3(g), dim3(256), 0, s, o, n16, 1.0f);
    }
    return (int)hipGetLastError();
}
