// experiment (round 3): one-shot 4 KiB store workgroups whose value / permission comes from a small class map --
// through a scalar load (uniform address), through a per-thread vector load, or not at all (baseline).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: no load.  1: one byte per workgroup, uniform address (scalar load).  2: one byte per thread (vector load, same byte for
// 8 consecutive threads).  3: as 1, but two 64-bit words per workgroup (s_load_dwordx4) -- room for per-tile classes.
template <int MODE>
__global__ __launch_bounds__(256) void k_row(f4 *__restrict__ out, int64_t n16, const uint8_t *__restrict__ map, int per_entry_wgs)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v = 1.0f;
    bool on = true;
    if (MODE == 1) {
        const uint8_t c = map[blockIdx.x / per_entry_wgs];
        on = c != 0; v = c == 2 ? 0.96875f : 0.0f;
    } else if (MODE == 2) {
        const uint8_t c = map[(blockIdx.x / per_entry_wgs) * 32 + (threadIdx.x >> 3)];
        on = c != 0; v = c == 2 ? 0.96875f : 0.0f;
    } else if (MODE == 3) {
        const uint64_t *m = (const uint64_t *)map + 2 * (int64_t)(blockIdx.x / per_entry_wgs);
        const uint64_t a = m[0], b = m[1];
        const int t = threadIdx.x >> 3;
        on = (a >> t) & 1; v = ((b >> t) & 1) ? 0.96875f : 0.0f;
    }
    if (on && i < n16) __builtin_nontemporal_store((f4){v, v, v, v}, out + i);
}

extern "C" int exp_row(void *out, int64_t bytes, int mode, const void *map, int per_entry_wgs, void *stream)
{
    const int64_t n16 = bytes / 16;
    const unsigned g = (unsigned)((n16 + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    f4 *o = (f4 *)out;
    const uint8_t *m = (const uint8_t *)map;
    if (mode == 0) hipLaunchKernelGGL((k_row<0>), dim3(g), dim3(256), 0, s, o, n16, m, per_entry_wgs);
    else if (mode == 1) hipLaunchKernelGGL((k_row<1>), dim3(g), dim3(256), 0, s, o, n16, m, per_entry_wgs);
    else if (mode == 2) hipLaunchKernelGGL((k_row<2>), dim3(g), dim3(256), 0, s, o, n16, m, per_entry_wgs);
    else hipLaunchKernelGGL((k_row<3>), dim3(g), dim3(256), 0, s, o, n16, m, per_entry_wgs);
    return (int)hipGetLastError();
}
