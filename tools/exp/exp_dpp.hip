// micro-benchmark: issue cost of DPP flavours on gfx950 (wave_shr:1 vs row_shr:1 vs plain mov vs bitop3)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
template <int MODE>
__global__ __launch_bounds__(256) void k(u32 *out, int iters)
{
    u32 a[8];
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (MODE == 0) a[i] = (u32)__builtin_amdgcn_mov_dpp((int)a[i], 0x138, 0xf, 0xf, true) ^ 1u;
            if (MODE == 1) a[i] = (u32)__builtin_amdgcn_mov_dpp((int)a[i], 0x111, 0xf, 0xf, true) ^ 1u;
            if (MODE == 2) a[i] = (a[i] >> 1) ^ 1u;
            if (MODE == 3) a[i] = (u32)__builtin_amdgcn_mov_dpp((int)a[i], 0x130, 0xf, 0xf, true) ^ 1u;
            if (MODE == 4) a[i] = (u32)__builtin_amdgcn_mov_dpp((int)a[i], 0x121, 0xf, 0xf, true) ^ 1u;   // row_ror:1
            if (MODE == 5) a[i] = (u32)__shfl_up((int)a[i], 1, 64) ^ 1u;
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, u32 *d)
{
    const int iters = 2000, blocks = 256 * 8;      // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per SIMD: 8 waves x iters x 32 pairs (2 instrs: dpp/shift + xor)
    double pairs = 8.0 * iters * 32;
    printf("%-10s %.3f ms  -> %.2f cycles per (op + xor) pair at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / pairs);
}
int main()
{
    u32 *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<2>("shift", d); run<0>("wave_shr", d); run<3>("wave_shl", d); run<1>("row_shr", d); run<4>("row_ror", d); run<5>("shfl_up", d);
    return 0;
}
