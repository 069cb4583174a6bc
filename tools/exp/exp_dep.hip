// micro-benchmark: VALU issue rate of one wave vs. number of independent dependency chains, 1 and 2 waves per SIMD (gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
template <int C, int DPP>
__global__ __launch_bounds__(256) void k(u32 *out, int iters, u32 kk)
{
    u32 a[C];
    for (int i = 0; i < C; i++) a[i] = threadIdx.x * 2654435761u + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / C; r++)
#pragma unroll
            for (int i = 0; i < C; i++) {
                if (DPP) a[i] = __builtin_amdgcn_alignbit((u32)__builtin_amdgcn_mov_dpp((int)a[i], 0x138, 0xf, 0xf, true), kk, 7);
                else a[i] = __builtin_amdgcn_alignbit(a[i], kk, 7);
            }
    }
    u32 s = 0;
    for (int i = 0; i < C; i++) s ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int C, int DPP> void run(u32 *d, int blocks_per_cu)
{
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<C, DPP>), dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<C, DPP>), dim3(blocks), dim3(256), 0, 0, d, iters, 12345u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)iters * 64 * (DPP ? 2 : 1);          // instructions per wave
    printf("chains %d %s waves/SIMD %d: %.3f ms -> %.2f cycles per instruction per wave (2.4 GHz), SIMD: %.2f\n", C, DPP ? "dpp+alignbit" : "alignbit    ",
           blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / n, ms * 1e-3 * 2.4e9 / n / blocks_per_cu);
}
int main()
{
    u32 *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1, 0>(d, 1); run<2, 0>(d, 1); run<4, 0>(d, 1); run<8, 0>(d, 1);
    run<1, 0>(d, 2); run<2, 0>(d, 2); run<4, 0>(d, 2); run<8, 0>(d, 2);
    run<8, 0>(d, 4); run<8, 0>(d, 8);
    run<1, 1>(d, 1); run<2, 1>(d, 1); run<4, 1>(d, 1); run<8, 1>(d, 1);
    run<1, 1>(d, 2); run<4, 1>(d, 2); run<8, 1>(d, 2); run<8, 1>(d, 8);
    return 0;
}
