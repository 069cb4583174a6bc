// experiment: how fast can a block-structured store kernel with the field kernel's geometry go?
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef unsigned int u32;
extern __shared__ u32 s_bits[];
// mode bit0: stage loads, bit1: second barrier + fake wconst, bit2: persistent over ychunks
template <int ROWS>
__global__ __launch_bounds__(256) void k_store(const u32* __restrict__ ext32, float* __restrict__ field, int Ny, int64_t pitch,
                                               int EY, int EWX32, int mode, int coff, unsigned long long* sig)
{
    const int tid = threadIdx.x;
    const int Z = blockIdx.z, Y0 = blockIdx.y * ROWS;
    const int Y1 = Y0 + ROWS < Ny ? Y0 + ROWS : Ny;
    const int nrows = Y1 - Y0 + 4, W = 34, RS = (ROWS + 4) * W;
    u32 acc = 0;
    if (mode & 1) {
        if (tid < 5 * W) {
            const int64_t sstride = (int64_t)EY * EWX32;
            const int k = tid / W, w = tid - k * W;
            const u32* sp = ext32 + (int64_t)Z * sstride + (int64_t)Y0 * EWX32 + (int64_t)k * sstride + w;
            u32 v[ROWS + 4];
#pragma unroll
            for (int yy = 0; yy < ROWS + 4; yy++) v[yy] = yy < nrows ? sp[(int64_t)yy * EWX32] : 0u;
#pragma unroll
            for (int yy = 0; yy < ROWS + 4; yy++) { s_bits[k * RS + yy * W + w] = v[yy]; acc |= v[yy]; }
        }
        __syncthreads();
    }
    if (mode & 2) {
        if (__any(acc != 0) ) s_bits[0] = 1;
        __syncthreads();
        acc = s_bits[0];
    }
    float kf = acc ? 1.0f : 0.0f;
    float* o = field + ((int64_t)Z * Ny + Y0) * pitch + 4 * tid + coff;
    float4 k4 = make_float4(kf, kf, kf, kf);
    if (mode & 8) {
        int lane = tid & 63, wave = tid >> 6;
        for (int Y = Y0; Y < Y1; Y++, o += pitch) {
            *(float4*)o = k4;
            if (lane < 4) sig[(((int64_t)Z * Ny + Y) * 5 + wave) * 4 + lane] = (unsigned long long)lane + acc;
        }
    } else
    for (int Y = Y0; Y < Y1; Y++, o += pitch) *(float4*)o = k4;
    if (mode & 4) {   // pad columns: lanes < rows of wave 0 / wave 3 store one float each
        int lane = tid & 63, wave = tid >> 6;
        float* q = field + ((int64_t)Z * Ny + Y0 + lane) * pitch;
        if (lane < Y1 - Y0) {
            if (wave == 0) q[coff - 1] = kf;
            if (wave == 3) q[coff + 1024] = kf;
        }
    }
}
extern "C" int exp_store(const void* ext, float* field, int Nz, int Ny, int64_t pitch, int EY, int EWX32, int mode, int rows, int coff, int lds, void* sig, void* stream)
{
    if (rows == 32) {
        dim3 grid(1, (Ny + 31) / 32, Nz);
        hipLaunchKernelGGL(k_store<32>, grid, dim3(256), lds, (hipStream_t)stream, (const u32*)ext, field, Ny, pitch, EY, EWX32, mode, coff, (unsigned long long*)sig);
    } else {
        dim3 grid(1, (Ny + 63) / 64, Nz);
        hipLaunchKernelGGL(k_store<64>, grid, dim3(256), 5 * 68 * 34 * 4, (hipStream_t)stream, (const u32*)ext, field, Ny, pitch, EY, EWX32, mode, coff, (unsigned long long*)sig);
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
