// experiment: pure-store kernels with the tile kernel's geometry (no loads) -- which block shape / order / occupancy
// reaches the memset rate?
#include <hip/hip_runtime.h>
#include <stdint.h>
extern __shared__ unsigned s_dyn[];
typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef STFLAGS
#define STFLAGS ""
#endif
__device__ static inline void st4(float* p, float4 v) { v4f w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off " STFLAGS :: "v"(p), "v"(w) : "memory"); }
// block = zgn slices x rows rows x full pitch; flat row-major line order, 4 waves interleaved
__global__ __launch_bounds__(256) void k_store2(float* __restrict__ field, int Nz, int Ny, int64_t pitch, int zgn, int rows,
                                                int ntr, float val, int order, int touch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned lin = blockIdx.x;
    int tr, zg;
    if (order == 0) { tr = lin % ntr; zg = lin / ntr; }
    else {   // XCD-chunked: XCD x (= lin % 8) sweeps contiguous runs of 32 consecutive (zg, tr) blocks
        unsigned xcd = lin & 7u, idx = lin >> 3;
        unsigned run = idx / 32u, within = idx % 32u;
        unsigned b = (run * 8u + xcd) * 32u + within;
        tr = b % ntr; zg = b / ntr;
    }
    const int nt = (int)(pitch / 32);
    const int Y0 = tr * rows, Z0 = zg * zgn;
    if (Z0 >= Nz) return;
    const int nrows = Y0 + rows <= Ny ? rows : Ny - Y0;
    const int nlines = nrows * nt;
    const float4 k4 = make_float4(val, val, val, val);
    for (int z = 0; z < zgn && Z0 + z < Nz; z++) {
        float* zbase = field + ((int64_t)(Z0 + z) * Ny + Y0) * pitch + 4 * (lane & 7);
        if (touch == 0 || touch == 1) {
            for (int line = 8 * wave + (lane >> 3); line < nlines; line += 32)
                st4(zbase + (int64_t)line * 32, k4);
        } else if (touch == 2) {          // each wave a contiguous quarter
            int per = ((nlines + 31) / 32) * 8;
            for (int line = per * wave + (lane >> 3); line < per * (wave + 1) && line < nlines; line += 8)
                st4(zbase + (int64_t)line * 32, k4);
        } else if (touch == 3) {          // row-wise: wave w takes 1 KiB groups w, w+4.. of a row, row after row
            for (int g = wave; g * 8 < nt; g += 4) {
                int jl = 8 * g + (lane >> 3);
                if (jl < nt) for (int r = 0; r < nrows; r++) st4(zbase + (int64_t)r * pitch + 32 * jl, k4);
            }
        } else if (touch == 4) {          // rotated interleave
            int i = 0;
            for (int base = 0; base < nlines; base += 32, i++) {
                int line = base + 8 * ((wave + i) & 3) + (lane >> 3);
                if (line < nlines) st4(zbase + (int64_t)line * 32, k4);
            }
        } else if (touch == 5) {          // all 4 waves: 2 lines each per instr? no: 16 B x 256 threads = 4 KiB per block-instr, thread-linear
            for (int q = threadIdx.x; q < nlines * 8; q += 256)
                st4(field + ((int64_t)(Z0 + z) * Ny + Y0) * pitch + (int64_t)q * 4, k4);
        }
    }
}
extern "C" int EXPNAME(float* field, int Nz, int Ny, int64_t pitch, int zgn, int rows, float val, int order, int lds, int mode, void* stream)
{
    int ntr = (Ny + rows - 1) / rows, nzg = (Nz + zgn - 1) / zgn;
    int64_t blocks = (int64_t)ntr * nzg;
    if (order) blocks = (blocks + 255) / 256 * 256;
    if (lds > 65536) hipFuncSetAttribute((const void*)k_store2, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k_store2, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, field, Nz, Ny, pitch, zgn, rows, ntr, val, order, mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
