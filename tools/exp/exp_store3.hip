// experiment: replicas of a linear fill kernel, to find what separates 0.74 ms (torch fill) from 0.80 ms (block-structured)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ static inline void st4(float* p, v4f w) { asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(w) : "memory"); }
// each block: T threads, U float4 per thread, thread-linear (block writes T*U*16 contiguous bytes)
template <int U>
__global__ void k_fill(float* __restrict__ f, int64_t n4, float val)
{
    v4f k = {val, val, val, val};
    int64_t base = (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x;
#pragma unroll
    for (int i = 0; i < U; i++) { int64_t q = base + (int64_t)i * blockDim.x; if (q < n4) st4(f + q * 4, k); }
}
// each lane: U consecutive float4 (64 B per lane for U = 4)
template <int U>
__global__ void k_fill_lane(float* __restrict__ f, int64_t n4, float val)
{
    v4f k = {val, val, val, val};
    int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * U;
#pragma unroll
    for (int i = 0; i < U; i++) { int64_t q = base + i; if (q < n4) st4(f + q * 4, k); }
}
// plain C++ stores (compiler chooses)
template <int U>
__global__ void k_fill_c(float4* __restrict__ f, int64_t n4, float val)
{
    float4 k = make_float4(val, val, val, val);
    int64_t base = (int64_t)blockIdx.x * blockDim.x * U + threadIdx.x;
#pragma unroll
    for (int i = 0; i < U; i++) { int64_t q = base + (int64_t)i * blockDim.x; if (q < n4) f[q] = k; }
}
// row-structured sweep: one block per (row, group of WPB segments); each wave one 1-KiB segment of a row of `pitch` floats
__global__ void k_rowsweep(float* __restrict__ f, int64_t nrows, int64_t pitch, int nseg, float val)
{
    v4f k = {val, val, val, val};
    const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int gpr = (nseg + wpb - 1) / wpb;              // block groups per row
    const int64_t row = blockIdx.x / gpr;
    const int seg = (int)(blockIdx.x % gpr) * wpb + wave;
    const int64_t col = (int64_t)seg * 256 + lane * 4;
    if (row < nrows && col < pitch) st4(f + row * pitch + col, k);
}
// persistent grid-stride page sweep: block b writes pages b, b+G, b+2G, ...; optional dependent byte load per line
__global__ __launch_bounds__(256) void k_persist(float* __restrict__ f, int64_t npages, const unsigned char* __restrict__ cls, int mode)
{
    v4f k = {0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x;
    int64_t pg = blockIdx.x;
    if (mode == 0) {
        for (; pg < npages; pg += gridDim.x) st4(f + pg * 1024 + t * 4, k);
    } else if (mode == 1) {   // dependent load, no prefetch
        for (; pg < npages; pg += gridDim.x) {
            unsigned char c = cls[pg * 32 + (t >> 3)];
            if (c < 2) st4(f + pg * 1024 + t * 4, k);
        }
    } else {                  // prefetch `mode - 1` pages ahead (2 or 3)
        unsigned char c0 = pg < npages ? cls[pg * 32 + (t >> 3)] : 3;
        unsigned char c1 = pg + gridDim.x < npages ? cls[(pg + gridDim.x) * 32 + (t >> 3)] : 3;
        for (; pg < npages; pg += gridDim.x) {
            int64_t nx = pg + 2 * (int64_t)gridDim.x;
            unsigned char c2 = nx < npages ? cls[nx * 32 + (t >> 3)] : 3;
            if (c0 < 2) st4(f + pg * 1024 + t * 4, k);
            c0 = c1; c1 = c2;
        }
    }
}
extern "C" int exp_persist(float* f, int64_t nfloats, const void* cls, int blocks, int mode, void* stream)
{
    hipLaunchKernelGGL(k_persist, dim3(blocks), dim3(256), 0, (hipStream_t)stream, f, nfloats / 1024, (const unsigned char*)cls, mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
// one page per block with a dependent class load (what field_sweep_kernel does)
__global__ __launch_bounds__(256) void k_page_dep(float* __restrict__ f, int64_t npages, const unsigned char* __restrict__ cls)
{
    v4f k = {0.f, 0.f, 0.f, 0.f};
    const int t = threadIdx.x;
    int64_t pg = blockIdx.x;
    unsigned char c = cls[pg * 32 + (t >> 3)];
    if (c < 2) st4(f + pg * 1024 + t * 4, k);
}
// one page per block, class bytes of the wave's 8 lines by ONE scalar load (s_load_dwordx2)
__global__ __launch_bounds__(256) void k_page_sdep(float* __restrict__ f, int64_t npages, const unsigned long long* __restrict__ cls8)
{
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int64_t pg = blockIdx.x;
    const unsigned long long c8 = cls8[pg * 4 + wave];        // uniform address -> scalar load
    const unsigned c = (unsigned)(c8 >> (8 * ((t >> 3) & 7))) & 0xffu;
    const float kf = c ? 1.0f : 0.0f;
    v4f k = {kf, kf, kf, kf};
    if (c < 2) st4(f + pg * 1024 + t * 4, k);
}
extern "C" int exp_page_sdep(float* f, int64_t nfloats, const void* cls, void* stream)
{
    hipLaunchKernelGGL(k_page_sdep, dim3((unsigned)(nfloats / 1024)), dim3(256), 0, (hipStream_t)stream, f, nfloats / 1024, (const unsigned long long*)cls);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int exp_page_dep(float* f, int64_t nfloats, const void* cls, void* stream)
{
    hipLaunchKernelGGL(k_page_dep, dim3((unsigned)(nfloats / 1024)), dim3(256), 0, (hipStream_t)stream, f, nfloats / 1024, (const unsigned char*)cls);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int exp_rowsweep(float* f, int64_t nrows, int64_t pitch, int wpb, void* stream)
{
    int nseg = (int)((pitch + 255) / 256);
    int gpr = (nseg + wpb - 1) / wpb;
    hipLaunchKernelGGL(k_rowsweep, dim3((unsigned)(nrows * gpr)), dim3(64 * wpb), 0, (hipStream_t)stream, f, nrows, pitch, nseg, 0.0f);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
extern "C" int exp_fill(float* f, int64_t nfloats, int T, int U, int kind, void* stream)
{
    int64_t n4 = nfloats / 4;
    int64_t blocks = (n4 + (int64_t)T * U - 1) / ((int64_t)T * U);
    hipStream_t s = (hipStream_t)stream;
#define L(K, UU) hipLaunchKernelGGL(K<UU>, dim3((unsigned)blocks), dim3(T), 0, s, f, n4, 0.0f)
#define LC(UU) hipLaunchKernelGGL(k_fill_c<UU>, dim3((unsigned)blocks), dim3(T), 0, s, (float4*)f, n4, 0.0f)
    if (kind == 0) { if (U == 1) L(k_fill, 1); else if (U == 2) L(k_fill, 2); else if (U == 4) L(k_fill, 4); else if (U == 8) L(k_fill, 8); else L(k_fill, 16); }
    else if (kind == 1) { if (U == 1) L(k_fill_lane, 1); else if (U == 2) L(k_fill_lane, 2); else if (U == 4) L(k_fill_lane, 4); else L(k_fill_lane, 8); }
    else { if (U == 1) LC(1); else if (U == 2) LC(2); else if (U == 4) LC(4); else if (U == 8) LC(8); else LC(16); }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
