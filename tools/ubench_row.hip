// Micro-benchmark of the exhaustive kernel's inner loop: per window row, W4 LDS dword-pair loads,
// W4 encoder dwords (LDS broadcast or none) and 2*W4 (or 4*W4) v_qsad_pk_u16_u8 in CH dependent chains.
// Prints the achieved fraction of the packed-SAD issue peak for several occupancies.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
typedef uint32_t u32; typedef unsigned long long u64;
constexpr int W4 = 16, ROWS = 64, TASKS = 64;

__device__ __forceinline__ u64 lds_pair(const u32 *p) { u64 v; __builtin_memcpy(&v, p, 8); return v; }

template <int CH, int FENC>      // CH candidate rows per lane task (2 or 4); FENC: 0 none, 1 LDS broadcast
__global__ void __launch_bounds__(256) k_row(u32 *out, int pitch, int lds_dwords)
{
    extern __shared__ u32 lds[];
    for (int i = threadIdx.x; i < lds_dwords; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    const u32 *fenc = lds;                       // ROWS * W4 dwords
    const u32 *win = lds + ROWS * W4;
    u64 acc[CH];
    for (int i = 0; i < CH; i++) acc[i] = 0;
    u32 sum = 0;
    for (int t = 0; t < TASKS; t++)
    {
        const u32 *row = win + ((threadIdx.x + t) & 31) + ((threadIdx.x >> 5) & 1) * pitch * 2;
        for (int r = 0; r < ROWS; r++)
        {
            u64 d[W4]; u32 f[W4];
#pragma unroll
            for (int k = 0; k < W4; k++) d[k] = lds_pair(row + r * pitch + k);
#pragma unroll
            for (int k = 0; k < W4; k++) f[k] = FENC ? fenc[r * W4 + k] : (u32)(r + k);
#pragma unroll
            for (int k = 0; k < W4; k++)
#pragma unroll
                for (int c = 0; c < CH; c++)
                    acc[c] = __builtin_amdgcn_qsad_pk_u16_u8(d[k], f[k] + (FENC ? 0 : c), acc[c]);
        }
        for (int c = 0; c < CH; c++) { sum += (u32)acc[c] + (u32)(acc[c] >> 32); acc[c] = 0; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int CH, int FENC>
int run(const char *name, int wg_per_cu, int cus, double clk, u32 *d_out)
{
    const int pitch = 49;
    const int lds_bytes = 160 * 1024 / wg_per_cu - 256;
    CHECK(hipFuncSetAttribute((const void*)k_row<CH, FENC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = cus * wg_per_cu * 2;
    hipLaunchKernelGGL((k_row<CH, FENC>), dim3(blocks), dim3(256), lds_bytes, 0, d_out, pitch, lds_bytes / 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_row<CH, FENC>), dim3(blocks), dim3(256), lds_bytes, 0, d_out, pitch, lds_bytes / 4);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double qsads = (double)blocks * 256 * TASKS * ROWS * W4 * CH;          // lane-level instructions
    double per = qsads / (ms * 1e-3) / clk / cus;
    printf("%-28s wg/CU %d  %7.3f ms  %5.2f qsad lane-ops/clk/CU  (%.0f%% of 16)\n", name, wg_per_cu, ms, per, per / 16 * 100);
    return 0;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    u32 *d_out; CHECK(hipMalloc(&d_out, (size_t)cus * 16 * 256 * 4 * 2));
    for (int wg = 1; wg <= 6; wg++)
    {
        if (run<2, 1>("2 chains, LDS enc", wg, cus, clk, d_out)) return 1;
        if (run<2, 0>("2 chains, no enc loads", wg, cus, clk, d_out)) return 1;
        if (run<4, 1>("4 chains, LDS enc", wg, cus, clk, d_out)) return 1;
    }
    return 0;
}
