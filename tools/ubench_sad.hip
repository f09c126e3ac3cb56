// Micro-benchmark: per-instruction throughput of the packed-SAD VALU ops and LDS read forms
// that the block-matching kernels are built from.  Prints lane-ops per clock per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef uint32_t u32;
typedef uint64_t u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) k_valu(u32 *out, u32 seed)
{
    u32 a[8], b = seed * 7 + threadIdx.x;
    u64 q[8];
    u32x4 m[4];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; q[i] = a[i]; }
    for (int i = 0; i < 4; i++) m[i] = (u32x4){a[i], a[i + 1], a[i + 2], a[i + 3]};
    u64 r64 = ((u64)b << 32) | seed;
    for (int it = 0; it < ITERS; it++)
    {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
        {
            int k = u & 7;
            if (OP == 0) a[k] = __builtin_amdgcn_sad_u8(a[k], b, a[k]);
            if (OP == 1) a[k] = __builtin_amdgcn_sad_u16(a[k], b, a[k]);
            if (OP == 2) q[k] = __builtin_amdgcn_qsad_pk_u16_u8(r64, b, q[k]);
            if (OP == 3) m[k & 3] = __builtin_amdgcn_mqsad_u32_u8(r64, b, m[k & 3]);
            if (OP == 4) a[k] = __builtin_amdgcn_alignbyte(a[k], b, 1);
            if (OP == 5) a[k] = __builtin_amdgcn_perm(a[k], b, 0x02010003);
            if (OP == 6) a[k] = a[k] + b;
            if (OP == 7) a[k] = __builtin_amdgcn_msad_u8(a[k], b, a[k]);
            if (OP == 8) q[k] = __builtin_amdgcn_mqsad_pk_u16_u8(r64, b, q[k]);
            if (OP == 9) a[k] = __builtin_amdgcn_sad_hi_u8(a[k], b, a[k]);
            if (OP == 10) q[0] = __builtin_amdgcn_qsad_pk_u16_u8(r64, b, q[0]);          // one dependent chain
            if (OP == 11) q[k & 1] = __builtin_amdgcn_qsad_pk_u16_u8(r64, b, q[k & 1]);  // two chains
            if (OP == 12) a[0] = __builtin_amdgcn_sad_u8(a[0], b, a[0]);                  // dependent v_sad_u8 chain
        }
    }
    u32 s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + (u32)q[i] + (u32)(q[i] >> 32);
    for (int i = 0; i < 4; i++) s += m[i].x + m[i].y + m[i].z + m[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS read throughput: WIDTH bytes per lane per read, consecutive lanes -> consecutive WIDTH-byte slots
template <int WIDTH>
__global__ void __launch_bounds__(256) k_lds(u32 *out, u32 seed)
{
    __shared__ u32 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i * seed;
    __syncthreads();
    u32 acc = 0;
    u32 addr = (u32)(size_t)lds + (threadIdx.x & 63) * WIDTH;
    for (int it = 0; it < ITERS; it++)
    {
        u32 a0, a1; u32x2 b0, b1; u32x4 c0, c1;
#pragma unroll
        for (int u = 0; u < UNROLL; u += 2)
        {
            if (WIDTH == 4)
            {
                asm volatile("ds_read_b32 %0, %2 offset:0\n ds_read_b32 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(a0), "=v"(a1) : "v"(addr));
                acc += a0 ^ a1;
            }
            if (WIDTH == 8)
            {
                asm volatile("ds_read_b64 %0, %2 offset:0\n ds_read_b64 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(b0), "=v"(b1) : "v"(addr));
                acc += b0.x ^ b1.y;
            }
            if (WIDTH == 16)
            {
                asm volatile("ds_read_b128 %0, %2 offset:0\n ds_read_b128 %1, %2 offset:1024\n s_waitcnt lgkmcnt(0)" : "=v"(c0), "=v"(c1) : "v"(addr));
                acc += c0.x ^ c1.w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// unaligned 4-byte global loads: checks that byte-offset dword reads work and what they cost
__global__ void k_unaligned(const uint8_t *src, u32 *out, int off)
{
    const uint8_t *p = src + threadIdx.x * 4 + off;
    u32 v;
    __builtin_memcpy(&v, p, 4);
    out[threadIdx.x] = v;
}

template <typename F>
static float time_kernel(F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; i++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;
    printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    const int waves_per_cu = 16;
    int blocks = cus * waves_per_cu / 4;
    u32 *out;
    CHECK(hipMalloc(&out, blocks * 256 * 4));
    const char *names[] = {"v_sad_u8", "v_sad_u16", "v_qsad_pk_u16_u8", "v_mqsad_u32_u8", "v_alignbyte_b32", "v_perm_b32", "v_add_u32", "v_msad_u8", "v_mqsad_pk_u16_u8", "v_sad_hi_u8", "qsad 1 chain", "qsad 2 chains", "sad_u8 1 chain"};
    float ms[13];
    ms[0] = time_kernel([&] { k_valu<0><<<blocks, 256>>>(out, 3); });
    ms[1] = time_kernel([&] { k_valu<1><<<blocks, 256>>>(out, 3); });
    ms[2] = time_kernel([&] { k_valu<2><<<blocks, 256>>>(out, 3); });
    ms[3] = time_kernel([&] { k_valu<3><<<blocks, 256>>>(out, 3); });
    ms[4] = time_kernel([&] { k_valu<4><<<blocks, 256>>>(out, 3); });
    ms[5] = time_kernel([&] { k_valu<5><<<blocks, 256>>>(out, 3); });
    ms[6] = time_kernel([&] { k_valu<6><<<blocks, 256>>>(out, 3); });
    ms[7] = time_kernel([&] { k_valu<7><<<blocks, 256>>>(out, 3); });
    ms[8] = time_kernel([&] { k_valu<8><<<blocks, 256>>>(out, 3); });
    ms[9] = time_kernel([&] { k_valu<9><<<blocks, 256>>>(out, 3); });
    ms[10] = time_kernel([&] { k_valu<10><<<blocks, 256>>>(out, 3); });
    ms[11] = time_kernel([&] { k_valu<11><<<blocks, 256>>>(out, 3); });
    ms[12] = time_kernel([&] { k_valu<12><<<blocks, 256>>>(out, 3); });
    for (int i = 0; i < 13; i++)
    {
        double laneops = (double)blocks * 256 * ITERS * UNROLL;
        double per_clk_cu = laneops / (ms[i] * 1e-3) / clk / cus;
        printf("%-20s %8.3f ms  %7.1f lane-ops/clk/CU (at nominal clock)\n", names[i], ms[i], per_clk_cu);
    }
    float l4 = time_kernel([&] { k_lds<4><<<blocks, 256>>>(out, 3); });
    float l8 = time_kernel([&] { k_lds<8><<<blocks, 256>>>(out, 3); });
    float l16 = time_kernel([&] { k_lds<16><<<blocks, 256>>>(out, 3); });
    double reads = (double)blocks * 256 * ITERS * UNROLL;
    printf("ds_read_b32  %8.3f ms  %6.1f B/clk/CU\n", l4, reads * 4 / (l4 * 1e-3) / clk / cus);
    printf("ds_read_b64  %8.3f ms  %6.1f B/clk/CU\n", l8, reads * 8 / (l8 * 1e-3) / clk / cus);
    printf("ds_read_b128 %8.3f ms  %6.1f B/clk/CU\n", l16, reads * 16 / (l16 * 1e-3) / clk / cus);

    uint8_t *src; CHECK(hipMalloc(&src, 4096));
    std::vector<uint8_t> h(4096);
    for (int i = 0; i < 4096; i++) h[i] = (uint8_t)(i * 7 + 1);
    CHECK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
    for (int off = 0; off < 4; off++)
    {
        k_unaligned<<<1, 64>>>(src, out, off);
        u32 r[64];
        CHECK(hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost));
        int ok = 1;
        for (int l = 0; l < 64; l++)
        {
            u32 e; memcpy(&e, &h[l * 4 + off], 4);
            if (e != r[l]) ok = 0;
        }
        printf("unaligned global dword load off %d: %s\n", off, ok ? "ok" : "WRONG");
    }
    return 0;
}
