// Developer probe: does the VOP3 clamp bit saturate v_mad_u32_u16 on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t *o, const uint32_t *a, const uint32_t *b, uint32_t mul)
{
    int i = threadIdx.x;
    uint32_t r0, r1;
    asm volatile("v_mad_u32_u16 %0, %1, %2, %3 clamp" : "=v"(r0) : "v"(a[i]), "s"(mul), "v"(b[i]));
    asm volatile("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0] clamp" : "=v"(r1) : "v"(a[i]), "s"(mul), "v"(b[i]));
    o[2 * i] = r0;
    o[2 * i + 1] = r1;
}
int main()
{
    uint32_t ha[64], hb[64], ho[128], *da, *db, *dout;
    for (int i = 0; i < 64; i++) { ha[i] = (uint32_t)(i * 1021u + 7u) | ((uint32_t)(65535 - i * 900) << 16); hb[i] = i < 32 ? 0xE0000000u : 0x00012345u * (i - 31); }
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 512);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db, 32768u);
    hipMemcpy(ho, dout, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; i++)
    {
        uint64_t e0 = (uint64_t)(ha[i] & 0xffff) * 32768u + hb[i], e1 = (uint64_t)(ha[i] >> 16) * 32768u + hb[i];
        uint32_t s0 = e0 > 0xffffffffull ? 0xffffffffu : (uint32_t)e0, s1 = e1 > 0xffffffffull ? 0xffffffffu : (uint32_t)e1;
        if (ho[2 * i] != s0 || ho[2 * i + 1] != s1) { if (bad < 4) printf("lane %d: got %08x %08x want %08x %08x (wrapped %08x %08x)\n", i, ho[2*i], ho[2*i+1], s0, s1, (uint32_t)e0, (uint32_t)e1); bad++; }
    }
    printf("v_mad_u32_u16 clamp: %s (%d mismatches)\n", bad ? "does NOT saturate" : "saturates", bad);
    return 0;
}
