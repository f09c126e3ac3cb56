// Developer aid: checks wave_sum_lanes<N> (x265_amd/csrc/me_device.h) against host sums on the GPU.
//   hipcc --offload-arch=gfx950 -O2 -I x265_amd/csrc -I include -o tools/reduce_probe tools/reduce_probe.hip && tools/reduce_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "me_device.h"

template <int N>
__global__ void k_probe(const int *in, int *out)
{
    int v[N];
    for (int k = 0; k < N; k++)
        v[k] = in[k * 64 + threadIdx.x];
    out[threadIdx.x] = xa::wave_sum_lanes<N>(v);
}

template <int N>
static int run(const int *d_in, int *d_out, const int *h_in)
{
    hipLaunchKernelGGL(k_probe<N>, dim3(1), dim3(64), 0, 0, d_in, d_out);
    int h_out[64];
    if (hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    constexpr int M = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : 8;
    int bad = 0;
    for (int l = 0; l < 64; l++)
    {
        const int k = l & (M - 1);
        int want = 0;
        if (k < N)
            for (int i = 0; i < 64; i++)
                want += h_in[k * 64 + i];
        if (h_out[l] != want)
            bad++;
    }
    printf("wave_sum_lanes<%d>: %s\n", N, bad ? "MISMATCH" : "ok");
    return bad != 0;
}

int main()
{
    int h_in[8 * 64];
    srand(1);
    for (int i = 0; i < 8 * 64; i++)
        h_in[i] = rand() % 100000;
    int *d_in, *d_out;
    if (hipMalloc(&d_in, sizeof(h_in)) != hipSuccess || hipMalloc(&d_out, 64 * sizeof(int)) != hipSuccess)
        return 2;
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    int rc = run<1>(d_in, d_out, h_in) | run<2>(d_in, d_out, h_in) | run<3>(d_in, d_out, h_in) | run<4>(d_in, d_out, h_in) |
             run<6>(d_in, d_out, h_in) | run<8>(d_in, d_out, h_in);
    hipFree(d_in); hipFree(d_out);
    return rc;
}
