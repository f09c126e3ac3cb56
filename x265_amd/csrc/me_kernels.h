// __global__ kernels of the motion-estimation engine (gfx950).
#pragma once

#include "me_search.h"

namespace xa {

constexpr int MAX_PLANES = 256;
constexpr int FULL_MAX_CAND = 256;       // candidates per axis the exhaustive kernel handles (larger ranges: general kernel)

struct BatchParams
{
    const PlaneDesc *planes;             // device array of registered planes
    const uint16_t  *cost[52];           // per-QP MV cost tables (centre pointers); NULL if not built
    int32_t          enc_plane;
    int32_t          ref_planes[X265AMD_MAX_REFS];
    int32_t          n_refs;
};

// Does the dedicated exhaustive kernel handle this job?  (8-bit pixels, SAD metric, bounded range)
__device__ __host__ inline bool full_kernel_eligible(const x265amd_me_job &j, int bpp)
{
    int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    return j.method == X265AMD_FULL && j.icost == X265AMD_SAD && bpp == 1 &&
           nx > 0 && ny > 0 && nx <= FULL_MAX_CAND && ny <= FULL_MAX_CAND;
}

// ------------------------------------------------------------------------------------------
// General search: one wavefront per job, any method / sub-pel level / pixel type.
template <typename PIX>
__global__ void __launch_bounds__(64) k_search(BatchParams P, const x265amd_me_job *jobs, int n,
                                               const FullBest *fb, int fb_stride, x265amd_me_result *out,
                                               unsigned long long *evals)
{
    __shared__ uint32_t s_fenc[1024 * PixTraits<PIX>::DW];
    __shared__ int s_roff[1024];
    int id = blockIdx.x;
    if (id >= n)
        return;
    const x265amd_me_job &j = jobs[id];
    const PlaneDesc &enc = P.planes[P.enc_plane];
    const PlaneDesc &ref = P.planes[P.ref_planes[j.ref_idx]];
    Wave<PIX> W;
    W.fenc = s_fenc;
    W.roff = s_roff;
    W.cost = P.cost[j.qp];
    W.bind(enc, ref, j.pu_x, j.pu_y, j.pu_x, j.pu_y, j.w, j.h, true);
    Search<PIX> S(W, j);
    int ox, oy;
    const FullBest *f = (fb && full_kernel_eligible(j, (int)sizeof(PIX))) ? fb + (size_t)id * fb_stride : nullptr;
    int cost = S.run(f, fb_stride, ox, oy);
    if (W.lane == 0)
    {
        x265amd_me_result r;
        r.mvx = (int16_t)ox;
        r.mvy = (int16_t)oy;
        r.cost = cost;
        out[id] = r;
        atomicAdd(evals, (unsigned long long)W.evals);
    }
}

// ------------------------------------------------------------------------------------------
// Batched block costs: one wavefront per (enc block, ref block) pair.
template <typename PIX>
__global__ void __launch_bounds__(64) k_cost(const PlaneDesc *planes, int kind, int w, int h,
                                             const x265amd_block_pair *pairs, int n, int32_t *out)
{
    __shared__ uint32_t s_fenc[1024 * PixTraits<PIX>::DW];
    __shared__ int s_roff[1024];
    int id = blockIdx.x;
    if (id >= n)
        return;
    const x265amd_block_pair &p = pairs[id];
    Wave<PIX> W;
    W.fenc = s_fenc;
    W.roff = s_roff;
    W.cost = nullptr;
    W.bind(planes[p.enc_plane], planes[p.ref_plane], p.enc_x, p.enc_y, p.ref_x, p.ref_y, w, h, false);
    int c = W.template cost_kind<PRED_INT>(kind, W.off_of(0, 0), 0, 0);
    if (W.lane == 0)
        out[id] = c;
}

} // namespace xa
