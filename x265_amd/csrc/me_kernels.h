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
                                               const FullBest *fb, x265amd_me_result *out,
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
    const FullBest *f = (fb && full_kernel_eligible(j, (int)sizeof(PIX))) ? fb + id : nullptr;
    int cost = S.run(f, ox, oy);
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

// ------------------------------------------------------------------------------------------
// Exhaustive (FULL_SEARCH) integer stage for 8-bit planes: one 256-thread workgroup per job.
//
// The reference window of the job is staged into LDS once (coalesced dword reads).  A lane
// task is a group of 4 horizontally consecutive candidates on 2 vertically consecutive
// candidate rows.  For each window row the lane reads W4+1 consecutive dwords and issues
// v_qsad_pk_u16_u8 -- 4 byte-shifted 4-pixel SADs per instruction, no byte realignment --
// against the matching dword of the encoder block (staged in LDS, read as a broadcast).
// Each window dword feeds both candidate rows (vertical reuse).  16-bit packed accumulators
// are flushed to 32 bits before they can overflow (every <=256 pixels).
// Winner = minimum of (cost, raster index): the candidate the reference's strict-`<` raster
// scan would keep (encoder/motion.cpp:738-771).
template <int W4>
__device__ void full_body(const PlaneDesc &enc, const PlaneDesc &ref, const uint16_t *cost,
                          const x265amd_me_job &j, uint32_t *lds, int lds_dwords, FullBest *out)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int w = W4 * 4, h = j.h;
    const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    const int XG = (nx + 3) >> 2;
    const int XGe = XG + (XG & 1);                 // tasks per candidate-row pair (even: keeps LDS banks disjoint)
    const int ND = XGe + W4;                       // dwords per staged window row
    int P = ND + ((((XGe >> 1) - ND) % 16) + 16) % 16;   // pitch: 2*P == XGe (mod 32)

    uint32_t *s_fenc = lds;                        // [h * W4]
    int *s_cx = (int*)(lds + h * W4);              // [nx]  mvcost of candidate column
    int *s_cy = s_cx + nx;                         // [ny]
    uint32_t *s_win = (uint32_t*)(s_cy + ny);
    const int win_dwords = lds_dwords - (h * W4 + nx + ny);
    const int rows_fit = win_dwords / P;           // window rows that fit
    int TH = rows_fit - h;                         // candidate rows per tile (need TH + h - 1 (+1 for the pair) rows)
    TH &= ~1;

    // encoder block and MV cost slices
    const uint8_t *e = (const uint8_t*)enc.origin + (int64_t)j.pu_y * enc.stride + j.pu_x;
    for (int i = tid; i < h * W4; i += 256)
        s_fenc[i] = ld_u32(e + (int64_t)(i / W4) * enc.stride + 4 * (i % W4));
    for (int i = tid; i < nx; i += 256)
        s_cx[i] = cost[(int16_t)((int16_t)((j.mvmin_x + i) << 2) - j.qmvp_x)];
    for (int i = tid; i < ny; i += 256)
        s_cy[i] = cost[(int16_t)((int16_t)((j.mvmin_y + i) << 2) - j.qmvp_y)];

    const int64_t blockOffset = (int64_t)j.pu_y * ref.stride + j.pu_x;
    const uint8_t *r0 = (const uint8_t*)ref.origin;
    const int64_t lo = ref.lo, hi = ref.hi - 4;

    unsigned long long best = ~0ull;               // (cost << 32) | raster index
    const int rows_per_flush = 256 / w;

    if (TH < 2)
        TH = 0;                                    // window row does not fit: leave best = none (host avoids this)

    for (int ty0 = 0; ty0 < ny && TH; ty0 += TH)
    {
        const int th = (ny - ty0 < TH) ? ny - ty0 : TH;          // candidate rows in this tile
        const int rows = th + h - 1 + (th & 1);                  // staged rows (pairs may read one extra)
        __syncthreads();
        for (int r = wv; r < rows; r += 4)
        {
            int64_t o = blockOffset + (int64_t)(j.mvmin_y + ty0 + r) * ref.stride + j.mvmin_x;
            for (int c = lane; c < ND; c += 64)
            {
                int64_t a = o + 4 * c;
                a = a < lo ? lo : (a > hi ? hi : a);
                s_win[r * P + c] = ld_u32(r0 + a);
            }
        }
        __syncthreads();

        const int npairs = (th + 1) >> 1;
        const int ntasks = npairs * XGe;
        for (int task = tid; task < ntasks; task += 256)
        {
            const int yp = task / XGe, xg = task - yp * XGe;
            if (xg >= XG)
                continue;
            const uint32_t *row = s_win + (2 * yp) * P + xg;
            unsigned long long pa = 0, pb = 0;                    // packed 4 x u16 partial sums, rows y0 / y0+1
            uint32_t sa[4] = { 0, 0, 0, 0 }, sb[4] = { 0, 0, 0, 0 };
            uint32_t fprev[W4], fcur[W4], d[W4 + 1];
            int since = 0;
#pragma unroll
            for (int k = 0; k < W4; k++)
                fprev[k] = 0;
            for (int r = 0; r <= h; r++)
            {
#pragma unroll
                for (int k = 0; k <= W4; k++)
                    d[k] = row[k];
                row += P;
                if (r < h)
                {
#pragma unroll
                    for (int k = 0; k < W4; k++)
                        fcur[k] = s_fenc[r * W4 + k];
#pragma unroll
                    for (int k = 0; k < W4; k++)
                        pa = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)d[k + 1] << 32) | d[k], fcur[k], pa);
                }
                if (r > 0)
                {
#pragma unroll
                    for (int k = 0; k < W4; k++)
                        pb = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)d[k + 1] << 32) | d[k], fprev[k], pb);
                }
#pragma unroll
                for (int k = 0; k < W4; k++)
                    fprev[k] = fcur[k];
                if (++since >= rows_per_flush || r == h)
                {
                    since = 0;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        sa[i] += (uint32_t)(pa >> (16 * i)) & 0xffff;
                        sb[i] += (uint32_t)(pb >> (16 * i)) & 0xffff;
                    }
                    pa = pb = 0;
                }
            }
            const int iy = ty0 + 2 * yp;
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
            {
                const int y = iy + rr;
                if (y >= ty0 + th)
                    break;
                const int cy = s_cy[y];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int x = 4 * xg + i;
                    if (x < nx)
                    {
                        uint32_t c = (rr ? sb[i] : sa[i]) + (uint32_t)((s_cx[x] + cy) & 0xffff);
                        unsigned long long key = ((unsigned long long)c << 32) | (uint32_t)(y * nx + x);
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }

    // workgroup min-reduce
    for (int m = 32; m >= 1; m >>= 1)
    {
        unsigned long long o = __shfl_xor(best, m);
        best = o < best ? o : best;
    }
    __syncthreads();
    unsigned long long *s_red = (unsigned long long*)lds;
    if (lane == 0)
        s_red[wv] = best;
    __syncthreads();
    if (tid == 0)
    {
        for (int i = 1; i < 4; i++)
            best = s_red[i] < best ? s_red[i] : best;
        FullBest fbv;
        if (best == ~0ull)
        {
            fbv.cost = 0x7fffffff;
            fbv.mvx = fbv.mvy = 0;
        }
        else
        {
            int idx = (int)(best & 0xffffffffu);
            fbv.cost = (int)(best >> 32);
            fbv.mvx = (int16_t)(j.mvmin_x + idx % nx);
            fbv.mvy = (int16_t)(j.mvmin_y + idx / nx);
        }
        *out = fbv;
    }
}

__global__ void __launch_bounds__(256) k_full_u8(BatchParams P, const x265amd_me_job *jobs, int n,
                                                 FullBest *fb, int lds_dwords)
{
    extern __shared__ uint32_t lds[];
    int id = blockIdx.x;
    if (id >= n)
        return;
    const x265amd_me_job &j = jobs[id];
    const PlaneDesc &enc = P.planes[P.enc_plane];
    if (!full_kernel_eligible(j, enc.bpp))
        return;
    const PlaneDesc &ref = P.planes[P.ref_planes[j.ref_idx]];
    const uint16_t *cost = P.cost[j.qp];
    switch (j.w >> 2)
    {
    case 1:  full_body<1>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 2:  full_body<2>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 3:  full_body<3>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 4:  full_body<4>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 6:  full_body<6>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 8:  full_body<8>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 12: full_body<12>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    case 16: full_body<16>(enc, ref, cost, j, lds, lds_dwords, fb + id); break;
    default: break;
    }
}

} // namespace xa
