// The lookahead's half-resolution path (gfx950): producing the four half-pel planes of a picture
// and searching a whole half-resolution frame with the reference's neighbour-predicted MV chain.
//
// Reference semantics being reproduced (paths relative to the reference's source/):
//   frame_init_lowres_core          common/pixel.cpp:614-637
//   Lowres::create / Lowres::init   common/lowres.cpp:30-58, 109-186
//   border extension                Lib/TLibCommon/TComPicYuv.cpp:259-286
//   Lookahead::estimateFrameCost / estimateCUCost, motion-search part
//                                   encoder/slicetype.cpp:48-63, 306-316, 346-369, 381-407
#pragma once

#include "me_kernels.h"

namespace xa {

// ------------------------------------------------------------------------------------------
// Half-resolution planes.  One thread produces 4 adjacent pixels of all four planes (F, H, V, C)
// at one position of the PADDED output domain: positions in the margins take the value of the
// nearest picture pixel, which is what extending the borders afterwards yields; the columns that pad
// the stride behind the right margin are zero.  The source is a
// border-replicated plane; its reads reach at most 2 pixels beyond twice the rounded-up
// half-resolution size (<= 16 pixels into its margins).
template <typename PIX>
__global__ void __launch_bounds__(256) k_lowres_init(const PIX *src, int64_t slo, int64_t shi, int sstride,
                                                     PIX *dst, int dstride, int rows, int width, int lines,
                                                     int mx, int my, int64_t lr_step)
{
    const int per_row = dstride >> 2;
    const int t = (int)(blockIdx.x * 256 + threadIdx.x);
    if (t >= per_row * rows)
        return;
    const int y = t / per_row, x0 = (t - y * per_row) * 4;
    int py = y - my;
    py = py < 0 ? 0 : py > lines - 1 ? lines - 1 : py;
    uint32_t o[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        int px = x0 + i - mx;
        px = px < 0 ? 0 : px > width - 1 ? width - 1 : px;
        uint32_t a[3][3];
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
            {
                int64_t off = (int64_t)(2 * py + r) * sstride + 2 * px + c;
                off = off < slo ? slo : off > shi - 1 ? shi - 1 : off;      // (never outside a plane with 16-pixel margins)
                a[r][c] = src[off];
            }
        // "slower than naive bilinear, but matches asm": two rounded means, then their rounded mean
        auto avg4 = [](uint32_t p, uint32_t q, uint32_t r, uint32_t s) { return (((p + q + 1) >> 1) + ((r + s + 1) >> 1) + 1) >> 1; };
        o[0][i] = avg4(a[0][0], a[1][0], a[0][1], a[1][1]);
        o[1][i] = avg4(a[0][1], a[1][1], a[0][2], a[1][2]);
        o[2][i] = avg4(a[1][0], a[2][0], a[1][1], a[2][1]);
        o[3][i] = avg4(a[1][1], a[2][1], a[1][2], a[2][2]);
        // Columns behind the right margin only pad the stride to a multiple of 32.  The reference leaves them
        // uninitialised (common/lowres.cpp:37-51), yet UMH's vertical cross singles, measured at x = omv.x + omv.y
        // (encoder/motion.cpp:176,271-273), can land in them when the vertical MV is large.  They hold zeros here,
        // as in a zero-initialised allocation, so that such searches are deterministic.
        if (x0 + i >= width + 2 * mx)
            o[0][i] = o[1][i] = o[2][i] = o[3][i] = 0;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        PIX *q = dst + k * lr_step + (int64_t)y * dstride + x0;
        if (sizeof(PIX) == 1)
            *(uint32_t*)q = o[k][0] | (o[k][1] << 8) | (o[k][2] << 16) | (o[k][3] << 24);
        else
            *(uint2*)q = make_uint2(o[k][0] | (o[k][1] << 16), o[k][2] | (o[k][3] << 16));
    }
}

// ------------------------------------------------------------------------------------------
// Border extension of freshly written picture rows [y0, y0 + n): their left / right margins, plus
// the top margin if y0 == 0 and the bottom margin if the rows end the picture - what FrameFilter does
// on the host before it publishes a reconstructed row (encoder/framefilter.cpp:201-262,
// Lib/TLibCommon/TComPicYuv.cpp:259-286).  Every margin pixel is the nearest picture pixel, read
// from the interior directly, so the order of the writes does not matter.  `org` = pixel (0,0).
template <typename PIX>
__global__ void __launch_bounds__(256) k_extend_rows(PIX *org, int stride, int width, int height, int mx, int my,
                                                     int y0, int n)
{
    const int first = y0 == 0 ? -my : y0, last = y0 + n == height ? height + my : y0 + n;     // padded rows [first, last)
    const int pw = width + 2 * mx;
    const int t = (int)(blockIdx.x * 256 + threadIdx.x);
    if (t >= (last - first) * pw)
        return;
    const int y = first + t / pw, x = t % pw - mx;
    if (x >= 0 && x < width && y >= 0 && y < height)
        return;                                     // interior: just written by the caller
    const int cy = y < 0 ? 0 : y > height - 1 ? height - 1 : y;
    const int cx = x < 0 ? 0 : x > width - 1 ? width - 1 : x;
    org[(int64_t)y * stride + x] = org[(int64_t)cy * stride + cx];
}

// ------------------------------------------------------------------------------------------
// Whole-frame lookahead search.  The reference walks the 8x8 CUs of the half-resolution frame in
// reverse raster order on one thread; a CU's predictor and candidates are the MVs of its right,
// lower, lower-left and lower-right neighbours.  Here one wavefront owns one CU ROW and walks it
// right to left; it may search CU (x, y) once the row below has finished CU x - 1, so the rows
// form a wavefront two CUs apart and the result is the reference's, bit for bit.
//
// Hand-over between rows: a CU's result is written with ONE 64-bit device-scope atomic store, and
// the result array starts out filled with a word no search can produce (MV component 0x7fff, the
// reference's own "not searched yet" mark, common/lowres.cpp:126-127): a reader polls the MV word
// of the neighbour it needs with device-scope atomic loads until it differs.  No fences, hence no
// cache write-backs or invalidations.  Row y waits only on row y + 1, i.e. workgroup b only on
// workgroup b - 1 of the same pair, which was dispatched earlier: the chain always makes progress.
// A time limit guards against a lost predecessor: the row that gives up stops waiting altogether,
// finishes with whatever it reads, and the call reports the failure.
struct LookPair { int32_t enc_plane, ref_plane; };

constexpr long long LOOK_WAIT_TICKS = 10ll * 100000000;   // 10 s of the 100 MHz real-time counter (a CU normally appears
                                                          // within tens of microseconds; an exhaustive search over a whole
                                                          // small frame can take tens of milliseconds per CU)
constexpr int LOOK_UNSET = 0x7fff7fff;

struct LookLeaf
{
    // MV words (mvy << 16 | mvx & 0xffff) of the lower-left, lower and lower-right neighbours
    // (`below` - 1, `below`, `below` + 1; bit k of `mask`: neighbour k exists), once they are all
    // there: lanes 0..2 poll one each.  Returned per lane; LOOK_UNSET in a lane that gave up at the
    // time limit, 0 for an absent neighbour.  Out of line: the kernel body keeps only uniform
    // control flow.
    static __device__ __noinline__ int wait_mv3(const x265amd_me_result *below, int mask_)
    {
        const int mask = uni(mask_), lane = lane_id();
        const bool mine = lane < 3 && ((mask >> lane) & 1);
        int w = mine ? LOOK_UNSET : 0;
        const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
        for (;;)
        {
            if (mine)
                w = __hip_atomic_load((const int*)(below + (lane - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__ballot(w == LOOK_UNSET) == 0 || (long long)__builtin_amdgcn_s_memrealtime() - t0 > LOOK_WAIT_TICKS)
                break;
            __builtin_amdgcn_s_sleep(2);
        }
        return w;
    }
    // one lane publishes a CU's result (inline: a call would wait for the store to complete before it returns)
    static __device__ __forceinline__ void publish(x265amd_me_result *out, int mvx, int mvy, int cost)
    {
        if (threadIdx.x == 0)
        {
            const unsigned long long v = ((unsigned long long)(uint32_t)cost << 32) | (uint32_t)((mvy << 16) | (mvx & 0xffff));
            __hip_atomic_store((unsigned long long*)out, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // one lane stores a word
    static __device__ __noinline__ void store_word(int *p, int v)
    {
        if (threadIdx.x == 0)
            *p = v;
    }
    static __device__ __noinline__ void finish(unsigned long long *evals, unsigned n_evals, int *error, int failed)
    {
        if (threadIdx.x == 0)
        {
            atomicAdd(evals, (unsigned long long)n_evals);
            if (failed)
                atomicOr(error, 1);
        }
    }
};

// median() of encoder/slicetype.cpp:48-57 with its 16-bit intermediates
__device__ __forceinline__ int look_median(int a, int b, int c)
{
    int t = XA_S16(a - b < 0 ? a - b : 0);
    a = XA_S16(a - t);
    b = XA_S16(b + t);
    b = XA_S16(b - (b - c < 0 ? b - c : 0));
    b = XA_S16(b + (a - b < 0 ? a - b : 0));
    return b;
}

template <class L>
__global__ void __launch_bounds__(64) k_lookahead(const PlaneDesc *planes, const uint16_t *cost, const LookPair *pairs,
                                                  int cu_w, int cu_h, int merange, int method, int subme, int qp, int flags,
                                                  x265amd_me_result *out, int *error, unsigned long long *evals)
{
    static_assert(L::NW == 1, "one wavefront per CU row");
    __shared__ __align__(16) L lds;
    const int pair = (int)blockIdx.x / cu_h;
    const int cuy = cu_h - 1 - ((int)blockIdx.x - pair * cu_h);         // workgroup 0 of a pair: the bottom row
    const PlaneDesc &enc = planes[pairs[pair].enc_plane];
    const PlaneDesc &ref = planes[pairs[pair].ref_plane];
    x265amd_me_result *res = out + (size_t)pair * cu_w * cu_h;
    const bool have_b = cuy < cu_h - 1;
    unsigned n_evals = 0;
    int failed = 0;
    int right = 0;                                                       // MV word of the CU searched last (right neighbour)
    Wave<L> W;
    W.sliced = false;
#ifdef XA_PROFILE
    long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif

    for (int cux = cu_w - 1; cux >= 0; cux--)
    {
        const bool have_r = cux < cu_w - 1;
        // the row below must have finished its CUs down to cux - 1 (it walks right to left, so
        // that one is the last of the three lower neighbours to appear)
        const x265amd_me_result *below = res + (size_t)(cuy + 1) * cu_w + cux;
        int w_b = 0, w_bl = 0, w_br = 0;
        W.stamp(0);
        // (the PU is cached while the neighbours may still be busy)
        W.bind(lds, enc, ref, 8 * cux, 8 * cuy, 8 * cux, 8 * cuy, 8, 8, cost, flags, true);
        W.stamp(1);
        if (have_b && !failed)              // (after one lost neighbour the row only runs to its end)
        {
            const int v = LookLeaf::wait_mv3(below, (cux > 0 ? 1 : 0) | 2 | (have_r ? 4 : 0));
            w_bl = __builtin_amdgcn_readlane(v, 0);
            w_b = __builtin_amdgcn_readlane(v, 1);
            w_br = __builtin_amdgcn_readlane(v, 2);
            failed |= (w_bl == LOOK_UNSET) | (w_b == LOOK_UNSET) | (w_br == LOOK_UNSET);
        }
        W.stamp(6);
        // slicetype.cpp:385-405: candidates right, below, below-left, below-right (those that exist),
        // appended in that order; unused entries 0 and 2 are zero vectors
        int cx[4] = { 0, 0, 0, 0 }, cy[4] = { 0, 0, 0, 0 }, numc = 0;
        auto push = [&](bool have, int word)
        {
            if (have)
            {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (numc == k)
                    {
                        cx[k] = XA_S16(word & 0xffff);
                        cy[k] = XA_S16(word >> 16);
                    }
                numc++;
            }
        };
        push(have_r, right);
        push(have_b, w_b);
        push(have_b && cux > 0, w_bl);
        push(have_b && have_r, w_br);

        JobRegs j;
        j.pu_x = 8 * cux;
        j.pu_y = 8 * cuy;
        j.w = j.h = 8;
        // search bounds that do not cross the extended frame (slicetype.cpp:366-369)
        j.mvmin_x = XA_S16(-cux * 8 - 8);
        j.mvmin_y = XA_S16(-cuy * 8 - 8);
        j.mvmax_x = XA_S16((cu_w - cux - 1) * 8 + 8);
        j.mvmax_y = XA_S16((cu_h - cuy - 1) * 8 + 8);
        j.merange = merange; j.method = method; j.subme = subme; j.qp = qp;
        j.numc = numc; j.ref_idx = 0; j.icost = X265AMD_SAD;
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            j.mvc[k][0] = cx[k];
            j.mvc[k][1] = cy[k];
        }
        if (numc <= 1)
        {
            j.qmvp_x = cx[0];
            j.qmvp_y = cy[0];
        }
        else
        {
            j.qmvp_x = look_median(cx[0], cx[1], cx[2]);
            j.qmvp_y = look_median(cy[0], cy[1], cy[2]);
        }

        Search<L> S(W, j);
        int ox, oy;
        const int c = S.run(nullptr, 0, ox, oy);
        n_evals += W.evals;
        right = (oy << 16) | (ox & 0xffff);
        LookLeaf::publish(res + (size_t)cuy * cu_w + cux, ox, oy, c);
#ifdef XA_PROFILE       // developer aid: clocks per phase, summed over the CUs of every 8th row
        W.stamp(7);
        if ((blockIdx.x & 7) == 0)
        {
            const int from[8] = { 0, 1, 6, 2, 3, 4, 5, 0 }, to[8] = { 1, 6, 2, 3, 4, 5, 7, 7 };
#pragma unroll
            for (int k = 0; k < 8; k++)
                prof[k] += W.tstamp[to[k]] - W.tstamp[from[k]];
        }
#endif
    }
    LookLeaf::finish(evals + (blockIdx.x % EVAL_SLOTS) * 16, n_evals, error, failed);
#ifdef XA_PROFILE
    if (threadIdx.x == 0 && (blockIdx.x & 7) == 0)
        for (int k = 0; k < 8; k++)
        {
            atomicAdd(evals + EVAL_SLOTS * 16 + 8 + k, (unsigned long long)prof[k]);
            atomicAdd(evals + (EVAL_SLOTS + 1) * 16 + k, (unsigned long long)W.cprof[k]);
        }
#endif
}

} // namespace xa
