// Wave-cooperative block-cost and motion-search device code for gfx950 (wave64).
//
// One wavefront owns one PU.  The PU is cut into groups of 4 horizontally adjacent pixels;
// lane l of pass t owns group t*64+l.  Groups are enumerated so that
//   * 4 consecutive lanes hold the 4 rows of one 4x4 tile  -> SATD's vertical butterfly is a
//     DPP quad permute, and
//   * (PUs whose sides are multiples of 8) 16 consecutive lanes hold one 8x8 tile and
//     (multiples of 16) 64 lanes hold one 16x16 tile -> SA8D's butterflies stay inside a
//     16-lane DPP row / the wave, and its per-8x8 / per-16x16 rounding is a lane-group sum.
// All search state (best MV, best cost, ...) is wave-uniform; only pixel work is per lane.
//
// Reference semantics being reproduced (paths relative to the reference's source/):
//   sad / satd / sa8d_inter      common/pixel.cpp:78-95, 187-274, 276-362, 645-826
//   8-tap luma interpolation     common/ipfilter.cpp:38-139, 265-309, 354-425
//   mvcost                       encoder/bitcost.h:43-46
//   search + sub-pel refinement  encoder/motion.cpp:277-1191
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/x265_amd.h"

namespace xa {

// Device-visible description of a registered plane.
struct PlaneDesc
{
    const void *origin;   // address of pixel (0,0)
    int64_t lo, hi;       // element offsets (relative to origin) of the padded allocation: [lo, hi)
    int32_t stride;       // pixels
    int32_t width, height;
    int32_t bpp;          // bytes per pixel
    int32_t depth;        // bit depth seen by the interpolation filters
    int32_t margin_x, margin_y;
};

struct P4 { int v[4]; };

__constant__ int c_luma[4][8] = {   // HEVC luma taps, Lib/TLibCommon/TComRom.cpp:176-182
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 } };

// ------------------------------------------------------------------ cross-lane helpers

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <int CTRL>
__device__ __forceinline__ int dpp(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
// partner exchange lane ^ 1 / ^ 2 (quad permutes), ^ 8 (rotate a 16-lane row by 8), ^ 4 / ^ 16 (swizzle), ^ 32
__device__ __forceinline__ int xor1(int v) { return dpp<0xB1>(v); }
__device__ __forceinline__ int xor2(int v) { return dpp<0x4E>(v); }
__device__ __forceinline__ int xor8(int v) { return dpp<0x128>(v); }
__device__ __forceinline__ int xor4(int v) { return __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1f); }
__device__ __forceinline__ int xor16(int v) { return __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1f); }
__device__ __forceinline__ int xor32(int v) { return __shfl_xor(v, 32); }

__device__ __forceinline__ int sum16(int v)   // sum over the 16-lane row, result in every lane of the row
{
    v += xor1(v); v += xor2(v); v += xor4(v); v += xor8(v);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)  // sum over the wave, returned wave-uniform
{
    v = sum16(v);
    v += xor16(v);
    v += xor32(v);
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ------------------------------------------------------------------ pixel access

template <typename PIX> struct PixTraits;
template <> struct PixTraits<uint8_t>  { static constexpr int DW = 1; };   // dwords per 4-pixel group
template <> struct PixTraits<uint16_t> { static constexpr int DW = 2; };

__device__ __forceinline__ uint32_t ld_u32(const void *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);   // byte-aligned dword load (global memory allows it on gfx950)
    return v;
}
__device__ __forceinline__ uint64_t ld_u64(const void *p)
{
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__device__ __forceinline__ P4 load4(const uint8_t *p)
{
    uint32_t d = ld_u32(p);
    P4 r;
    r.v[0] = d & 0xff; r.v[1] = (d >> 8) & 0xff; r.v[2] = (d >> 16) & 0xff; r.v[3] = d >> 24;
    return r;
}
__device__ __forceinline__ P4 load4(const uint16_t *p)
{
    uint64_t d = ld_u64(p);
    P4 r;
    r.v[0] = (int)(d & 0xffff); r.v[1] = (int)((d >> 16) & 0xffff);
    r.v[2] = (int)((d >> 32) & 0xffff); r.v[3] = (int)(d >> 48);
    return r;
}
// 12 consecutive pixels starting at p (the 11 a 4-output 8-tap row needs, +1)
__device__ __forceinline__ void load12(const uint8_t *p, int *o)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        uint32_t d = ld_u32(p + 4 * i);
        o[4 * i] = d & 0xff; o[4 * i + 1] = (d >> 8) & 0xff; o[4 * i + 2] = (d >> 16) & 0xff; o[4 * i + 3] = d >> 24;
    }
}
__device__ __forceinline__ void load12(const uint16_t *p, int *o)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        uint64_t d = ld_u64(p + 4 * i);
        o[4 * i] = (int)(d & 0xffff); o[4 * i + 1] = (int)((d >> 16) & 0xffff);
        o[4 * i + 2] = (int)((d >> 32) & 0xffff); o[4 * i + 3] = (int)(d >> 48);
    }
}

// position (x,y) inside the PU of 4-pixel group g (see file header for the ordering)
__device__ __forceinline__ void group_xy(int g, int w, int h, int &x, int &y)
{
    if (((w | h) & 7) == 0)
    {
        int blk = g >> 4, in = g & 15;
        int row = (in & 3) | ((in >> 3) << 2), half = (in >> 2) & 1;
        int bx8, by8;
        if (((w | h) & 15) == 0)
        {
            int sb = blk >> 2, q = blk & 3, nsx = w >> 4;
            bx8 = (sb % nsx) * 2 + (q & 1);
            by8 = (sb / nsx) * 2 + (q >> 1);
        }
        else
        {
            int nbx = w >> 3;
            bx8 = blk % nbx;
            by8 = blk / nbx;
        }
        x = bx8 * 8 + half * 4;
        y = by8 * 8 + row;
    }
    else
    {
        int blk = g >> 2, row = g & 3, nbx = w >> 2;
        x = (blk % nbx) * 4;
        y = (blk / nbx) * 4 + row;
    }
}

// ------------------------------------------------------------------ per-wave PU state

enum { PRED_INT = 0, PRED_H = 1, PRED_V = 2, PRED_HV = 3 };

template <typename PIX>
struct Wave
{
    static constexpr int DW = PixTraits<PIX>::DW;

    // LDS slices owned by this wave
    uint32_t *fenc;        // [G * DW]  encoder pixels of group g (cached PU, encoder/motion.cpp:104-119)
    int      *roff;        // [G]       y*refStride + x of group g

    const PIX *refbase;    // ref origin + blockOffset  (encoder/motion.cpp:288)
    int64_t   offLo, offHi; // clamp range for candidate offsets relative to refbase (memory safety)
    int       stride;      // reference stride, pixels
    int       w, h, G, T;  // PU size, groups, passes
    int       depth;
    const uint16_t *cost;  // centre of the per-QP MV cost table
    int       mvpx, mvpy;
    int       ikind;       // integer-stage cost kind
    int       lane;
    unsigned  evals;

    // bind the PU: fill fenc / roff
    // `apron`: reserve the 8-tap filter apron in the candidate clamp (search); without it only the
    // block itself has to stay inside the allocation (plain block costs)
    __device__ void bind(const PlaneDesc &enc, const PlaneDesc &ref, int enc_x, int enc_y, int ref_x, int ref_y,
                         int w_, int h_, bool apron)
    {
        lane = lane_id();
        w = w_; h = h_;
        G = (w * h) >> 2;
        T = (G + 63) >> 6;
        stride = ref.stride;
        depth = ref.depth;
        ikind = X265AMD_SAD;
        mvpx = mvpy = 0;
        const PIX *e = (const PIX*)enc.origin + (int64_t)enc_y * enc.stride + enc_x;
        int64_t blockOffset = (int64_t)ref_y * ref.stride + ref_x;
        refbase = (const PIX*)ref.origin + blockOffset;
        if (apron)
        {
            // a candidate's 8-tap apron reaches 3 rows / 3 pixels before and (h+4) rows / (w+8) pixels after its origin
            offLo = ref.lo + 3 * (int64_t)stride + 3 - blockOffset;
            offHi = ref.hi - ((int64_t)(h + 4) * stride + w + 8) - blockOffset;
        }
        else
        {
            offLo = ref.lo - blockOffset;
            offHi = ref.hi - ((int64_t)(h - 1) * stride + w) - blockOffset;
        }
        for (int g = lane; g < G; g += 64)
        {
            int x, y;
            group_xy(g, w, h, x, y);
            roff[g] = y * stride + x;
            const PIX *p = e + (int64_t)y * enc.stride + x;
            if (DW == 1)
                fenc[g] = ld_u32(p);
            else
            {
                uint64_t d = ld_u64(p);
                fenc[2 * g] = (uint32_t)d;
                fenc[2 * g + 1] = (uint32_t)(d >> 32);
            }
        }
        evals = 0;
        __builtin_amdgcn_s_waitcnt(0);       // LDS writes of this wave are visible to its own later reads
        __builtin_amdgcn_wave_barrier();
    }

    __device__ __forceinline__ P4 fenc4(int g) const
    {
        P4 r;
        if (DW == 1)
        {
            uint32_t d = fenc[g];
            r.v[0] = d & 0xff; r.v[1] = (d >> 8) & 0xff; r.v[2] = (d >> 16) & 0xff; r.v[3] = d >> 24;
        }
        else
        {
            uint32_t a = fenc[2 * g], b = fenc[2 * g + 1];
            r.v[0] = a & 0xffff; r.v[1] = a >> 16; r.v[2] = b & 0xffff; r.v[3] = b >> 16;
        }
        return r;
    }

    // candidate offset (relative to refbase) of full-pel MV (mx,my), clamped into the allocation
    __device__ __forceinline__ int64_t off_of(int mx, int my) const
    {
        int64_t o = (int64_t)my * stride + mx;
        o = o < offLo ? offLo : o;
        o = o > offHi ? offHi : o;
        return o;
    }

    // mvcost(MV(x,y)) for q-pel x,y: uint16 sum of two table entries (encoder/bitcost.h:46)
    __device__ __forceinline__ int mvcost(int x, int y) const
    {
        int a = cost[(int16_t)(x - mvpx)];
        int b = cost[(int16_t)(y - mvpy)];
        return uni((a + b) & 0xffff);
    }

    // -------------------------------------------------------------- prediction fetch
    // 4 prediction pixels of group g for a candidate whose integer origin is `p` (= refbase + off)
    // and whose fractional phase is (xf,yf); filters follow common/ipfilter.cpp with hr = 14 - depth.
    template <int MODE>
    __device__ __forceinline__ P4 pred4(const PIX *p, int ro, int xf, int yf) const
    {
        const PIX *q = p + ro;
        if (MODE == PRED_INT)
            return load4(q);
        const int hr = 14 - depth;
        const int maxval = (1 << depth) - 1;
        P4 r;
        if (MODE == PRED_H)        // filterHorizontal_p_p, ipfilter.cpp:91-139
        {
            int px[12];
            load12(q - 3, px);
            const int offset = 1 << (hr - 1);
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                int sum = 0;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    sum += px[i + k] * c_luma[xf][k];
                int val = (int)(int16_t)(sum + offset) >> hr;
                val = (int16_t)val;
                val = val < 0 ? 0 : val;
                r.v[i] = val > maxval ? maxval : val;
            }
            return r;
        }
        if (MODE == PRED_V)        // filterVertical_p_p, ipfilter.cpp:354-425
        {
            int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                P4 row = load4(q + (int64_t)(k - 3) * stride);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    sum[i] += row.v[i] * c_luma[yf][k];
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                int val = (int16_t)((sum[i] + 32) >> 6);
                val = val < 0 ? 0 : val;
                r.v[i] = val > maxval ? maxval : val;
            }
            return r;
        }
        // PRED_HV: filterHorizontal_p_s (ipfilter.cpp:265-309) over rows -3..+4, then
        // filterVertical_s_p (ipfilter.cpp:38-89)
        {
            const int sh2 = 6 - hr;
            const int off2 = -8192 * (1 << sh2);
            const int sh3 = 6 + hr;
            const int off3 = (1 << (sh3 - 1)) + (8192 << 6);
            int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                int px[12];
                load12(q + (int64_t)(k - 3) * stride - 3, px);
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    int s = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        s += px[i + t] * c_luma[xf][t];
                    int hv = (int16_t)((int)(int16_t)(s + off2) >> sh2);
                    sum[i] += hv * c_luma[yf][k];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                int val = (int16_t)((sum[i] + off3) >> sh3);
                val = val < 0 ? 0 : val;
                r.v[i] = val > maxval ? maxval : val;
            }
            return r;
        }
    }

    // -------------------------------------------------------------- cost kernels (wave-uniform results)

    // per-lane |Hadamard| partial sums of one pass for SATD: d = fenc - pred of this lane's 4 pixels
    static __device__ __forceinline__ int had4_partial(const int *d, int lane)
    {
        int a0 = d[0] + d[1], a1 = d[0] - d[1], a2 = d[2] + d[3], a3 = d[2] - d[3];
        int v[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            int p = xor1(v[i]);
            int t = (lane & 1) ? p - v[i] : v[i] + p;
            p = xor2(t);
            t = (lane & 2) ? p - t : t + p;
            s += t < 0 ? -t : t;
        }
        return s;
    }

    // same for the 8x8 transform: lanes of a 16-lane row = (row&3 | half<<2 | (row>>2)<<3)
    static __device__ __forceinline__ int had8_partial(const int *d, int lane)
    {
        int a0 = d[0] + d[1], a1 = d[0] - d[1], a2 = d[2] + d[3], a3 = d[2] - d[3];
        int v[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            int t = v[i], p;
            p = xor4(t); t = (lane & 4) ? p - t : t + p;    // horizontal: the two 4-pixel halves
            p = xor1(t); t = (lane & 1) ? p - t : t + p;    // vertical: rows ^1, ^2, ^4
            p = xor2(t); t = (lane & 2) ? p - t : t + p;
            p = xor8(t); t = (lane & 8) ? p - t : t + p;
            s += t < 0 ? -t : t;
        }
        return s;
    }

    // block cost of kind KIND against a candidate given by integer origin offset `off` and phase
    template <int KIND, int MODE>
    __device__ int cost_block(int64_t off, int xf, int yf)
    {
        const PIX *p = refbase + off;
        const bool m8 = ((w | h) & 7) == 0;
        const bool m16 = ((w | h) & 15) == 0;
        int acc = 0, uacc = 0;
        for (int t = 0; t < T; t++)
        {
            int g = t * 64 + lane;
            int d[4] = { 0, 0, 0, 0 };
            if (g < G)
            {
                P4 f = fenc4(g);
                P4 r = pred4<MODE>(p, roff[g], xf, yf);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    d[i] = f.v[i] - r.v[i];
            }
            if (KIND == X265AMD_SAD)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc += d[i] < 0 ? -d[i] : d[i];
            }
            else if (KIND == X265AMD_SATD || !m8)
                acc += had4_partial(d, lane);
            else
            {
                int s = had8_partial(d, lane);
                if (m16)
                    uacc += (wave_sum(s) + 2) >> 2;                       // sa8d16: one rounding per 16x16
                else
                {
                    int s16 = sum16(s);                                   // sa8d8: one rounding per 8x8
                    acc += ((lane & 15) == 0) ? ((s16 + 2) >> 2) : 0;
                }
            }
        }
        evals++;
        int tot = wave_sum(acc);
        if (KIND == X265AMD_SATD || (KIND == X265AMD_SA8D && !m8))
            tot >>= 1;                                                    // every 4x4 |H| sum is even
        return tot + uacc;
    }

    template <int MODE>
    __device__ int cost_kind(int kind, int64_t off, int xf, int yf)
    {
        if (kind == X265AMD_SAD) return cost_block<X265AMD_SAD, MODE>(off, xf, yf);
        if (kind == X265AMD_SATD) return cost_block<X265AMD_SATD, MODE>(off, xf, yf);
        return cost_block<X265AMD_SA8D, MODE>(off, xf, yf);
    }

    // SADs of N full-pel candidates at once (sad / sad_x3 / sad_x4 semantics, common/pixel.cpp:78-142)
    template <int N>
    __device__ void sad_points(const int *mx, const int *my, int *out)
    {
        const PIX *p[N];
        int acc[N];
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            p[k] = refbase + off_of(mx[k], my[k]);
            acc[k] = 0;
        }
        for (int t = 0; t < T; t++)
        {
            int g = t * 64 + lane;
            if (g < G)
            {
                int ro = roff[g];
                if (DW == 1)
                {
                    uint32_t f = fenc[g];
#pragma unroll
                    for (int k = 0; k < N; k++)
                        acc[k] = __builtin_amdgcn_sad_u8(f, ld_u32(p[k] + ro), acc[k]);
                }
                else
                {
                    uint32_t f0 = fenc[2 * g], f1 = fenc[2 * g + 1];
#pragma unroll
                    for (int k = 0; k < N; k++)
                    {
                        uint64_t r = ld_u64(p[k] + ro);
                        acc[k] = __builtin_amdgcn_sad_u16(f0, (uint32_t)r, acc[k]);
                        acc[k] = __builtin_amdgcn_sad_u16(f1, (uint32_t)(r >> 32), acc[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < N; k++)
            out[k] = wave_sum(acc[k]);
        evals += N;
    }

    // integer-stage cost (the reference's `sad`) of N full-pel candidates
    template <int N>
    __device__ void icost_points(const int *mx, const int *my, int *out)
    {
        if (ikind == X265AMD_SAD)
            sad_points<N>(mx, my, out);
        else
        {
#pragma unroll
            for (int k = 0; k < N; k++)
                out[k] = cost_kind<PRED_INT>(ikind, off_of(mx[k], my[k]), 0, 0);
        }
    }

    // COST_MV's value: cost at full-pel (mx,my) + mvcost(MV(mx,my) << 2)   (motion.cpp:173-179)
    __device__ int cost_fpel(int mx, int my)
    {
        int c;
        icost_points<1>(&mx, &my, &c);
        return c + mvcost((int16_t)(mx << 2), (int16_t)(my << 2));
    }

    // subpelCompare for a full-resolution reference (motion.cpp:1140-1165)
    __device__ int subpel_cost(int qx, int qy, int kind)
    {
        int64_t off = off_of(qx >> 2, qy >> 2);
        int xf = qx & 3, yf = qy & 3;
        if (!(xf | yf)) return cost_kind<PRED_INT>(kind, off, 0, 0);
        if (!yf)        return cost_kind<PRED_H>(kind, off, xf, 0);
        if (!xf)        return cost_kind<PRED_V>(kind, off, 0, yf);
        return cost_kind<PRED_HV>(kind, off, xf, yf);
    }
};

} // namespace xa
