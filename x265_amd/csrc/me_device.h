// Wave-cooperative block-cost and motion-search device code for gfx950 (wave64).
//
// NW wavefronts (one workgroup) own one PU.  The PU is cut into groups of 4 horizontally adjacent
// pixels; thread i of pass t owns group t*NW*64+i.  Groups are enumerated so that
//   * 4 consecutive lanes hold the 4 rows of one 4x4 tile  -> SATD's vertical butterfly is a
//     DPP quad permute, and
//   * (PUs whose sides are multiples of 8) 16 consecutive lanes hold one 8x8 tile and
//     (multiples of 16) 64 lanes hold one 16x16 tile -> SA8D's butterflies stay inside a
//     16-lane DPP row / the wave, and its per-8x8 / per-16x16 rounding is a lane-group sum.
// All search state (best MV, best cost, ...) is wave-uniform; only pixel work is per lane.
//
// Structure: the search logic (me_search.h) is scalar code inlined into the kernel.  The pixel work
// lives in three out-of-line functions (sad_pts, cost_any, stage_window) that exist once per kernel;
// everything they need - PU pixels, group coordinates, a window of the reference plane around the
// current search centre, slices of the MV cost table, and the per-job scalars (JobState) - is in
// LDS and addressed through one 32-bit LDS offset, so a call passes no pointers to private memory
// and an evaluation inside the window touches no global memory at all.
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

#define XA_LDS __attribute__((address_space(3)))

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

// ------------------------------------------------------------------ per-job state in LDS

enum { PRED_INT = 0, PRED_H = 1, PRED_V = 2, PRED_HV = 3 };

constexpr int CS_LEN = 640;         // entries per MV-cost slice kept in LDS (covers merange 64 in q-pel units + slack)
constexpr int WIN_MARGIN = 8;       // pixels of search excursion the LDS reference window keeps around a staging centre
constexpr int WIN_LOCAL = 6;        // a point set no wider than this re-centres the window; wider ones read the plane

enum   // JobState::flags
{
    JF_NO_WINDOW = 1,               // never stage a reference window (plain block costs; tuning aid)
    JF_NO_SLICES = 2                // read MV costs from the global table only (tuning aid)
};
enum   // sad_pts pricing flags
{
    PF_NOTRUNC = 1                  // do not truncate the sum of the two MV-cost entries to 16 bits
};

// Job scalars.  Every wave of the job keeps its own copy (all copies always hold the same values,
// because all waves run the same uniform control flow): no wave ever reads what another wrote.
struct JobState
{
    uint64_t refbase;               // address of the reference pixel under the PU origin (encoder/motion.cpp:288)
    uint64_t cost;                  // centre of the per-QP MV cost table (0: none)
    int32_t  gLo, gHi;              // element offsets relative to refbase of the whole padded allocation: [gLo, gHi)
    int32_t  stride, depth;
    int32_t  w, h, G, T;            // PU size, 4-pixel groups, passes over the job's lanes
    int32_t  mvpx, mvpy;            // q-pel predictor the MV costs are relative to
    int32_t  cs_lo[2];              // MVD of entry 0 of the x / y cost slice
    int32_t  flags, par;            // JF_*, parity of the next cross-wave reduction slot
    int32_t  wx0, wy0, ww, wh;      // reference window: origin relative to the PU origin, size (pixels)
    int32_t  wpitch, wvalid;        // window pitch in bytes; 0 while nothing is staged
    uint32_t evals, pad;
    int32_t  pts[8];                // candidates of the next sad_pts call: (y << 16) | (x & 0xffff), full-pel
    int32_t  lbl[8];                // their labels for MV pricing:        (y << 16) | (x & 0xffff), q-pel
};

// LDS of one job: NW waves, PUs of up to MAXG 4-pixel groups (MAXG = 1024 covers 64x64)
template <typename PIX_, int NW_, int MAXG_>
struct JobLds
{
    typedef PIX_ PIX;
    static constexpr int NW = NW_, MAXG = MAXG_, NT = NW_ * 64, DW = PixTraits<PIX_>::DW;
    // window bytes for 8-bit pixels: worst PU shape of the class with the full margin
    static constexpr int WIN_BYTES = (MAXG >= 1024 ? 8704 : MAXG >= 256 ? 4352 : 3328) * (int)sizeof(PIX);

    uint32_t fenc[MAXG * DW];                   // encoder pixels of group g (cached PU, encoder/motion.cpp:104-119)
    uint16_t gxy[MAXG];                         // (y << 8) | x of group g inside the PU
    int16_t  imm[MAXG * 4 + 7 * 64];            // horizontally filtered rows -3..h+3 of an HV candidate (stride w)
    int      red[2][NW][8];                     // cross-wave partial sums: [parity][wave][value]
    uint32_t cs[2][CS_LEN / 2];                 // MV-cost slices (uint16 entries): cost[lo_x + i], cost[lo_y + i]
    uint32_t win[WIN_BYTES / 4 + 4];            // reference window
    JobState st[NW];
};

struct Res8 { int c[8]; };

// Device code of one job class.  All entry points take the job's LDS offset.
template <class L>
struct Dev
{
    typedef typename L::PIX PIX;
    static constexpr int NW = L::NW, NT = L::NT, DW = L::DW;
    typedef XA_LDS L *LP;
    typedef XA_LDS JobState *SP;

    static __device__ __forceinline__ LP lds(uint32_t base) { return (LP)(uintptr_t)base; }
    static __device__ __forceinline__ int wave_id() { return NW > 1 ? uni((int)(threadIdx.x >> 6)) : 0; }

    static __device__ __forceinline__ void bsync()
    {
        if (NW > 1)
            __syncthreads();
        else
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

    // uniform snapshot of the job scalars a pixel function works with
    struct View
    {
        LP  l;
        SP  s;
        int tid, lane, wv;
        int w, h, G, T, stride, depth, gLo, gHi, flags, par;
        int wx0, wy0, ww, wh, wpitch, wvalid;
        const PIX *refbase;

        __device__ __forceinline__ View(uint32_t base)
        {
            l = lds(base);
            tid = (int)threadIdx.x;
            lane = tid & 63;
            wv = wave_id();
            s = &l->st[wv];
            refbase = (const PIX*)(((uint64_t)(uint32_t)uni((int)(s->refbase >> 32)) << 32) | (uint32_t)uni((int)s->refbase));
            gLo = uni(s->gLo); gHi = uni(s->gHi);
            stride = uni(s->stride); depth = uni(s->depth);
            w = uni(s->w); h = uni(s->h); G = uni(s->G); T = uni(s->T);
            flags = uni(s->flags); par = uni(s->par);
            load_window();
        }
        __device__ __forceinline__ void load_window()
        {
            wx0 = uni(s->wx0); wy0 = uni(s->wy0); ww = uni(s->ww); wh = uni(s->wh);
            wpitch = uni(s->wpitch); wvalid = uni(s->wvalid);
        }
        // does the window hold every pixel (with the 8-tap apron) of blocks with origins in the box?
        __device__ __forceinline__ bool covers(int minx, int miny, int maxx, int maxy) const
        {
            return wvalid && minx - 3 >= wx0 && maxx + w + 5 <= wx0 + ww && miny - 3 >= wy0 && maxy + h + 4 <= wy0 + wh;
        }

        // sum N per-lane values over all lanes of the job; results uniform and equal in every wave
        template <int N>
        __device__ __forceinline__ void job_sum(int *v)
        {
#pragma unroll
            for (int k = 0; k < N; k++)
                v[k] = wave_sum(v[k]);
            if (NW > 1)
            {
                if (lane == 0)
                {
#pragma unroll
                    for (int k = 0; k < N; k++)
                        l->red[par][wv][k] = v[k];
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < N; k++)
                {
                    int t = (lane < NW) ? l->red[par][lane][k] : 0;
                    if (NW > 8)  t += xor8(t);
                    if (NW > 4)  t += xor4(t);
                    if (NW > 2)  t += xor2(t);
                    t += xor1(t);
                    v[k] = uni(t);
                }
                par ^= 1;     // the next reduction uses the other slot: one barrier per reduction suffices
                s->par = par;
            }
        }

        // ---- pixel fetch.  (px,py) is relative to the PU origin.
        // 4 pixels as packed dwords: u8 -> r[0]; u16 -> r[0], r[1]
        template <bool LW>
        __device__ __forceinline__ void rd4_packed(int px, int py, uint32_t *r) const
        {
            if (LW)
            {
                const int a = (py - wy0) * wpitch + (px - wx0) * (int)sizeof(PIX);
                const XA_LDS uint32_t *d = l->win + (a >> 2);
                r[0] = __builtin_amdgcn_alignbyte(d[1], d[0], a & 3);
                if (DW == 2)
                    r[1] = __builtin_amdgcn_alignbyte(d[2], d[1], a & 3);
            }
            else
            {
                const PIX *q = refbase + clamp_off(py * stride + px, 4);
                if (DW == 1)
                    r[0] = ld_u32(q);
                else
                {
                    uint64_t v = ld_u64(q);
                    r[0] = (uint32_t)v;
                    r[1] = (uint32_t)(v >> 32);
                }
            }
        }
        static __device__ __forceinline__ P4 unpack4(const uint32_t *d)
        {
            P4 r;
            if (DW == 1)
            {
                r.v[0] = d[0] & 0xff; r.v[1] = (d[0] >> 8) & 0xff; r.v[2] = (d[0] >> 16) & 0xff; r.v[3] = d[0] >> 24;
            }
            else
            {
                r.v[0] = d[0] & 0xffff; r.v[1] = d[0] >> 16; r.v[2] = d[1] & 0xffff; r.v[3] = d[1] >> 16;
            }
            return r;
        }
        template <bool LW>
        __device__ __forceinline__ P4 rd4(int px, int py) const
        {
            uint32_t d[2];
            rd4_packed<LW>(px, py, d);
            return unpack4(d);
        }
        // 12 consecutive pixels starting at (px,py)
        template <bool LW>
        __device__ __forceinline__ void rd12(int px, int py, int *o) const
        {
            if (LW)
            {
                const int a = (py - wy0) * wpitch + (px - wx0) * (int)sizeof(PIX);
                const XA_LDS uint32_t *d = l->win + (a >> 2);
                const int sh = a & 3;
                if (DW == 1)
                {
#pragma unroll
                    for (int i = 0; i < 3; i++)
                    {
                        uint32_t v = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
                        o[4 * i] = v & 0xff; o[4 * i + 1] = (v >> 8) & 0xff; o[4 * i + 2] = (v >> 16) & 0xff; o[4 * i + 3] = v >> 24;
                    }
                }
                else
                {
#pragma unroll
                    for (int i = 0; i < 6; i++)
                    {
                        uint32_t v = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
                        o[2 * i] = v & 0xffff; o[2 * i + 1] = v >> 16;
                    }
                }
            }
            else
                load12(refbase + clamp_off(py * stride + px, 12), o);
        }
        // per-access clamp of the direct path: a read of `len` pixels stays inside the allocation
        __device__ __forceinline__ int clamp_off(int o, int len) const
        {
            o = o < gLo ? gLo : o;
            return o > gHi - len ? gHi - len : o;
        }

        __device__ __forceinline__ P4 fenc4(int g) const
        {
            uint32_t d[2];
            d[0] = l->fenc[DW * g];
            if (DW == 2)
                d[1] = l->fenc[DW * g + 1];
            return unpack4(d);
        }

        // ---- prediction fetch
        // 4 prediction pixels of the group at (gx,gy) inside a candidate block whose integer origin is
        // (cx,cy), for interpolation mode `mode` / phase (xf,yf); filters follow common/ipfilter.cpp
        // with hr = 14 - depth.  PRED_HV reads the horizontally filtered intermediate from LDS (see
        // stage_hv).  `mode` is uniform: the switch is a scalar branch.
        template <bool LW>
        __device__ __forceinline__ P4 pred4(int mode, int cx, int cy, int gv, int xf, int yf) const
        {
            const int gx = gv & 0xff, gy = gv >> 8;
            if (mode == PRED_INT)
                return rd4<LW>(cx + gx, cy + gy);
            const int hr = 14 - depth;
            const int maxval = (1 << depth) - 1;
            P4 r;
            if (mode == PRED_H)        // filterHorizontal_p_p, ipfilter.cpp:91-139
            {
                int px[12];
                rd12<LW>(cx + gx - 3, cy + gy, px);
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
            if (mode == PRED_V)        // filterVertical_p_p, ipfilter.cpp:354-425
            {
                int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int k = 0; k < 8; k++)
                {
                    P4 row = rd4<LW>(cx + gx, cy + gy + k - 3);
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
            // PRED_HV, second half: filterVertical_s_p (ipfilter.cpp:38-89) over the LDS intermediate,
            // whose row k holds picture row (k - 3) of the candidate block
            {
                const int sh3 = 6 + hr;
                const int off3 = (1 << (sh3 - 1)) + (8192 << 6);
                const XA_LDS int16_t *m = l->imm + gy * w + gx;
                int sum[4] = { 0, 0, 0, 0 };
#pragma unroll
                for (int k = 0; k < 8; k++)
                {
                    const uint64_t d = *(const XA_LDS uint64_t*)(m + k * w);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        sum[i] += (int)(int16_t)(d >> (16 * i)) * c_luma[yf][k];
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

        // PRED_HV, first half: filterHorizontal_p_s (ipfilter.cpp:265-309) of rows -3 .. h+3 of the
        // candidate block at (cx,cy), shared by all lanes through LDS (as the reference shares
        // immedVal, encoder/motion.cpp:1158-1161)
        template <bool LW>
        __device__ __forceinline__ void stage_hv(int cx, int cy, int xf) const
        {
            const int hr = 14 - depth;
            const int sh2 = 6 - hr;
            const int off2 = -8192 * (1 << sh2);
            const int ngx = w >> 2;
            const int n = (h + 7) * ngx;
            for (int idx = tid; idx < n; idx += NT)
            {
                int row = idx / ngx, col = (idx - row * ngx) * 4;
                int px[12];
                rd12<LW>(cx + col - 3, cy + row - 3, px);
                uint64_t packed = 0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    int sm = 0;
#pragma unroll
                    for (int t = 0; t < 8; t++)
                        sm += px[i + t] * c_luma[xf][t];
                    int hv = (int16_t)((int)(int16_t)(sm + off2) >> sh2);
                    packed |= (uint64_t)(uint16_t)hv << (16 * i);
                }
                *(XA_LDS uint64_t*)(l->imm + row * w + col) = packed;
            }
            bsync();
        }
    };

    // ------------------------------------------------------------------ Hadamard partials

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

    // ------------------------------------------------------------------ out-of-line pixel functions
    // (Everything with per-lane control flow lives in these functions: the kernel body itself then
    // has only uniform branches and compiles to scalar code.)

    // cache the PU (encoder/motion.cpp:104-119): fill fenc / gxy from the encoder plane
    static __device__ __noinline__ void fill_pu(uint32_t base, uint64_t enc_addr, int enc_stride_)
    {
        View v(base);
        const int enc_stride = uni(enc_stride_);
        const PIX *e = (const PIX*)(((uint64_t)(uint32_t)uni((int)(enc_addr >> 32)) << 32) | (uint32_t)uni((int)enc_addr));
        for (int g = v.tid; g < v.G; g += NT)
        {
            int x, y;
            group_xy(g, v.w, v.h, x, y);
            v.l->gxy[g] = (uint16_t)((y << 8) | x);
            const PIX *p = e + (int64_t)y * enc_stride + x;
            if (DW == 1)
                v.l->fenc[g] = ld_u32(p);
            else
            {
                uint64_t d = ld_u64(p);
                v.l->fenc[2 * g] = (uint32_t)d;
                v.l->fenc[2 * g + 1] = (uint32_t)(d >> 32);
            }
        }
        bsync();
    }

    // copy the slices [cs_lo, cs_lo + CS_LEN) of the MV cost table into LDS
    static __device__ __noinline__ void stage_slices(uint32_t base)
    {
        View v(base);
        const uint16_t *cost = (const uint16_t*)(((uint64_t)(uint32_t)uni((int)(v.s->cost >> 32)) << 32) | (uint32_t)uni((int)v.s->cost));
        const int lo[2] = { uni(v.s->cs_lo[0]), uni(v.s->cs_lo[1]) };
        for (int i = v.tid; i < CS_LEN; i += NT)
        {
            const int axis = i >= CS_LEN / 2, k = i - axis * (CS_LEN / 2);
            // entries past +-32767 wrap like the (int16_t) index of a direct lookup would
            const int m0 = (int16_t)(lo[axis] + 2 * k), m1 = (int16_t)(lo[axis] + 2 * k + 1);
            v.l->cs[axis][k] = (uint32_t)cost[m0] | ((uint32_t)cost[m1] << 16);
        }
        bsync();
    }

    // one lane publishes the job's result and evaluation count
    static __device__ __noinline__ void publish(uint32_t base, x265amd_me_result *out, int mvx, int mvy, int cost,
                                                unsigned long long *evals)
    {
        View v(base);
        if (v.tid == 0)
        {
            x265amd_me_result r;
            r.mvx = (int16_t)mvx;
            r.mvy = (int16_t)mvy;
            r.cost = cost;
            *out = r;
            atomicAdd(evals, (unsigned long long)v.s->evals);
        }
    }

    // (Re)stage the reference window so that it covers every block (with the 8-tap apron) whose
    // origin lies in [minx,maxx] x [miny,maxy], with as much of WIN_MARGIN around as fits.
    // Returns 0 if even margin 0 does not fit; the caller then reads the plane directly.
    static __device__ __noinline__ int stage_window(uint32_t base, int minx_, int miny_, int maxx_, int maxy_)
    {
        const int minx = uni(minx_), miny = uni(miny_), maxx = uni(maxx_), maxy = uni(maxy_);
        View v(base);
        const int nw = maxx - minx + v.w + 8, nh = maxy - miny + v.h + 7;
        int m = WIN_MARGIN;
        auto bytes = [&](int mm) { return ((((nw + 2 * mm) * (int)sizeof(PIX) + 3) & ~3) + 4) * (nh + 2 * mm); };
        while (m > 0 && bytes(m) > L::WIN_BYTES)
            m--;
        if (bytes(m) > L::WIN_BYTES)
            return 0;
        const int wx0 = minx - 3 - m, wy0 = miny - 3 - m, ww = nw + 2 * m, wh = nh + 2 * m;
        const int wpitch = (((ww * (int)sizeof(PIX)) + 3) & ~3) + 4;
        // the previous window's readers are all past their reduction barrier, except (NW > 1) readers
        // of a function that ended without one; a barrier here covers those
        if (NW > 1)
            __syncthreads();
        const int pdw = wpitch >> 2, n = pdw * wh;
        for (int i = v.tid; i < n; i += NT)
        {
            int r = i / pdw, c = i - r * pdw;
            int o = (wy0 + r) * v.stride + wx0 + c * (4 / (int)sizeof(PIX));
            o = o < v.gLo ? v.gLo : o;
            o = o > v.gHi - 4 ? v.gHi - 4 : o;
            v.l->win[i] = ld_u32(v.refbase + o);
        }
        v.s->wx0 = wx0; v.s->wy0 = wy0; v.s->ww = ww; v.s->wh = wh;
        v.s->wpitch = wpitch; v.s->wvalid = 1;
        bsync();
        return 1;
    }

    template <int N, bool LW>
    static __device__ __forceinline__ void sad_loop(const View &v, const int *p, int *acc)
    {
        for (int t = 0; t < v.T; t++)
        {
            const int g = t * NT + v.tid;
            if (g < v.G)
            {
                const int gv = v.l->gxy[g], gx = gv & 0xff, gy = gv >> 8;
                uint32_t f[2];
                f[0] = v.l->fenc[DW * g];
                if (DW == 2)
                    f[1] = v.l->fenc[DW * g + 1];
#pragma unroll
                for (int k = 0; k < N; k++)
                {
                    uint32_t r[2];
                    v.template rd4_packed<LW>((int)(int16_t)p[k] + gx, (p[k] >> 16) + gy, r);
                    if (DW == 1)
                        acc[k] = __builtin_amdgcn_sad_u8(f[0], r[0], acc[k]);
                    else
                    {
                        acc[k] = __builtin_amdgcn_sad_u16(f[0], r[0], acc[k]);
                        acc[k] = __builtin_amdgcn_sad_u16(f[1], r[1], acc[k]);
                    }
                }
            }
        }
    }

    // SAD (common/pixel.cpp:78-142, the sad / sad_x3 / sad_x4 primitives) of the n <= 8 full-pel
    // candidates in JobState::pts, each plus the MV cost of its label in JobState::lbl
    // (COST_MV / COST_MV_X4, encoder/motion.cpp:173-249).  Entries k >= n of the result are undefined.
    static __device__ __noinline__ Res8 sad_pts(uint32_t base, int n_, int pflags_)
    {
        const int n = uni(n_), pflags = uni(pflags_);
        View v(base);
        int p[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            p[k] = uni(v.s->pts[k]);
        int minx = (int16_t)p[0], maxx = minx, miny = p[0] >> 16, maxy = miny;
#pragma unroll
        for (int k = 1; k < 8; k++)
        {
            if (k >= n)
                p[k] = p[0];                  // padding: re-measure candidate 0 (never reported)
            int x = (int16_t)p[k], y = p[k] >> 16;
            minx = x < minx ? x : minx; maxx = x > maxx ? x : maxx;
            miny = y < miny ? y : miny; maxy = y > maxy ? y : maxy;
        }
        // MV cost of label k, computed by lane k while the SADs are in flight
        int mv = 0;
        if (v.lane < n)
        {
            const int lb = v.s->lbl[v.lane];
            const int dx = (int16_t)((int16_t)lb - v.s->mvpx), dy = (int16_t)((lb >> 16) - v.s->mvpy);
            const uint16_t *tab = (const uint16_t*)v.s->cost;
            const unsigned ix = (unsigned)(dx - v.s->cs_lo[0]), iy = (unsigned)(dy - v.s->cs_lo[1]);
            const XA_LDS uint16_t *cs = (const XA_LDS uint16_t*)&v.l->cs[0][0];
            int a = ix < (unsigned)CS_LEN ? cs[ix] : tab[dx];
            int b = iy < (unsigned)CS_LEN ? cs[CS_LEN + iy] : tab[dy];
            mv = (pflags & PF_NOTRUNC) ? a + b : ((a + b) & 0xffff);
        }
        bool lw = false;
        if (!(v.flags & JF_NO_WINDOW))
        {
            lw = v.covers(minx, miny, maxx, maxy);
            if (!lw && maxx - minx <= WIN_LOCAL && maxy - miny <= WIN_LOCAL)
            {
                lw = stage_window(base, minx, miny, maxx, maxy) != 0;
                v.load_window();
            }
        }
        int acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (n == 1)
        {
            if (lw) sad_loop<1, true>(v, p, acc);
            else    sad_loop<1, false>(v, p, acc);
            v.template job_sum<1>(acc);
        }
        else if (n <= 4)
        {
            if (lw) sad_loop<4, true>(v, p, acc);
            else    sad_loop<4, false>(v, p, acc);
            v.template job_sum<4>(acc);
        }
        else
        {
            if (lw) sad_loop<8, true>(v, p, acc);
            else    sad_loop<8, false>(v, p, acc);
            v.template job_sum<8>(acc);
        }
        v.s->evals += (uint32_t)n;
        Res8 r;
#pragma unroll
        for (int k = 0; k < 8; k++)
            r.c[k] = acc[k] + __builtin_amdgcn_readlane(mv, k);
        return r;
    }

    // Block cost of kind `kind` (sad / satd / sa8d_inter) of the cached PU against the candidate with
    // integer origin (cx,cy) and fractional phase (xf,yf): subpelCompare for a full-resolution
    // reference (encoder/motion.cpp:1140-1165) when the phase is non-zero.  One body for all kinds
    // and interpolation modes (uniform switches) keeps the instruction footprint small.
    static __device__ __noinline__ int cost_any(uint32_t base, int cx_, int cy_, int frac_, int kind_)
    {
        const int cx = uni(cx_), cy = uni(cy_), kind = uni(kind_);
        const int xf = uni(frac_) & 3, yf = uni(frac_) >> 2;
        View v(base);
        const int mode = (xf ? PRED_H : 0) | (yf ? PRED_V : 0);
        const bool m8 = ((v.w | v.h) & 7) == 0;
        const bool m16 = ((v.w | v.h) & 15) == 0;
        bool lw = false;
        if (!(v.flags & JF_NO_WINDOW))
        {
            lw = v.covers(cx, cy, cx, cy);
            if (!lw)
            {
                lw = stage_window(base, cx, cy, cx, cy) != 0;
                v.load_window();
            }
        }
        if (mode == PRED_HV)
        {
            if (lw) v.template stage_hv<true>(cx, cy, xf);
            else    v.template stage_hv<false>(cx, cy, xf);
        }
        // which accumulation: 0 sad, 1 4x4 Hadamard (halved at the end), 2 8x8 with 8x8 rounding, 3 8x8 with 16x16 rounding
        const int am = kind == X265AMD_SAD ? 0 : (kind == X265AMD_SATD || !m8) ? 1 : (m16 ? 3 : 2);
        int acc[2] = { 0, 0 };      // [0] per-lane partials, [1] sums already rounded per 16x16 (lane 0 only)
        for (int t = 0; t < v.T; t++)
        {
            const int g = t * NT + v.tid;
            int d[4] = { 0, 0, 0, 0 };
            if (g < v.G)
            {
                const int gv = v.l->gxy[g];
                P4 f = v.fenc4(g);
                P4 r = lw ? v.template pred4<true>(mode, cx, cy, gv, xf, yf) : v.template pred4<false>(mode, cx, cy, gv, xf, yf);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    d[i] = f.v[i] - r.v[i];
            }
            if (am == 0)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                    acc[0] += d[i] < 0 ? -d[i] : d[i];
            }
            else if (am == 1)
                acc[0] += had4_partial(d, v.lane);
            else
            {
                int sm = had8_partial(d, v.lane);
                if (am == 3)
                {
                    int s64 = wave_sum(sm);                                  // sa8d16: one rounding per 16x16 (= one wave pass)
                    acc[1] += (v.lane == 0) ? ((s64 + 2) >> 2) : 0;
                }
                else
                {
                    int s16 = sum16(sm);                                     // sa8d8: one rounding per 8x8
                    acc[0] += ((v.lane & 15) == 0) ? ((s16 + 2) >> 2) : 0;
                }
            }
        }
        v.s->evals += 1;
        v.template job_sum<2>(acc);
        int tot = acc[0];
        if (am == 1)
            tot >>= 1;                                                       // every 4x4 |H| sum is even
        return tot + acc[1];
    }
};

// ------------------------------------------------------------------ caller side

// What the (scalar, inlined) search logic holds of a job: the LDS offset and the few scalars it
// needs itself.  Everything else lives in LDS (JobState).
template <class L>
struct Wave
{
    typedef typename L::PIX PIX;
    typedef Dev<L> D;
    static constexpr int NW = L::NW, NT = L::NT, DW = L::DW;

    uint32_t base;
    typename D::LP l;
    typename D::SP s;
    int tid, w, h, mvpx, mvpy, ikind;
    int cs_lo[2];
    const uint16_t *cost;

    // bind the PU: fill fenc / gxy / JobState.  (enc_x,enc_y) / (ref_x,ref_y): PU origin in the two planes.
    __device__ __forceinline__ void bind(L &lds_obj, const PlaneDesc &enc, const PlaneDesc &ref, int enc_x, int enc_y, int ref_x, int ref_y,
                         int w_, int h_, const uint16_t *cost_, int flags)
    {
        l = (typename D::LP)&lds_obj;
        base = (uint32_t)(uintptr_t)l;
        tid = (int)threadIdx.x;
        s = &l->st[D::wave_id()];
        w = w_; h = h_;
        cost = cost_;
        ikind = X265AMD_SAD;
        mvpx = mvpy = 0;
        cs_lo[0] = cs_lo[1] = 0x7fffffff;
        const int G = (w * h) >> 2;
        const int64_t blockOffset = (int64_t)ref_y * ref.stride + ref_x;
        s->refbase = (uint64_t)((const PIX*)ref.origin + blockOffset);
        s->cost = (uint64_t)cost_;
        s->gLo = (int32_t)(ref.lo - blockOffset);
        s->gHi = (int32_t)(ref.hi - blockOffset);
        s->stride = ref.stride; s->depth = ref.depth;
        s->w = w; s->h = h; s->G = G; s->T = (G + NT - 1) / NT;
        s->mvpx = 0; s->mvpy = 0;
        s->cs_lo[0] = s->cs_lo[1] = 0x7fffffff;
        s->flags = flags; s->par = 0;
        s->wx0 = s->wy0 = s->ww = s->wh = s->wpitch = s->wvalid = 0;
        s->evals = 0;
        const PIX *e = (const PIX*)enc.origin + (int64_t)enc_y * enc.stride + enc_x;
        D::fill_pu(base, (uint64_t)e, enc.stride);
    }

    __device__ __forceinline__ unsigned evals() const { return (unsigned)uni((int)s->evals); }
    __device__ __forceinline__ void add_evals(unsigned n) { s->evals += n; }

    // MV cost table entry for MVD `mvd` on an axis: from the LDS slice when inside it
    __device__ __forceinline__ int cost_entry(int axis, int mvd) const
    {
        unsigned i = (unsigned)(mvd - cs_lo[axis]);
        if (i < (unsigned)CS_LEN)
            return uni(((const XA_LDS uint16_t*)&l->cs[0][0])[axis * CS_LEN + i]);
        return uni(cost[(int16_t)mvd]);
    }
    // mvcost(MV(x,y)) for q-pel x,y: uint16 sum of two table entries (encoder/bitcost.h:46)
    __device__ __forceinline__ int mvcost(int x, int y) const
    {
        int a = cost_entry(0, (int16_t)(x - mvpx));
        int b = cost_entry(1, (int16_t)(y - mvpy));
        return (a + b) & 0xffff;
    }

    // setMVP (encoder/bitcost.h:43) + keep the part of the MV-cost table the search will index in
    // LDS: MVDs of q-pel MVs in [4*(mvmin-8), ...) relative to the predictor
    __device__ __forceinline__ void set_mvp(int mvmin_x, int mvmin_y, int mvpx_, int mvpy_, int flags)
    {
        mvpx = mvpx_;
        mvpy = mvpy_;
        s->mvpx = mvpx; s->mvpy = mvpy;
        if (flags & JF_NO_SLICES)
            return;
        // even first index: the slice is copied as aligned dwords (two entries each)
        cs_lo[0] = (4 * (mvmin_x - 8) - mvpx) & ~1;
        cs_lo[1] = (4 * (mvmin_y - 8) - mvpy) & ~1;
        s->cs_lo[0] = cs_lo[0]; s->cs_lo[1] = cs_lo[1];
        D::stage_slices(base);
    }

    // Integer-stage cost (the reference's `sad`) of N full-pel candidates measured at (sx,sy)[k],
    // plus the MV cost of their labels (lx,ly)[k] (full-pel; priced at << 2, or << 3 for the one
    // label index `shift3`).
    template <int N>
    __device__ __forceinline__ void eval_points(const int *sx, const int *sy, const int *lx, const int *ly, int *out,
                                                int pflags = 0, int shift3 = -1)
    {
        int qx[N], qy[N];
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            const int sh = (k == shift3) ? 3 : 2;
            qx[k] = (int16_t)((int16_t)lx[k] << sh);
            qy[k] = (int16_t)((int16_t)ly[k] << sh);
        }
        if (ikind == X265AMD_SAD)
        {
#pragma unroll
            for (int k = 0; k < N; k++)
            {
                s->pts[k] = (sy[k] << 16) | (sx[k] & 0xffff);
                s->lbl[k] = (qy[k] << 16) | (qx[k] & 0xffff);
            }
            Res8 r = D::sad_pts(base, N, pflags);
#pragma unroll
            for (int k = 0; k < N; k++)
                out[k] = uni(r.c[k]);
        }
        else
        {
#pragma unroll
            for (int k = 0; k < N; k++)
            {
                int c = uni(D::cost_any(base, sx[k], sy[k], 0, ikind));
                int a = cost_entry(0, (int16_t)(qx[k] - mvpx)), b = cost_entry(1, (int16_t)(qy[k] - mvpy));
                out[k] = c + ((pflags & PF_NOTRUNC) ? a + b : ((a + b) & 0xffff));
            }
        }
    }

    // COST_MV's value: cost at full-pel (mx,my) + mvcost(MV(mx,my) << 2)   (motion.cpp:173-179)
    __device__ __forceinline__ int cost_fpel(int mx, int my)
    {
        int c;
        eval_points<1>(&mx, &my, &mx, &my, &c);
        return c;
    }

    // subpelCompare for a full-resolution reference (motion.cpp:1140-1165), no MV cost
    __device__ __forceinline__ int subpel_cost(int qx, int qy, int kind)
    {
        return uni(D::cost_any(base, qx >> 2, qy >> 2, (qx & 3) | ((qy & 3) << 2), kind));
    }

    // plain block cost against the block at the reference origin
    __device__ __forceinline__ int cost_direct(int kind) { return uni(D::cost_any(base, 0, 0, 0, kind)); }
};

} // namespace xa
