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
    int64_t lr_step;      // 0, or: this is the F plane of a half-resolution set whose H, V, C planes follow
                          // 1, 2, 3 x lr_step elements behind it (ReferencePlanes::lowresPlane, common/reference.h:36-53)
};

struct P4 { int v[4]; };

static __constant__ int c_luma[4][8] = {   // HEVC luma taps, Lib/TLibCommon/TComRom.cpp:176-182
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 } };

// the same taps packed for the dot-product instructions: four signed bytes per dword (taps 0-3, 4-7)
// and two signed 16-bit values per dword (taps 0-1, 2-3, 4-5, 6-7)
#define XA_B4(a, b, c, d) (int)(((a) & 0xffu) | (((b) & 0xffu) << 8) | (((c) & 0xffu) << 16) | (((unsigned)(d) & 0xffu) << 24))
#define XA_H2(a, b) (int)(((a) & 0xffffu) | ((unsigned)(b) << 16))
static __constant__ int c_luma_b4[4][2] = {
    { XA_B4(0, 0, 0, 64),   XA_B4(0, 0, 0, 0) },
    { XA_B4(-1, 4, -10, 58), XA_B4(17, -5, 1, 0) },
    { XA_B4(-1, 4, -11, 40), XA_B4(40, -11, 4, -1) },
    { XA_B4(0, 1, -5, 17),  XA_B4(58, -10, 4, -1) } };
static __constant__ int c_luma_h2[4][4] = {
    { XA_H2(0, 0),  XA_H2(0, 64),   XA_H2(0, 0),   XA_H2(0, 0) },
    { XA_H2(-1, 4), XA_H2(-10, 58), XA_H2(17, -5), XA_H2(1, 0) },
    { XA_H2(-1, 4), XA_H2(-11, 40), XA_H2(40, -11), XA_H2(4, -1) },
    { XA_H2(0, 1),  XA_H2(-5, 17),  XA_H2(58, -10), XA_H2(4, -1) } };

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

// row_half_mirror / row_mirror: lane i of a 16-lane row reads lane (i ^ 7) / (i ^ 15) ... of the same row
__device__ __forceinline__ int sum16(int v)   // sum over the 16-lane row, result in every lane of the row
{
    v += xor1(v); v += xor2(v);
    v += dpp<0x141>(v);                       // row_half_mirror: the other quad of the 8-lane half
    v += dpp<0x140>(v);                       // row_mirror: the other half of the row
    return v;
}
__device__ __forceinline__ int wave_sum_last(int v)  // sum over the wave; valid in lane 63 only
{
    v = sum16(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_sum(int v)  // sum over the wave, returned wave-uniform
{
    return __builtin_amdgcn_readlane(wave_sum_last(v), 63);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Sums of N <= 8 per-lane values over the wavefront, all at once: lane l receives the total of value l & (M - 1),
// M = N rounded up to a power of two.  While a lane still holds several values, each exchange step halves them: the
// lane keeps the values whose index has the step's bit equal to its own lane-number bit and adds the partner's partial
// sums of those (partner = lane ^ 1, lane ^ 2, then the lane 4 away in the row rotated by 4: it keeps the other half and
// every lane is some lane's partner, so no partial is lost or counted twice).  The remaining steps are plain adds of
// the lanes 4 / 8 away in the 16-lane row, the same lane of the neighbouring row (v_permlane16_swap) and of the other
// half of the wavefront (v_permlane32_swap).  26 instructions for 8 values against 8 x 6 dependent DPP adds.
template <int N>
__device__ __forceinline__ int wave_sum_lanes(const int *v)
{
    static_assert(N >= 1 && N <= 8, "1 to 8 values");
    constexpr int M = N <= 1 ? 1 : N <= 2 ? 2 : N <= 4 ? 4 : 8;
    const int lane = lane_id();
    int a[8];
#pragma unroll
    for (int k = 0; k < 8; k++)
        a[k] = k < N ? v[k] : 0;
    if (M > 1)
    {
        const bool b = (lane & 1) != 0;
#pragma unroll
        for (int k = 0; k < M / 2; k++)
            a[k] = (b ? a[2 * k + 1] : a[2 * k]) + xor1(b ? a[2 * k] : a[2 * k + 1]);
    }
    else
        a[0] += xor1(a[0]);
    if (M > 2)
    {
        const bool b = (lane & 2) != 0;
#pragma unroll
        for (int k = 0; k < M / 4; k++)
            a[k] = (b ? a[2 * k + 1] : a[2 * k]) + xor2(b ? a[2 * k] : a[2 * k + 1]);
    }
    else
        a[0] += xor2(a[0]);
    if (M > 4)
    {
        const bool b = (lane & 4) != 0;
        a[0] = (b ? a[1] : a[0]) + dpp<0x124>(b ? a[0] : a[1]);      // row_ror:4
    }
    else
        a[0] += dpp<0x124>(a[0]);
    int t = a[0] + xor8(a[0]);
    auto r16 = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
    t = (int)(r16[0] + r16[1]);
    auto r32 = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
    return (int)(r32[0] + r32[1]);
}

// ------------------------------------------------------------------ pixel access

template <typename PIX> struct PixTraits;
template <> struct PixTraits<uint8_t>  { static constexpr int DW = 1; };   // dwords per 4-pixel group
template <> struct PixTraits<uint16_t> { static constexpr int DW = 2; };

// Byte-aligned dword / qword loads of plane memory (gfx950 allows them).  The pointer is cast to the
// global address space: a generic pointer would make these FLAT loads, whose results are waited for
// on the LDS counter as well as the memory counter.
#define XA_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ uint32_t ld_u32(const void *p)
{
    struct __attribute__((packed)) U { uint32_t v; };
    return ((const XA_GLOBAL U*)(uintptr_t)p)->v;
}
__device__ __forceinline__ uint64_t ld_u64(const void *p)
{
    struct __attribute__((packed)) U { uint64_t v; };
    return ((const XA_GLOBAL U*)(uintptr_t)p)->v;
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

// Phase word of a candidate (JobState::frc).  Full-resolution reference: xf | yf << 2.  Half-resolution
// set (subpelCompare's isLowres branch, encoder/motion.cpp:1112-1137): LR_FLAG | plane A | plane B << 2 |
// average << 4 | (B's block origin relative to A's: dx << 5 | dy << 6).
enum { LR_FLAG = 0x80 };

constexpr int CS_LEN = 160;         // entries of the MV-cost slice kept in LDS: MVDs within +-20 pixels of the predictor
constexpr int CS_LO = -CS_LEN / 2;  // MVD of its first entry (a multiple of 8: 16-byte aligned copies)
constexpr int WIN_MARGIN = 8;       // pixels of search excursion the LDS reference window keeps around a staging centre
constexpr int WIN_LOCAL = 6;        // a point set no wider than this re-centres the window; wider ones read the plane

enum   // job flags (BatchParams::flags, tuning aids)
{
    JF_NO_WINDOW = 1,               // never stage a reference window
    JF_NO_SLICES = 2                // read MV costs from the global table only
};
enum   // per-call flags of the pixel functions
{
    CF_WINDOW  = 1,                 // candidate offsets address the LDS window (bytes); else the plane (pixels)
    CF_NOTRUNC = 2,                 // do not truncate the sum of the two MV-cost entries to 16 bits
    CF_PAR     = 4,                 // parity of the (first) cross-wave reduction slot to use
    CF_NOMV    = 8,                 // do not add MV costs
    CF_NOMV0   = 16                 // do not add the MV cost of candidate 0
};

// Job scalars the out-of-line pixel functions read.  Written by the job's leader wave only, and only
// while the other waves wait at the command barrier (see Dev::worker).
struct JobState
{
    uint64_t refbase;               // address of the reference pixel under the PU origin (encoder/motion.cpp:288)
    uint64_t cost;                  // centre of the per-QP MV cost table (0: none)
    int32_t  gLo, gHi;              // element offsets relative to refbase of the whole padded allocation: [gLo, gHi)
    int32_t  stride, depth;
    int32_t  w, h, G, T;            // PU size, 4-pixel groups, passes over the job's lanes
    int32_t  mvpx, mvpy;            // q-pel predictor the MV costs are relative to
    int32_t  sliced, lrstep;        // the LDS cost slice is staged; PlaneDesc::lr_step of the reference
    int32_t  wx0, wy0, ww, wh;      // reference window as last staged: origin relative to the PU origin, size (pixels)
    int32_t  wpitch, pad;           // its pitch in bytes
    int32_t  pts[8];                // candidates of the next eval call: offset of the block origin (window bytes / plane pixels)
    int32_t  lbl[8];                // their labels for MV pricing: (y << 16) | (x & 0xffff), q-pel
    int32_t  frc[8];                // their fractional phases xf | yf << 2 (eval_any)
#ifdef XA_PROFILE
    long long cpt[4];               // developer aid: clock at entry of eval_any, after the MV costs, after the pixel work, at exit
#endif
};

// LDS of one job: NW waves, PUs of up to MAXG 4-pixel groups (MAXG = 1024 covers 64x64).
// LR: the kernel also handles half-resolution reference sets (PlaneDesc::lr_step != 0); kernels
// built without it are leaner and must not be given such references.
// INL: a single-wavefront kernel that takes the evaluation code inline instead of calling it (latency-bound chains).
// CP: a single-wavefront class (NW = 1: the search and every evaluation run on the job's first wavefront, inline) whose workgroup
// holds up to CP_WAVES - 1 further wavefronts parked at a barrier (the launch's block size decides how many); a sub-pel round's
// candidates are dealt to all of them, each wavefront pricing its candidates over the whole PU (Dev::cp_round).  For chains of
// dependent searches of ONE PU at a time (me_field.h).
template <typename PIX_, int NW_, int MAXG_, bool LR_ = false, bool INL_ = false, bool CP_ = false>
struct JobLds
{
    typedef PIX_ PIX;
    static constexpr int NW = NW_, MAXG = MAXG_, NT = NW_ * 64, DW = PixTraits<PIX_>::DW;
    static constexpr bool LR = LR_, INL = INL_, CP = CP_;
    static constexpr int CP_WAVES = CP_ ? 8 : 1;
    static_assert(!CP_ || NW_ == 1, "candidate-parallel rounds are a feature of the single-wavefront classes");
    // A window of the reference plane around the search centre is kept in LDS only by the class of the largest PUs:
    // measured on MI355X, staging it costs the smaller classes more instructions than reading candidates from the
    // plane (L2) does (8x8 .. 32x32 PUs: 3 - 14 % faster without, 64x64: 6 - 16 % slower without).
    static constexpr bool USE_WIN = MAXG >= 1024;
    // window bytes for 8-bit pixels: the square PU of the class with the full margin (long thin PUs get less margin)
    static constexpr int WIN_BYTES = USE_WIN ? 8704 * (int)sizeof(PIX) : 0;

    uint32_t fenc[MAXG * DW];                   // encoder pixels of group g (cached PU, encoder/motion.cpp:104-119)
    uint16_t gxy[MAXG];                         // (y << 8) | x of group g inside the PU
    int16_t  imm[MAXG * 4 + 8 * 64];            // horizontally filtered rows -3..h+4 of an HV candidate (stride w)
    int      red[2][8][NW];                     // cross-wave partial sums: [parity][value][wave]
    uint32_t cs[CS_LEN / 2];                    // MV-cost slice (uint16 entries): cost[CS_LO + i], used for both axes
    uint32_t win[WIN_BYTES / 4 + 4];            // reference window
    JobState st;
    int      cmd[8];                            // command posted by the leader wave: op, arguments (NW > 1)
    // candidate-parallel rounds: the horizontal intermediates of wavefronts 1 .. CP_WAVES - 1, the command {go, n, flags, kind}
    // and the candidates' costs
    int16_t  cp_imm[CP_ ? CP_WAVES - 1 : 1][CP_ ? MAXG * 4 + 8 * 64 : 4];
    int      cp_cmd[4];
    int      cp_res[8];
};

struct Res8 { int c[8]; };

// Device code of one job class.  All entry points take the job's LDS offset.  Scalar arguments
// arrive in vector registers (the calling convention has no scalar arguments) but hold the same
// value in every lane; values that steer control flow are moved to scalar registers with uni().
template <class L>
struct Dev
{
    typedef typename L::PIX PIX;
    static constexpr int NW = L::NW, NT = L::NT, DW = L::DW, PB = (int)sizeof(PIX);
    typedef XA_LDS L *LP;
    typedef XA_LDS JobState *SP;

    static __device__ __forceinline__ LP lds(uint32_t base) { return (LP)(uintptr_t)base; }
    static __device__ __forceinline__ int wave_id() { return NW > 1 ? uni((int)(threadIdx.x >> 6)) : 0; }
    static __device__ __forceinline__ uint64_t uni64(uint64_t v)
    {
        return ((uint64_t)(uint32_t)uni((int)(v >> 32)) << 32) | (uint32_t)uni((int)v);
    }

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

    // what every pixel function starts from
    struct View
    {
        LP  l;
        SP  s;
        XA_LDS int16_t *imm;        // horizontal intermediate of HV candidates (a candidate-parallel worker has its own)
        int tid, lane, wv, G, T;

        __device__ __forceinline__ View(uint32_t base)
        {
            l = lds((uint32_t)uni((int)base));
            imm = l->imm;
            tid = NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x;     // (a candidate-parallel workgroup has more threads than NT)
            lane = tid & 63;
            wv = wave_id();
            s = &l->st;
            G = uni(s->G);
            T = uni(s->T);
        }

        // Sum N per-lane values over all lanes of the job: lane k < N of the job's first wave receives the total of
        // value k (the other waves receive nothing and return 0).  One barrier per reduction: consecutive reductions
        // alternate between two slots (`par`).
        template <int N>
        __device__ __forceinline__ int job_sum_lanes(const int *v, int par)
        {
            int t = wave_sum_lanes<N>(v);
            if (NW == 1)
                return t;
            if (lane < N)
                l->red[par][lane][wv] = t;
            __syncthreads();
            if (wv != 0)
                return 0;                   // only the leader wave consumes results
            // lane k + 8 i sums the partials of value k held by waves i*NW/S .. (S = lanes per value)
            constexpr int S = NW > 8 ? 8 : NW, PER = NW / S;
            t = 0;
            if (lane < 8 * S)
            {
                const XA_LDS int *q = &l->red[par][lane & 7][(lane >> 3) * PER];
#pragma unroll
                for (int i = 0; i < PER; i++)
                    t += q[i];
            }
            if (S > 1) t += xor8(t);
            if (S > 2)
            {
                auto r = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
                t = (int)(r[0] + r[1]);
            }
            if (S > 4)
            {
                auto r = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
                t = (int)(r[0] + r[1]);
            }
            return t;
        }
        // the same with uniform results in v[] (equal in every lane of the job's first wave)
        template <int N>
        __device__ __forceinline__ void job_sum(int *v, int par)
        {
            const int t = job_sum_lanes<N>(v, par);
#pragma unroll
            for (int k = 0; k < N; k++)
                v[k] = __builtin_amdgcn_readlane(t, k);
        }
    };

    // ---- pixel fetch.  `off` addresses 4 (12) consecutive pixels: a byte offset into the LDS window
    // (LW) or a pixel offset relative to refbase, clamped into the allocation.
    struct Src
    {
        const XA_LDS uint32_t *win;
        const PIX *refbase;
        int gLo, gHi;

        __device__ __forceinline__ Src(const View &v, bool lw)
        {
            win = v.l->win;
            refbase = nullptr;
            gLo = gHi = 0;
            if (!lw)
            {
                // (same value in every lane, but only ever used in per-lane address arithmetic)
                refbase = (const PIX*)v.s->refbase;
                gLo = v.s->gLo;
                gHi = v.s->gHi;
            }
        }
        __device__ __forceinline__ int clamp_off(int o, int len) const
        {
            o = o < gLo ? gLo : o;
            return o > gHi - len ? gHi - len : o;
        }
        // 4 pixels as packed dwords: u8 -> r[0]; u16 -> r[0], r[1]
        template <bool LW>
        __device__ __forceinline__ void rd4_packed(int off, uint32_t *r) const
        {
            if (LW)
            {
                const XA_LDS uint32_t *d = win + (off >> 2);
                r[0] = __builtin_amdgcn_alignbyte(d[1], d[0], off);
                if (DW == 2)
                    r[1] = __builtin_amdgcn_alignbyte(d[2], d[1], off);
            }
            else
            {
                const PIX *q = refbase + clamp_off(off, 4);
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
        // 4 pixels of another plane of a half-resolution set: `add` elements behind the clamped position
        __device__ __forceinline__ void rd4_plane(int off, int add, uint32_t *r) const
        {
            const PIX *q = refbase + clamp_off(off, 4) + add;
            if (DW == 1)
                r[0] = ld_u32(q);
            else
            {
                uint64_t v = ld_u64(q);
                r[0] = (uint32_t)v;
                r[1] = (uint32_t)(v >> 32);
            }
        }
        // 12 pixels as packed dwords: u8 -> r[0..2]; u16 -> r[0..5]
        template <bool LW>
        __device__ __forceinline__ void rd12_packed(int off, uint32_t *r) const
        {
            if (LW)
            {
                const XA_LDS uint32_t *d = win + (off >> 2);
#pragma unroll
                for (int i = 0; i < 3 * DW; i++)
                    r[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], off);
            }
            else
            {
                const PIX *q = refbase + clamp_off(off, 12);
#pragma unroll
                for (int i = 0; i < 3 * DW; i++)
                    r[i] = ld_u32((const uint8_t*)q + 4 * i);
            }
        }
    };

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

    // 8-tap filter of 4 adjacent outputs over 12 packed pixels: out[i] = sum_k c[k] * px[i + k].
    // The taps come packed for the dot-product instructions: 8-bit pixels use two 4-way signed byte
    // dots per output on pixels biased by -128 (the bias, 128 * sum(c) = 8192, is added back);
    // 16-bit pixels use four 2-way 16-bit dots.
    struct Taps8                // 8-bit pixels: c0..c3, c4..c7 as signed bytes
    {
        int lo, hi;
        __device__ __forceinline__ Taps8(int f) : lo(c_luma_b4[f][0]), hi(c_luma_b4[f][1]) {}
    };
    struct Taps16               // 16-bit values: (c0,c1) (c2,c3) (c4,c5) (c6,c7) as signed 16-bit pairs
    {
        int p[4];
        __device__ __forceinline__ Taps16(int f)
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
                p[i] = c_luma_h2[f][i];
        }
    };
    typedef short short2v __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ int dot2(uint32_t a, int b, int acc)
    {
        short2v x, y;
        __builtin_memcpy(&x, &a, 4);
        __builtin_memcpy(&y, &b, 4);
        return __builtin_amdgcn_sdot2(x, y, acc, false);
    }
    // out[i] = bias + sum_k c[k] * px[i + k] for 4 adjacent outputs over 12 packed pixels
    static __device__ __forceinline__ void filt8x4(const uint32_t *r, int f, int bias, int *out)
    {
        if (DW == 1)
        {
            const Taps8 t(f);
            const uint32_t b0 = r[0] ^ 0x80808080u, b1 = r[1] ^ 0x80808080u, b2 = r[2] ^ 0x80808080u;
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const uint32_t a = i ? __builtin_amdgcn_alignbyte(b1, b0, i) : b0;
                const uint32_t b = i ? __builtin_amdgcn_alignbyte(b2, b1, i) : b1;
                int sm = __builtin_amdgcn_sdot4((int)a, t.lo, 8192 + bias, false);
                out[i] = __builtin_amdgcn_sdot4((int)b, t.hi, sm, false);
            }
        }
        else
        {
            const Taps16 t(f);
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                // pixels i .. i+7 start at 16-bit element i: dword i/2, odd i straddles dwords
                int sm = bias;
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int e = i + 2 * k;
                    const uint32_t pr = (e & 1) ? __builtin_amdgcn_alignbyte(r[e / 2 + 1], r[e / 2], 2) : r[e / 2];
                    sm = dot2(pr, t.p[k], sm);
                }
                out[i] = sm;
            }
        }
    }

    // two rows (a above b) of 4 packed 16-bit values each, times the tap pair (c[k], c[k+1]), added per column
    static __device__ __forceinline__ void vpair16(const uint32_t *a, const uint32_t *b, int taps, int *sum)
    {
        sum[0] = dot2(__builtin_amdgcn_perm(b[0], a[0], 0x05040100), taps, sum[0]);
        sum[1] = dot2(__builtin_amdgcn_perm(b[0], a[0], 0x07060302), taps, sum[1]);
        sum[2] = dot2(__builtin_amdgcn_perm(b[1], a[1], 0x05040100), taps, sum[2]);
        sum[3] = dot2(__builtin_amdgcn_perm(b[1], a[1], 0x07060302), taps, sum[3]);
    }
    // vertical 8-tap over 8 rows of 4 packed pixels: sum[i] = bias + sum_k c[k] * row_k[i]
    static __device__ __forceinline__ void vfilt8x4(const uint32_t (*rw)[2], int f, int bias, int *sum)
    {
        if (DW == 1)
        {
            const Taps8 t(f);
            // byte transposes turn 4 rows x 4 columns into one dword per column, then two 4-way dots
            uint32_t col[2][4];
#pragma unroll
            for (int hf = 0; hf < 2; hf++)
            {
                const uint32_t r0 = rw[4 * hf][0] ^ 0x80808080u, r1 = rw[4 * hf + 1][0] ^ 0x80808080u;
                const uint32_t r2 = rw[4 * hf + 2][0] ^ 0x80808080u, r3 = rw[4 * hf + 3][0] ^ 0x80808080u;
                const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400), t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602);
                const uint32_t u0 = __builtin_amdgcn_perm(r3, r2, 0x05010400), u1 = __builtin_amdgcn_perm(r3, r2, 0x07030602);
                col[hf][0] = __builtin_amdgcn_perm(u0, t0, 0x05040100); col[hf][1] = __builtin_amdgcn_perm(u0, t0, 0x07060302);
                col[hf][2] = __builtin_amdgcn_perm(u1, t1, 0x05040100); col[hf][3] = __builtin_amdgcn_perm(u1, t1, 0x07060302);
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
                sum[i] = __builtin_amdgcn_sdot4((int)col[1][i], t.hi, __builtin_amdgcn_sdot4((int)col[0][i], t.lo, 8192 + bias, false), false);
        }
        else
        {
            const Taps16 t(f);
#pragma unroll
            for (int i = 0; i < 4; i++)
                sum[i] = bias;
#pragma unroll
            for (int k = 0; k < 8; k += 2)
                vpair16(rw[k], rw[k + 1], t.p[k / 2], sum);
        }
    }
    // the reference's `short` intermediates: a no-op for 8-bit pixels, whose sums all fit 16 bits
    static __device__ __forceinline__ int as_short(int v) { return DW == 1 ? v : (int)(int16_t)v; }
    static __device__ __forceinline__ int clamp_pix(int v, int maxval)
    {
        v = v < 0 ? 0 : v;
        return v > maxval ? maxval : v;
    }

    // ---- the reference's four luma filter primitives on one group of 4 outputs (common/ipfilter.cpp
    // with hr = headRoom = 14 - depth); the search's prediction code and the stand-alone filter
    // kernel (k_ipfilter) both go through these.
    // ipfilter_pp[H_8] = filterHorizontal_p_p (ipfilter.cpp:91-139): 12 packed pixels -> 4 pixels
    static __device__ __forceinline__ void h_pp(const uint32_t *px, int xf, int depth, int *out)
    {
        const int hr = 14 - depth, maxval = (1 << depth) - 1;
        int sum[4];
        filt8x4(px, xf, 1 << (hr - 1), sum);
#pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = clamp_pix(as_short(as_short(sum[i]) >> hr), maxval);      // cast to short before the shift (:129)
    }
    // ipfilter_pp[V_8] = filterVertical_p_p (ipfilter.cpp:354-425): 8 rows of 4 packed pixels -> 4 pixels
    static __device__ __forceinline__ void v_pp(const uint32_t (*rw)[2], int yf, int depth, int *out)
    {
        const int maxval = (1 << depth) - 1;
        int sum[4];
        vfilt8x4(rw, yf, 32, sum);
#pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = clamp_pix(as_short(sum[i] >> 6), maxval);
    }
    // ipfilter_ps[H_8] = filterHorizontal_p_s (ipfilter.cpp:265-309): 12 packed pixels -> 4 `short` values
    static __device__ __forceinline__ void h_ps(const uint32_t *px, int xf, int depth, int *out)
    {
        const int hr = 14 - depth, sh2 = 6 - hr;
        int sm[4];
        filt8x4(px, xf, -8192 * (1 << sh2), sm);
#pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = as_short(as_short(sm[i]) >> sh2);                          // cast before the shift (:302)
    }
    // ipfilter_sp[V_8] = filterVertical_s_p (ipfilter.cpp:38-89): 8 rows of 4 packed `short` values -> 4 pixels
    static __device__ __forceinline__ void v_sp(const uint32_t (*rw)[2], int yf, int depth, int *out)
    {
        const int hr = 14 - depth, maxval = (1 << depth) - 1;
        const int sh3 = 6 + hr, off3 = (1 << (sh3 - 1)) + (8192 << 6);
        const Taps16 ty(yf);
        int sum[4] = { off3, off3, off3, off3 };
#pragma unroll
        for (int k = 0; k < 8; k += 2)
            vpair16(rw[k], rw[k + 1], ty.p[k / 2], sum);
#pragma unroll
        for (int i = 0; i < 4; i++)
            out[i] = clamp_pix(as_short(sum[i] >> sh3), maxval);
    }

    // ------------------------------------------------------------------ Hadamard partials
    // d = fenc - pred of this lane's 4 pixels.  The row transform is done inside the lane; the column
    // transform runs across lanes, each stage as t = +-t + partner(t) with the sign a per-lane
    // constant.  With 8-bit pixels every intermediate of the 4x4 and 8x8 transforms fits 16 bits, so
    // the cross-lane stages, the absolute values and the sums work on two values per register.

    static __device__ __forceinline__ short2v as_s2(int v) { short2v r; __builtin_memcpy(&r, &v, 4); return r; }
    static __device__ __forceinline__ int as_i(short2v v) { int r; __builtin_memcpy(&r, &v, 4); return r; }
    template <int CTRL> static __device__ __forceinline__ short2v dpp2(short2v v) { return as_s2(dpp<CTRL>(as_i(v))); }
    static __device__ __forceinline__ int abs_sum2(short2v a, short2v b)      // |a.x| + |a.y| + |b.x| + |b.y|
    {
        typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
        const short2v ma = __builtin_elementwise_max(a, -a), mb = __builtin_elementwise_max(b, -b);
        ushort2v ua, ub, one = { 1, 1 };
        __builtin_memcpy(&ua, &ma, 4);
        __builtin_memcpy(&ub, &mb, 4);
        return (int)__builtin_amdgcn_udot2(ua, one, __builtin_amdgcn_udot2(ub, one, 0u, false), false);
    }

    // per-lane |Hadamard| partial sums of one pass for SATD (4 lanes = the 4 rows of a 4x4 tile)
    static __device__ __forceinline__ int had4_partial(const int *d, int lane)
    {
        int a0 = d[0] + d[1], a1 = d[0] - d[1], a2 = d[2] + d[3], a3 = d[2] - d[3];
        int v[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
        if (DW == 1)
        {
            const short2v s1 = as_s2((lane & 1) ? -1 : 0x00010001), s2 = as_s2((lane & 2) ? -1 : 0x00010001);
            short2v p = as_s2(__builtin_amdgcn_perm(v[1], v[0], 0x05040100)), q = as_s2(__builtin_amdgcn_perm(v[3], v[2], 0x05040100));
            p = p * s1 + dpp2<0xB1>(p); q = q * s1 + dpp2<0xB1>(q);      // lanes ^1: rows r, r^1 -> sum / difference
            p = p * s2 + dpp2<0x4E>(p); q = q * s2 + dpp2<0x4E>(q);      // lanes ^2
            return abs_sum2(p, q);
        }
        const int s1 = (lane & 1) ? -1 : 1, s2 = (lane & 2) ? -1 : 1;
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            int t = v[i] * s1 + xor1(v[i]);
            t = t * s2 + xor2(t);
            s += t < 0 ? -t : t;
        }
        return s;
    }

    // same for the 8x8 transform: lanes of a 16-lane row = (row&3 | half<<2 | (row>>2)<<3)
    static __device__ __forceinline__ int had8_partial(const int *d, int lane)
    {
        int a0 = d[0] + d[1], a1 = d[0] - d[1], a2 = d[2] + d[3], a3 = d[2] - d[3];
        int v[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
        if (DW == 1)
        {
            const short2v s1 = as_s2((lane & 1) ? -1 : 0x00010001), s2 = as_s2((lane & 2) ? -1 : 0x00010001);
            const short2v s4 = as_s2((lane & 4) ? -1 : 0x00010001), s8 = as_s2((lane & 8) ? -1 : 0x00010001);
            short2v p = as_s2(__builtin_amdgcn_perm(v[1], v[0], 0x05040100)), q = as_s2(__builtin_amdgcn_perm(v[3], v[2], 0x05040100));
            p = p * s4 + as_s2(xor4(as_i(p))); q = q * s4 + as_s2(xor4(as_i(q)));   // horizontal: the two 4-pixel halves
            p = p * s1 + dpp2<0xB1>(p);  q = q * s1 + dpp2<0xB1>(q);                // vertical: rows ^1, ^2, ^4
            p = p * s2 + dpp2<0x4E>(p);  q = q * s2 + dpp2<0x4E>(q);
            p = p * s8 + dpp2<0x128>(p); q = q * s8 + dpp2<0x128>(q);
            return abs_sum2(p, q);
        }
        const int s1 = (lane & 1) ? -1 : 1, s2 = (lane & 2) ? -1 : 1, s4 = (lane & 4) ? -1 : 1, s8 = (lane & 8) ? -1 : 1;
        int s = 0;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            int t = v[i];
            t = t * s4 + xor4(t);
            t = t * s1 + xor1(t);
            t = t * s2 + xor2(t);
            t = t * s8 + xor8(t);
            s += t < 0 ? -t : t;
        }
        return s;
    }

    // ------------------------------------------------------------------ out-of-line pixel functions
    // (Everything with per-lane control flow lives in these functions: the kernel body itself then
    // has only uniform branches and compiles to scalar code.  All of them are leaves, so none needs
    // a stack frame.)

    // cache the PU (encoder/motion.cpp:104-119): fill fenc / gxy from the encoder plane
    static __device__ __noinline__ void fill_pu(uint32_t base, uint64_t enc_addr, int enc_stride_)
    {
        View v(base);
        const int enc_stride = uni(enc_stride_), w = uni(v.s->w), h = uni(v.s->h);
        const PIX *e = (const PIX*)uni64(enc_addr);
        for (int g = v.tid; g < v.G; g += NT)
        {
            int x, y;
            group_xy(g, w, h, x, y);
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

    // copy the slice [CS_LO, CS_LO + CS_LEN) of the MV cost table into LDS
    static __device__ __noinline__ void stage_slices(uint32_t base)
    {
        View v(base);
        const uint16_t *cost = (const uint16_t*)uni64(v.s->cost);
        for (int i = v.tid; i < CS_LEN / 8; i += NT)         // 16-byte pieces
        {
            const uint4 d = *(const uint4*)(cost + CS_LO + 8 * i);
            XA_LDS uint32_t *q = &v.l->cs[4 * i];
            q[0] = d.x; q[1] = d.y; q[2] = d.z; q[3] = d.w;
        }
        bsync();
    }

    // one lane publishes the job's result and evaluation count (hist: optional per-job record of the count)
    static __device__ __noinline__ void publish(x265amd_me_result *out, int mvx, int mvy, int cost,
                                                unsigned long long *evals, unsigned n_evals, unsigned *hist)
    {
        if (threadIdx.x == 0)
        {
            x265amd_me_result r;
            r.mvx = (int16_t)mvx;
            r.mvy = (int16_t)mvy;
            r.cost = cost;
            *out = r;
            atomicAdd(evals, (unsigned long long)n_evals);
            if (hist)
                *hist = n_evals;            // the job's length, for the next run's launch order
        }
    }

    // (Re)stage the LDS reference window so that it holds every block (with the 8-tap apron) whose
    // origin lies in [minx,maxx] x [miny,maxy], with as much of WIN_MARGIN around as fits; the new
    // geometry goes to JobState.  Returns 0 (nothing changed) if even margin 0 does not fit.
    // (The previous window's readers are all past their reduction barrier.)
    static __device__ __noinline__ int restage(uint32_t base, int minx_, int miny_, int maxx_, int maxy_)
    {
        View v(base);
        const int minx = uni(minx_), miny = uni(miny_), maxx = uni(maxx_), maxy = uni(maxy_);
        const int nw = maxx - minx + uni(v.s->w) + 8, nh = maxy - miny + uni(v.s->h) + 7;
        auto bytes = [&](int mm) { return ((((nw + 2 * mm) * PB + 3) & ~3) + 4) * (nh + 2 * mm); };
        int m = WIN_MARGIN;
        while (m > 0 && bytes(m) > L::WIN_BYTES)
            m--;
        if (bytes(m) > L::WIN_BYTES)
        {
            bsync();            // (every cooperative function passes at least one barrier: see worker())
            return 0;
        }
        const int wx0 = minx - 3 - m, wy0 = miny - 3 - m, ww = nw + 2 * m, wh = nh + 2 * m;
        const int wpitch = ((ww * PB + 3) & ~3) + 4, pdw = wpitch >> 2;
        v.s->wx0 = wx0; v.s->wy0 = wy0; v.s->ww = ww; v.s->wh = wh; v.s->wpitch = wpitch;
        const int stride = uni(v.s->stride), gLo = uni(v.s->gLo), gHi = uni(v.s->gHi);
        const PIX *refbase = (const PIX*)uni64(v.s->refbase);
        const int n = pdw * wh;
        // every lane's loads are issued before the first is stored (one memory round trip per restage);
        // row = i / pdw by multiplication: exact for i * pdw < 2^32
        constexpr int IT = (L::WIN_BYTES / 4 + NT - 1) / NT;
        const uint32_t magic = 0xffffffffu / (uint32_t)pdw + 1;
        uint32_t d[IT > 0 ? IT : 1];
#pragma unroll
        for (int k = 0; k < IT; k++)
        {
            const int i = v.tid + k * NT;
            if (i < n)
            {
                const int r = (int)__umulhi((uint32_t)i, magic), c = i - r * pdw;
                int o = (wy0 + r) * stride + wx0 + c * (4 / PB);
                o = o < gLo ? gLo : o;
                o = o > gHi - 4 / PB ? gHi - 4 / PB : o;
                d[k] = ld_u32(refbase + o);
            }
        }
#pragma unroll
        for (int k = 0; k < IT; k++)
        {
            const int i = v.tid + k * NT;
            if (i < n)
                v.l->win[i] = d[k];
        }
        bsync();
        return 1;
    }

    template <int N, bool LW>
    static __device__ __forceinline__ void sad_loop(const View &v, const Src &src, int pitch, const int *p, int *acc)
    {
        for (int t = 0; t < v.T; t++)
        {
            const int g = t * NT + v.tid;
            if (g < v.G)
            {
                const int gv = v.l->gxy[g];
                const int lo = (gv >> 8) * pitch + (gv & 0xff) * (LW ? PB : 1);
                uint32_t f[2];
                f[0] = v.l->fenc[DW * g];
                if (DW == 2)
                    f[1] = v.l->fenc[DW * g + 1];
#pragma unroll
                for (int k = 0; k < N; k++)
                {
                    // (at most 4 candidates' loads in flight: keeps the function within 64 registers,
                    // i.e. 8 waves per SIMD)
                    if (k == 4)
                        __builtin_amdgcn_sched_barrier(0);
                    uint32_t r[2];
                    src.template rd4_packed<LW>(p[k] + lo, r);
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

    // SADs of n candidates of a PU of at most 16 groups: candidate c0 + row in 16-lane row `row`;
    // lane k of `res` receives the SAD of candidate k.
    template <bool LW>
    static __device__ __forceinline__ void sad_rows(const View &v, const Src &src, int pitch, int n, int cflags, int &res)
    {
        const int row = v.lane >> 4, gl = v.lane & 15;
        const bool in_pu = gl < v.G;
        const int gv = v.l->gxy[in_pu ? gl : 0];
        const int lo = (gv >> 8) * pitch + (gv & 0xff) * (LW ? PB : 1);
        uint32_t f[2];
        f[0] = v.l->fenc[DW * (in_pu ? gl : 0)];
        if (DW == 2)
            f[1] = v.l->fenc[DW * (in_pu ? gl : 0) + 1];
        for (int c0 = 0; c0 < n; c0 += 4)
        {
            const int k = c0 + row;
            const int off = v.s->pts[k < n ? k : 0];
            uint32_t r[2];
            src.template rd4_packed<LW>(off + lo, r);
            if (c0 == 0)
                res += label_cost(v, n, cflags);        // (its LDS look-ups run while the pixel reads are in flight)
            int sad;
            if (DW == 1)
                sad = __builtin_amdgcn_sad_u8(f[0], r[0], 0);
            else
                sad = __builtin_amdgcn_sad_u16(f[1], r[1], __builtin_amdgcn_sad_u16(f[0], r[0], 0));
            const int tot = sum16((in_pu && k < n) ? sad : 0);
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const int t = __builtin_amdgcn_readlane(tot, 16 * q);
                res += (v.lane == c0 + q) ? t : 0;
            }
        }
    }

    // 4 prediction pixels of one group for interpolation mode `mode` / phase (xf,yf); filters follow
    // common/ipfilter.cpp with hr = 14 - depth.  `off` addresses the group's first pixel in the
    // candidate block; PRED_HV reads the horizontally filtered intermediate from LDS (stage_hv) at
    // element `ioff`.  `mode` is uniform: the switch is a scalar branch.
    template <bool LW>
    static __device__ __forceinline__ P4 pred4(const View &v, const Src &src, int mode, int off, int pitch, int ioff, int w,
                                               int depth_rt, int xf, int yf)
    {
        constexpr int step = LW ? PB : 1;
        if (mode == PRED_INT)
        {
            uint32_t d[2];
            src.template rd4_packed<LW>(off, d);
            return unpack4(d);
        }
        const int depth = DW == 1 ? 8 : depth_rt;     // 8-bit planes: every shift and bound is a constant
        P4 r;
        if (mode == PRED_H)        // filterHorizontal_p_p, ipfilter.cpp:91-139
        {
            uint32_t px[6];
            src.template rd12_packed<LW>(off - 3 * step, px);
            h_pp(px, xf, depth, r.v);
            return r;
        }
        if (mode == PRED_V)        // filterVertical_p_p, ipfilter.cpp:354-425
        {
            uint32_t rw[8][2];
#pragma unroll
            for (int k = 0; k < 8; k++)
                src.template rd4_packed<LW>(off + (k - 3) * pitch, rw[k]);
            v_pp(rw, yf, depth, r.v);
            return r;
        }
        // PRED_HV, second half: filterVertical_s_p (ipfilter.cpp:38-89) over the LDS intermediate,
        // whose row k holds picture row (k - 3) of the candidate block
        {
            const XA_LDS int16_t *m = v.imm + ioff;
            uint32_t rw[8][2];
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const uint64_t d = *(const XA_LDS uint64_t*)(m + k * w);
                rw[k][0] = (uint32_t)d;
                rw[k][1] = (uint32_t)(d >> 32);
            }
            v_sp(rw, yf, depth, r.v);
            return r;
        }
    }
    // PRED_HV, first half: filterHorizontal_p_s (ipfilter.cpp:265-309) of rows -3 .. h+4 of the
    // candidate block, shared by all lanes through LDS (as the reference shares immedVal,
    // encoder/motion.cpp:1158-1161).  One row more than a candidate needs: the candidate one row
    // below with the same horizontal phase (the other diagonal neighbour of a refinement round)
    // then reuses the intermediate.
    template <bool LW>
    static __device__ __forceinline__ void stage_hv(const View &v, const Src &src, int off, int pitch, int w, int h, int depth_rt,
                                                    int xf)
    {
        constexpr int step = LW ? PB : 1;
        const int depth = DW == 1 ? 8 : depth_rt;
        const int ngx = w >> 2;
        const int n = (h + 8) * ngx;
        for (int idx = v.tid; idx < n; idx += NT)
        {
            const int row = idx / ngx, col = (idx - row * ngx) * 4;
            uint32_t px[6];
            src.template rd12_packed<LW>(off + (row - 3) * pitch + (col - 3) * step, px);
            int sm[4];
            h_ps(px, xf, depth, sm);
            const uint32_t lo = __builtin_amdgcn_perm(sm[1], sm[0], 0x05040100), hi = __builtin_amdgcn_perm(sm[3], sm[2], 0x05040100);
            *(XA_LDS uint64_t*)(v.imm + row * w + col) = (uint64_t)lo | ((uint64_t)hi << 32);
        }
        bsync();
    }

    // Block cost of kind `kind` (sad / satd / sa8d_inter) of the cached PU against the candidate
    // whose block origin is at `off` (window bytes if lw, else plane pixels) with fractional phase
    // (xf,yf): subpelCompare for a full-resolution reference (encoder/motion.cpp:1140-1165) when the
    // phase is non-zero.  One body for all kinds and interpolation modes (uniform switches) keeps
    // the instruction footprint small.
    struct HvCache { int off, xf; bool valid; };    // which horizontal intermediate LDS holds (uniform)

    // which accumulation a cost kind needs for this PU: 0 sad, 1 4x4 Hadamard (halved at the end),
    // 2 8x8 Hadamard rounded per 8x8, 3 8x8 Hadamard rounded per 16x16 (common/pixel.cpp:652-826)
    static __device__ __forceinline__ int accum_mode(int kind, int w, int h)
    {
        const bool m8 = ((w | h) & 7) == 0, m16 = ((w | h) & 15) == 0;
        return kind == X265AMD_SAD ? 0 : (kind == X265AMD_SATD || !m8) ? 1 : (m16 ? 3 : 2);
    }
    // add one pass's differences d[4] of this lane: a0 per-lane partials, a1 sums already rounded per
    // 16x16 (lane 63 only).  Executed by every lane (inactive ones with d = 0): the transforms cross lanes.
    static __device__ __forceinline__ void accum(int am, const int *d, int lane, int &a0, int &a1)
    {
        if (am == 0)
        {
#pragma unroll
            for (int i = 0; i < 4; i++)
                a0 += d[i] < 0 ? -d[i] : d[i];
        }
        else if (am == 1)
            a0 += had4_partial(d, lane);
        else
        {
            int sm = had8_partial(d, lane);
            if (am == 3)
            {
                int s64 = wave_sum_last(sm);                             // sa8d16: one rounding per 16x16 (= one wave pass)
                a1 += (lane == 63) ? ((s64 + 2) >> 2) : 0;
            }
            else
            {
                int s16 = sum16(sm);                                     // sa8d8: one rounding per 8x8
                a0 += ((lane & 15) == 0) ? ((s16 + 2) >> 2) : 0;
            }
        }
    }

    static __device__ __forceinline__ int cost_body(View &v, const Src &src, bool lw, int off, int pitch, int frac,
                                                    int kind, int par, HvCache &hc)
    {
        const int w = uni(v.s->w), h = uni(v.s->h), depth = v.s->depth;
        const int xf = frac & 3, yf = (frac >> 2) & 3;
        const int mode = (xf ? PRED_H : 0) | (yf ? PRED_V : 0);
        int hv_row = 0;             // first intermediate row of this candidate
        if (mode == PRED_HV)
        {
            const int uoff = uni(off), upitch = uni(pitch);
            if (hc.valid && hc.xf == xf && (uoff == hc.off || uoff == hc.off + upitch))
                hv_row = uoff != hc.off;
            else
            {
                if (lw) stage_hv<true>(v, src, off, pitch, w, h, depth, xf);
                else    stage_hv<false>(v, src, off, pitch, w, h, depth, xf);
                hc.off = uoff; hc.xf = xf; hc.valid = true;
            }
        }
        const int am = accum_mode(kind, w, h);
        int acc[2] = { 0, 0 };      // [0] per-lane partials, [1] sums already rounded per 16x16 (lane 63 only)
        for (int t = 0; t < v.T; t++)
        {
            const int g = t * NT + v.tid;
            int d[4] = { 0, 0, 0, 0 };
            if (g < v.G)
            {
                const int gv = v.l->gxy[g], gx = gv & 0xff, gy = gv >> 8;
                uint32_t fd[2];
                fd[0] = v.l->fenc[DW * g];
                if (DW == 2)
                    fd[1] = v.l->fenc[DW * g + 1];
                const P4 f = unpack4(fd);
                const int lo = off + gy * pitch + gx * (lw ? PB : 1);
                const P4 r = lw ? pred4<true>(v, src, mode, lo, pitch, (gy + hv_row) * w + gx, w, depth, xf, yf)
                                : pred4<false>(v, src, mode, lo, pitch, (gy + hv_row) * w + gx, w, depth, xf, yf);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    d[i] = f.v[i] - r.v[i];
            }
            accum(am, d, v.lane, acc[0], acc[1]);
        }
        v.template job_sum<2>(acc, par);
        int tot = acc[0];
        if (am == 1)
            tot >>= 1;                                                       // every 4x4 |H| sum is even
        return tot + acc[1];
    }

    // MV cost of label k (JobState::lbl), computed by lane k < n; 0 elsewhere or with CF_NOMV
    static __device__ __forceinline__ int label_cost(const View &v, int n, int cflags)
    {
        int mv = 0;
        if (!(cflags & CF_NOMV) && v.lane < n && !((cflags & CF_NOMV0) && v.lane == 0))
        {
            const int lb = v.s->lbl[v.lane];
            const int dx = (int16_t)((int16_t)lb - v.s->mvpx), dy = (int16_t)((lb >> 16) - v.s->mvpy);
            const XA_GLOBAL uint16_t *tab = (const XA_GLOBAL uint16_t*)(uintptr_t)v.s->cost;    // (global, not FLAT, loads)
            const unsigned ix = (unsigned)(dx - CS_LO), iy = (unsigned)(dy - CS_LO);
            const XA_LDS uint16_t *cs = (const XA_LDS uint16_t*)&v.l->cs[0];
            int a = (v.s->sliced && ix < (unsigned)CS_LEN) ? cs[ix] : tab[dx];
            int b = (v.s->sliced && iy < (unsigned)CS_LEN) ? cs[iy] : tab[dy];
            mv = (cflags & CF_NOTRUNC) ? a + b : ((a + b) & 0xffff);
        }
        return mv;
    }

    // The two evaluation entry points: costs of the n <= 8 candidates whose block-origin offsets are
    // in JobState::pts, each plus (unless CF_NOMV) the MV cost of its label in JobState::lbl
    // (COST_MV / COST_MV_X4, encoder/motion.cpp:173-249).  Entries k >= n of the result are undefined.
    //
    // eval_sad: full-pel SADs, all candidates in one pass over the PU, like the sad / sad_x3 /
    // sad_x4 primitives (common/pixel.cpp:78-142).
    // (out of line for the general kernels, whose body must keep to uniform control flow; the single-wavefront kernels
    // for half-resolution references are latency-bound chains and take the same code inline: call_sad / call_any)
    static __device__ __noinline__ Res8 eval_sad(uint32_t base, int n_, int cflags_) { return eval_sad_body(base, n_, cflags_); }
    static __device__ __noinline__ Res8 eval_any(uint32_t base, int n_, int cflags_, int kind_) { return eval_any_body(base, n_, cflags_, kind_); }
    static constexpr bool INLINE_EVAL = L::INL;
    static __device__ __forceinline__ Res8 call_sad(uint32_t base, int n, int cflags)
    {
        if constexpr (INLINE_EVAL) return eval_sad_body(base, n, cflags);
        else                       return eval_sad(base, n, cflags);
    }
    static __device__ __forceinline__ Res8 call_any(uint32_t base, int n, int cflags, int kind)
    {
        if constexpr (INLINE_EVAL) return eval_any_body(base, n, cflags, kind);
        else                       return eval_any(base, n, cflags, kind);
    }
    static __device__ __forceinline__ Res8 eval_sad_body(uint32_t base, int n_, int cflags_)
    {
        const int n = uni(n_), cflags = uni(cflags_);
        View v(base);
        const bool lw = L::USE_WIN && (cflags & CF_WINDOW) != 0;
        const int pitch = lw ? v.s->wpitch : v.s->stride;
        Src src(v, lw);
        const int par = (cflags & CF_PAR) ? 1 : 0;
        if (NW == 1 && n > 1 && v.G <= 16)
        {
            // small PU (8x8 and below) on its own wavefront: four candidates side by side, one per 16-lane row
            int res = 0;
            if (lw) sad_rows<true>(v, src, pitch, n, cflags, res);
            else    sad_rows<false>(v, src, pitch, n, cflags, res);
            Res8 r;
#pragma unroll
            for (int k = 0; k < 8; k++)
                r.c[k] = __builtin_amdgcn_readlane(res, k);
            return r;
        }
        const int mv = label_cost(v, n, cflags);
        int p[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
            p[k] = v.s->pts[k];             // (entries >= n repeat candidate 0)
        int acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        int t;                              // lane k: SAD of candidate k
        if (n == 1)
        {
            if (lw) sad_loop<1, true>(v, src, pitch, p, acc);
            else    sad_loop<1, false>(v, src, pitch, p, acc);
            t = v.template job_sum_lanes<1>(acc, par);
        }
        else if (n <= 4)
        {
            if (lw) sad_loop<4, true>(v, src, pitch, p, acc);
            else    sad_loop<4, false>(v, src, pitch, p, acc);
            t = v.template job_sum_lanes<4>(acc, par);
        }
        else
        {
            if (lw) sad_loop<8, true>(v, src, pitch, p, acc);
            else    sad_loop<8, false>(v, src, pitch, p, acc);
            t = v.template job_sum_lanes<8>(acc, par);
        }
        t += mv;
        Res8 r;
#pragma unroll
        for (int k = 0; k < 8; k++)
            r.c[k] = __builtin_amdgcn_readlane(t, k);
        return r;
    }

    // N candidates against a half-resolution reference set (phase words with LR_FLAG): every
    // candidate's plane reads are issued before the first cost is reduced, so a refinement round pays
    // the memory latency once, and ONE cross-wave reduction serves all candidates.  Lane k receives
    // the cost of candidate k in `res`.
    template <int N>
    static __device__ __forceinline__ void lr_batch(View &v, const Src &src, int kind, int par, int &res)
    {
        const int w = uni(v.s->w), h = uni(v.s->h), pitch = v.s->stride, step_lr = v.s->lrstep;
        const int am = accum_mode(kind, w, h);
        int offa[N], offb[N], adda[N], addb[N];
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            const int off = v.s->pts[k], fr = uni(v.s->frc[k]);
            const bool avg = (fr & 0x10) != 0;          // without averaging "B" is A again: (a + a + 1) >> 1 == a
            offa[k] = off;
            offb[k] = avg ? off + ((fr >> 5) & 1) + ((fr >> 6) & 1) * pitch : off;
            adda[k] = (fr & 3) * step_lr;
            addb[k] = (avg ? (fr >> 2) & 3 : fr & 3) * step_lr;
        }
        int acc[N];
#pragma unroll
        for (int k = 0; k < N; k++)
            acc[k] = 0;
        for (int t = 0; t < v.T; t++)
        {
            const int g = t * NT + v.tid;
            const bool active = g < v.G;
            const int gv = active ? v.l->gxy[g] : 0, lo = (gv >> 8) * pitch + (gv & 0xff);
            uint32_t fd[2] = { 0, 0 };
            if (active)
            {
                fd[0] = v.l->fenc[DW * g];
                if (DW == 2)
                    fd[1] = v.l->fenc[DW * g + 1];
            }
            const P4 f = unpack4(fd);
            uint32_t a[N][2], b[N][2];
#pragma unroll
            for (int k = 0; k < N; k++)
            {
                src.rd4_plane(offa[k] + lo, adda[k], a[k]);
                src.rd4_plane(offb[k] + lo, addb[k], b[k]);
            }
#pragma unroll
            for (int k = 0; k < N; k++)
            {
                P4 r;
                if (DW == 1)
                {
                    a[k][0] = __builtin_amdgcn_lerp(a[k][0], b[k][0], 0x01010101u);     // per byte (a + b + 1) >> 1
                    r = unpack4(a[k]);
                }
                else
                {
                    const P4 pa = unpack4(a[k]), pb = unpack4(b[k]);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        r.v[i] = (pa.v[i] + pb.v[i] + 1) >> 1;
                }
                int d[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                    d[i] = active ? f.v[i] - r.v[i] : 0;
                accum(am, d, v.lane, acc[k], acc[k]);
            }
        }
        const int t = v.template job_sum_lanes<N>(acc, par);
        res += v.lane < N ? (am == 1 ? t >> 1 : t) : 0;
    }

    // The same for PUs of at most 16 groups on a single wavefront (8x8 and smaller: the lookahead's
    // CUs): the PU occupies one 16-lane row, so four candidates are priced side by side, one per row,
    // instead of one after the other in the same lanes.
    static __device__ __forceinline__ void lr_batch_rows(View &v, const Src &src, int kind, int n, int cflags, int &res)
    {
        const int w = uni(v.s->w), h = uni(v.s->h), pitch = v.s->stride, step_lr = v.s->lrstep;
        const int am = accum_mode(kind, w, h);
        const int row = v.lane >> 4, gl = v.lane & 15;
        const bool in_pu = gl < v.G;
        const int gv = v.l->gxy[in_pu ? gl : 0], lo = (gv >> 8) * pitch + (gv & 0xff);
        uint32_t fd[2] = { 0, 0 };
        fd[0] = v.l->fenc[DW * (in_pu ? gl : 0)];
        if (DW == 2)
            fd[1] = v.l->fenc[DW * (in_pu ? gl : 0) + 1];
        const P4 f = unpack4(fd);
        for (int c0 = 0; c0 < n; c0 += 4)
        {
            const int k = c0 + row;
            const bool act = in_pu && k < n;
            const int kk = k < n ? k : 0;
            const int off = v.s->pts[kk], fr = v.s->frc[kk];
            const bool avg = (fr & 0x10) != 0;
            const int offb = avg ? off + ((fr >> 5) & 1) + ((fr >> 6) & 1) * pitch : off;
            uint32_t a[2], b[2];
            src.rd4_plane(off + lo, (fr & 3) * step_lr, a);
            src.rd4_plane(offb + lo, (avg ? (fr >> 2) & 3 : fr & 3) * step_lr, b);
            if (c0 == 0)
                res += label_cost(v, n, cflags);        // (its LDS look-ups run while the plane reads are in flight)
            P4 r;
            if (DW == 1)
            {
                a[0] = __builtin_amdgcn_lerp(a[0], b[0], 0x01010101u);
                r = unpack4(a);
            }
            else
            {
                const P4 pa = unpack4(a), pb = unpack4(b);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    r.v[i] = (pa.v[i] + pb.v[i] + 1) >> 1;
            }
            int d[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                d[i] = act ? f.v[i] - r.v[i] : 0;
#ifdef XA_PROFILE
            if (c0 == 0)
                v.s->cpt[2] = (long long)__builtin_readcyclecounter() + (__builtin_amdgcn_readfirstlane(d[0]) & 0);
#endif
            int a0 = 0;
            accum(am, d, v.lane, a0, a0);
            int tot = sum16(a0);                        // the row's (= the candidate's) total in every lane of the row
            if (am == 1)
                tot >>= 1;
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const int t = __builtin_amdgcn_readlane(tot, 16 * q);
                res += (v.lane == c0 + q) ? t : 0;
            }
        }
    }

    // Any cost kind / phase for PUs of at most 16 groups of 8-bit pixels on a single wavefront: four
    // candidates side by side, one per 16-lane row, all through ONE prediction path - the two-pass
    // (horizontal to 16-bit, then vertical) filter with the identity taps of phase 0 where a direction
    // is not fractional.  At 8-bit depth that path equals the reference's one-pass filters bit for bit:
    // the intermediate is 64 p - 8192 (or sum_h - 8192), no 16-bit cast bites, and
    // (64 S + 2048) >> 12 == (S + 32) >> 6 (common/ipfilter.cpp:38-139, 265-309, 354-425 with headRoom 6).
    static __device__ __forceinline__ int sel4(int f, int a, int b, int c, int d)
    {
        return f == 0 ? a : f == 1 ? b : f == 2 ? c : d;
    }
    // (candidates [k_lo, n): a candidate-parallel wavefront prices its share only)
    template <bool LW>
    // lbl_n > 0: also add the MV costs of the lbl_n labels to `res` (lane k: candidate k) - looked up while the first pixel reads
    // are in flight.
    static __device__ __forceinline__ void subpel_rows(View &v, const Src &src, int pitch, int kind, int n, int &res, int k_lo = 0,
                                                       int lbl_n = 0, int lbl_flags = 0)
    {
        static_assert(DW == 1, "8-bit pixels only");
        constexpr int step = LW ? PB : 1;
        const int w = uni(v.s->w), h = uni(v.s->h), ngx = w >> 2;
        const int per = (h + 7) * ngx;                  // intermediate groups (4 values each) per candidate
        const int lg = ngx == 1 ? 0 : ngx == 2 ? 1 : 2; // log2 of the groups per row (PU widths 4, 8, 16)
        const int am = accum_mode(kind, w, h);
        const int row = v.lane >> 4, gl = v.lane & 15;
        const bool in_pu = gl < v.G;
        const int gv = v.l->gxy[in_pu ? gl : 0], gx = gv & 0xff, gy = gv >> 8;
        uint32_t fd[2] = { v.l->fenc[in_pu ? gl : 0], 0 };
        const P4 f = unpack4(fd);
        for (int c0 = k_lo; c0 < n; c0 += 4)
        {
            // horizontal pass, one candidate after the other: rows -3 .. h+3, one 4-value group per lane
            // ((h + 7) * w / 4 <= 44 groups for every PU of at most 16 groups)
            const int lr = v.lane >> lg, col = (v.lane & (ngx - 1)) * 4;
            const int nq = n - c0 < 4 ? n - c0 : 4;
#pragma nounroll
            for (int q = 0; q < nq; q++)
            {
                const int kk = c0 + q;
                const int off = v.s->pts[kk], xf = v.s->frc[kk] & 3;
                if (v.lane < per)
                {
                    uint32_t px[6];
                    src.template rd12_packed<LW>(off + (lr - 3) * pitch + (col - 3) * step, px);
                    if (lbl_n > 0 && c0 == k_lo && q == 0)
                        res += label_cost(v, lbl_n, lbl_flags);
                    const int lo = sel4(xf, c_luma_b4[0][0], c_luma_b4[1][0], c_luma_b4[2][0], c_luma_b4[3][0]);
                    const int hi = sel4(xf, c_luma_b4[0][1], c_luma_b4[1][1], c_luma_b4[2][1], c_luma_b4[3][1]);
                    const uint32_t b0 = px[0] ^ 0x80808080u, b1 = px[1] ^ 0x80808080u, b2 = px[2] ^ 0x80808080u;
                    int sm[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        const uint32_t a = i ? __builtin_amdgcn_alignbyte(b1, b0, i) : b0;
                        const uint32_t b = i ? __builtin_amdgcn_alignbyte(b2, b1, i) : b1;
                        // (8192 from the -128 pixel bias and -8192 of the intermediate's offset cancel)
                        sm[i] = __builtin_amdgcn_sdot4((int)b, hi, __builtin_amdgcn_sdot4((int)a, lo, 0, false), false);
                    }
                    const uint32_t p01 = __builtin_amdgcn_perm(sm[1], sm[0], 0x05040100), p23 = __builtin_amdgcn_perm(sm[3], sm[2], 0x05040100);
                    *(XA_LDS uint64_t*)(v.imm + (q * per + v.lane) * 4) = (uint64_t)p01 | ((uint64_t)p23 << 32);
                }
            }
            bsync();
            // vertical pass + cost: candidate c0 + row in lane row `row`
            const int k = c0 + row, kk = k < n ? k : 0;
            const int yf = (v.s->frc[kk] >> 2) & 3;
            int tp[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                tp[i] = sel4(yf, c_luma_h2[0][i], c_luma_h2[1][i], c_luma_h2[2][i], c_luma_h2[3][i]);
            const XA_LDS int16_t *m = v.imm + (row * per + gy * ngx) * 4 + gx;
            int sum[4] = { 2048 + 524288, 2048 + 524288, 2048 + 524288, 2048 + 524288 };
#pragma unroll
            for (int t = 0; t < 8; t += 2)
            {
                const uint64_t d0 = *(const XA_LDS uint64_t*)(m + t * w);
                const uint64_t d1 = *(const XA_LDS uint64_t*)(m + (t + 1) * w);
                const uint32_t a[2] = { (uint32_t)d0, (uint32_t)(d0 >> 32) }, b[2] = { (uint32_t)d1, (uint32_t)(d1 >> 32) };
                vpair16(a, b, tp[t / 2], sum);
            }
            int d[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
                d[i] = (in_pu && k < n) ? f.v[i] - clamp_pix(sum[i] >> 12, 255) : 0;
            int a0 = 0;
            accum(am, d, v.lane, a0, a0);
            int tot = sum16(a0);
            if (am == 1)
                tot >>= 1;
#pragma unroll
            for (int q = 0; q < 4; q++)
            {
                const int t = __builtin_amdgcn_readlane(tot, 16 * q);
                res += (v.lane == c0 + q) ? t : 0;
            }
            bsync();                                    // the intermediate is reused by the next four
        }
    }

    // eval_any: any cost kind, each candidate with its own phase (JobState::frc), one cost_body per
    // candidate (half-resolution reference sets: one batch, lr_batch).
    static __device__ __forceinline__ Res8 eval_any_body(uint32_t base, int n_, int cflags_, int kind_)
    {
        const int n = uni(n_), cflags = uni(cflags_), kind = uni(kind_);
        View v(base);
        const bool lw = L::USE_WIN && (cflags & CF_WINDOW) != 0;
        const int pitch = lw ? v.s->wpitch : v.s->stride;
        Src src(v, lw);
#ifdef XA_PROFILE
        v.s->cpt[0] = (long long)__builtin_readcyclecounter();
#endif
        int par = (cflags & CF_PAR) ? 1 : 0;
        if (L::LR && NW == 1 && v.G <= 16 && (uni(v.s->frc[0]) & LR_FLAG))
        {
            int res = 0;                        // lane k collects candidate k
#ifdef XA_PROFILE
            v.s->cpt[1] = v.s->cpt[0];
#endif
            lr_batch_rows(v, src, kind, n, cflags, res);
            Res8 r;
#pragma unroll
            for (int k = 0; k < 8; k++)
                r.c[k] = __builtin_amdgcn_readlane(res, k);
#ifdef XA_PROFILE
            v.s->cpt[3] = (long long)__builtin_readcyclecounter() + (r.c[0] & 0);
#endif
            return r;
        }
        int res = label_cost(v, n, cflags);     // lane k collects candidate k
        if (L::LR && (uni(v.s->frc[0]) & LR_FLAG))
        {
#ifdef XA_PROFILE
            v.s->cpt[1] = (long long)__builtin_readcyclecounter() + (__builtin_amdgcn_readfirstlane(res) & 0);
#endif
            if (n == 1) lr_batch<1>(v, src, kind, par, res);
            else if (n <= 4) lr_batch<4>(v, src, kind, par, res);
            else             lr_batch<8>(v, src, kind, par, res);
            Res8 r;
#pragma unroll
            for (int k = 0; k < 8; k++)
                r.c[k] = __builtin_amdgcn_readlane(res, k);
#ifdef XA_PROFILE
            v.s->cpt[3] = (long long)__builtin_readcyclecounter() + (r.c[0] & 0);
#endif
            return r;
        }
        if constexpr (NW == 1 && DW == 1)
        {
            const int pw = uni(v.s->w);
            if (n > 1 && v.G <= 16 && (pw & (pw - 1)) == 0)     // (widths 4, 8, 16: a power-of-two number of groups per row)
            {
                if (lw) subpel_rows<true>(v, src, pitch, kind, n, res);
                else    subpel_rows<false>(v, src, pitch, kind, n, res);
                Res8 r;
#pragma unroll
                for (int k = 0; k < 8; k++)
                    r.c[k] = __builtin_amdgcn_readlane(res, k);
                return r;
            }
        }
        HvCache hc = { 0, 0, false };
        for (int k = 0; k < n; k++)
        {
            const int off = v.s->pts[k], frac = uni(v.s->frc[k]);
            const int c = cost_body(v, src, lw, off, pitch, frac, kind, par, hc);
            res += (v.lane == k) ? c : 0;
            par ^= 1;
        }
        Res8 r;
#pragma unroll
        for (int k = 0; k < 8; k++)
            r.c[k] = __builtin_amdgcn_readlane(res, k);
        return r;
    }

    // ---------------------------------------------------------------- candidate-parallel rounds (L::CP)
    // The n <= 8 candidates of JobState::pts / frc / lbl, read from the plane, dealt to the wavefronts in runs of ceil(n / 4):
    // of a refinement round's eight, wavefront i prices candidates 2 i and 2 i + 1 - the two vertical, the two horizontal and the
    // two pairs of diagonal neighbours, and a pair of diagonals shares its horizontal intermediate.  Cost + MV cost go to cp_res.
    static __device__ __forceinline__ void cp_part(uint32_t base, int wave)
    {
        View v(base);
        if (wave > 0)
            v.imm = v.l->cp_imm[wave - 1];
        const int n = uni(v.l->cp_cmd[1]), cflags = uni(v.l->cp_cmd[2]), kind = uni(v.l->cp_cmd[3]);
        const int nw = (int)(blockDim.x >> 6);          // wavefronts of the workgroup (the launch's choice, at most CP_WAVES)
        const int per = (n + nw - 1) / nw;
        const int k_lo = wave * per, k_hi = k_lo + per < n ? k_lo + per : n;
        if (k_lo >= k_hi)
            return;
        const int pitch = uni(v.s->stride);
        Src src(v, false);
        int res = 0;                                    // lane k: cost + MV cost of candidate k
        const int pw = uni(v.s->w);
        if (DW == 1 && v.G <= 16 && (pw & (pw - 1)) == 0)
        {
            if constexpr (DW == 1)
                subpel_rows<false>(v, src, pitch, kind, k_hi, res, k_lo, n, cflags);
        }
        else
        {
            res = label_cost(v, n, cflags);
            HvCache hc = { 0, 0, false };
            for (int k = k_lo; k < k_hi; k++)
            {
                const int off = v.s->pts[k], frac = uni(v.s->frc[k]);
                const int c = cost_body(v, src, false, off, pitch, frac, kind, 0, hc);
                res += (v.lane == k) ? c : 0;
            }
        }
        for (int k = k_lo; k < k_hi; k++)
        {
            const int c = __builtin_amdgcn_readlane(res, k);
            if (v.lane == 0)
                v.l->cp_res[k] = c;
        }
    }
    // leader: run one round (the candidates are in the job state)
    static __device__ __forceinline__ Res8 cp_round(uint32_t base, int n, int cflags, int kind)
    {
        LP l = lds(base);
        l->cp_cmd[0] = 1; l->cp_cmd[1] = n; l->cp_cmd[2] = cflags; l->cp_cmd[3] = kind;
        __syncthreads();
        cp_part(base, 0);
        __syncthreads();
        Res8 r;
#pragma unroll
        for (int k = 0; k < 8; k++)
            r.c[k] = l->cp_res[k];
        return r;
    }
    static __device__ __forceinline__ void cp_exit(uint32_t base)
    {
        if (L::CP)
        {
            lds(base)->cp_cmd[0] = 0;
            __syncthreads();
        }
    }
    // wavefronts 1 .. CP_WAVES - 1: parked at the barrier between rounds
    static __device__ __forceinline__ void cp_worker(uint32_t base, int wave)
    {
        LP l = lds(base);
        for (;;)
        {
            __syncthreads();
            if (uni(l->cp_cmd[0]) == 0)
                break;
            cp_part(base, wave);
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------- leader / worker protocol (NW > 1)
    // The search logic runs on the job's first wave only.  The other waves wait at a barrier for
    // a command {op, arguments} in JobLds::cmd, run the same cooperative function as the leader
    // (which contains the same barriers), and return to the barrier.  The leader posts a command
    // only after the previous function's last barrier, i.e. when every worker has read the
    // previous one.
    enum { OP_EXIT = 0, OP_FILL, OP_SLICES, OP_RESTAGE, OP_SAD, OP_ANY };

    static __device__ __forceinline__ void post(uint32_t base, int op, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0)
    {
        if (NW > 1)
        {
            LP l = lds(base);
            l->cmd[0] = op; l->cmd[1] = a0; l->cmd[2] = a1; l->cmd[3] = a2; l->cmd[4] = a3;
            __syncthreads();
        }
    }

    static __device__ __forceinline__ void worker(uint32_t base)
    {
        LP l = lds(base);
        for (;;)
        {
            __syncthreads();
            const int op = uni(l->cmd[0]);
            const int a0 = uni(l->cmd[1]), a1 = uni(l->cmd[2]), a2 = uni(l->cmd[3]), a3 = uni(l->cmd[4]);
            if (op == OP_EXIT)
                break;
            if (op == OP_SAD)
                eval_sad(base, a0, a1);
            else if (op == OP_ANY)
                eval_any(base, a0, a1, a2);
            else if (op == OP_RESTAGE)
                restage(base, a0, a1, a2, a3);
            else if (op == OP_SLICES)
                stage_slices(base);
            else
                fill_pu(base, ((uint64_t)(uint32_t)a1 << 32) | (uint32_t)a0, a2);
        }
    }
};

// ------------------------------------------------------------------ caller side

// Result of the dedicated exhaustive-search kernel for one job or one slab of candidate rows of
// a job: the minimum of (cost, raster index) over the candidates it covered.
struct FullBest
{
    int32_t cost;       // SAD + mvcost of the best raster candidate; INT32_MAX if none
    int16_t mvx, mvy;   // full-pel
};

// Winner over the n (<= 16) slab partials of one job: lowest cost, then lowest raster index, as a serial raster scan
// with strict `<` keeps it.  Lane s reads slot s; out of line, so that the kernel body keeps only uniform control flow.
// Returns the cost in lane 0's x and the MV word (mvy << 16 | mvx & 0xffff) in y; cost INT32_MAX: no candidate.
struct SlabBest { int cost, mv; };
static __device__ __noinline__ SlabBest combine_slabs(const FullBest *fb, int n_, int nxr_, int minx_, int miny_)
{
    const int n = uni(n_), nxr = uni(nxr_), minx = uni(minx_), miny = uni(miny_);
    const int lane = (int)(threadIdx.x & 63);
    unsigned long long key = ~0ull;
    int mv = 0;
    if (lane < n)
    {
        const FullBest f = fb[lane];
        mv = ((int)f.mvy << 16) | ((int)f.mvx & 0xffff);
        if (f.cost != 0x7fffffff)
            key = ((unsigned long long)(uint32_t)f.cost << 32) | (uint32_t)((f.mvy - miny) * nxr + (f.mvx - minx));
    }
    unsigned long long m = key;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1)
    {
        const unsigned long long o = __shfl_xor(m, d);
        m = o < m ? o : m;
    }
    m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    SlabBest r;
    r.cost = 0x7fffffff;
    r.mv = 0;
    if (m != ~0ull)
    {
        const unsigned long long hit = __ballot(key == m);
        r.cost = (int)(m >> 32);
        r.mv = __shfl(mv, __builtin_ctzll(hit));
    }
    return r;
}

// the 8 neighbours of a refinement round / the square patterns (encoder/motion.cpp:124); entry 0 = the centre
// (32-bit entries: uniform lookups then compile to scalar loads, which have no byte form on this target)
static __constant__ int c_square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };

// What the (scalar, inlined) search logic holds of a job: the LDS offset, the PU size, the MV-cost
// context and the geometry of the LDS reference window - all in scalar registers.
template <class L>
struct Wave
{
    typedef typename L::PIX PIX;
    typedef Dev<L> D;
    static constexpr int NW = L::NW, NT = L::NT, DW = L::DW, PB = (int)sizeof(PIX);

    uint32_t base;
    typename D::LP l;
    typename D::SP s;
    int w, h, stride, mvpx, mvpy, ikind, jflags;
    bool sliced;                    // the LDS cost slice is staged
    bool lowres;                    // the reference is a half-resolution set (ReferencePlanes::isLowres)
    const uint16_t *cost;
    int wx0, wy0, ww, wh, wpitch;   // reference window: origin relative to the PU origin, size (pixels), pitch (bytes)
    bool wvalid;
    int par;                        // parity of the next cross-wave reduction slot (CF_PAR or 0)
    unsigned evals;
#ifdef XA_PROFILE                   // developer aid: shader-clock stamps at phase boundaries of the search
    long long tstamp[8];
    long long cgen[6] = { 0, 0, 0, 0, 0, 0 };          // eval_n: SAD calls {count, set-up, call}, other calls {count, set-up, call}
    long long cprof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // eval_any calls: count, set-up, call -> entry, MV costs, pixel work, reduce, return, unpack
    __device__ __forceinline__ void stamp(int k) { tstamp[k] = (long long)__builtin_readcyclecounter(); }
#else
    __device__ __forceinline__ void stamp(int) {}
#endif

    // bind the PU: fill fenc / gxy / JobState.  (enc_x,enc_y) / (ref_x,ref_y): PU origin in the two planes.
    __device__ __forceinline__ void bind(L &lds_obj, const PlaneDesc &enc, const PlaneDesc &ref, int enc_x, int enc_y,
                                         int ref_x, int ref_y, int w_, int h_, const uint16_t *cost_, int flags,
                                         bool keep_slices = false)
    {
        l = (typename D::LP)&lds_obj;
        base = (uint32_t)(uintptr_t)l;
        s = &l->st;
        w = w_; h = h_;
        stride = ref.stride;
        cost = cost_;
        ikind = X265AMD_SAD;
        jflags = flags;
        mvpx = mvpy = 0;
        // (keep_slices: the same context was bound before with the same cost table, whose LDS slice -
        // independent of the predictor - is still valid)
        if (!keep_slices)
            sliced = false;
        wvalid = false;
        wx0 = wy0 = ww = wh = wpitch = 0;
        par = 0;
        evals = 0;
        const int G = (w * h) >> 2;
        const int64_t blockOffset = (int64_t)ref_y * ref.stride + ref_x;
        s->refbase = (uint64_t)((const PIX*)ref.origin + blockOffset);
        s->cost = (uint64_t)cost_;
        s->gLo = (int32_t)(ref.lo - blockOffset);
        s->gHi = (int32_t)(ref.hi - blockOffset);
        s->stride = ref.stride; s->depth = ref.depth;
        s->w = w; s->h = h; s->G = G; s->T = (G + NT - 1) / NT;
        s->mvpx = 0; s->mvpy = 0;
        s->sliced = sliced ? 1 : 0;
        lowres = L::LR && ref.lr_step != 0;
        s->lrstep = (int32_t)ref.lr_step;
        const PIX *e = (const PIX*)enc.origin + (int64_t)enc_y * enc.stride + enc_x;
        D::post(base, D::OP_FILL, (int)(uint32_t)(uint64_t)e, (int)((uint64_t)e >> 32), enc.stride);
        D::fill_pu(base, (uint64_t)e, enc.stride);
    }

    // MV cost table entry for MVD `mvd`: from the LDS slice when inside it
    __device__ __forceinline__ int cost_entry(int mvd) const
    {
        unsigned i = (unsigned)(mvd - CS_LO);
        if (sliced && i < (unsigned)CS_LEN)
            return uni(((const XA_LDS uint16_t*)&l->cs[0])[i]);
        return uni(((const XA_GLOBAL uint16_t*)(uintptr_t)cost)[(int16_t)mvd]);
    }
    // mvcost(MV(x,y)) for q-pel x,y: uint16 sum of two table entries (encoder/bitcost.h:46)
    __device__ __forceinline__ int mvcost(int x, int y) const
    {
        int a = cost_entry((int16_t)(x - mvpx));
        int b = cost_entry((int16_t)(y - mvpy));
        return (a + b) & 0xffff;
    }

    // setMVP (encoder/bitcost.h:43) + keep the part of the MV-cost table around MVD 0 in LDS: the
    // search mostly prices MVs near the predictor (anything else is looked up in the table)
    __device__ __forceinline__ void set_mvp(int mvpx_, int mvpy_)
    {
        mvpx = mvpx_;
        mvpy = mvpy_;
        s->mvpx = mvpx; s->mvpy = mvpy;
        if ((jflags & JF_NO_SLICES) || sliced)
            return;
        sliced = true;
        s->sliced = 1;
        D::post(base, D::OP_SLICES);
        D::stage_slices(base);
    }

    // ---------------------------------------------------------------- LDS reference window policy
    // Decide where the blocks with origins in [minx,maxx] x [miny,maxy] (with the 8-tap apron) are
    // read from: the window if it holds them; else, for a compact point set, a window re-staged
    // around them (unless the caller expects no follow-up near them); else the plane.
    // Returns CF_WINDOW or 0.
    __device__ __forceinline__ int place(int minx, int miny, int maxx, int maxy, bool may_restage)
    {
        if (!L::USE_WIN || (jflags & JF_NO_WINDOW))
            return 0;
        if (wvalid && minx - 3 >= wx0 && maxx + w + 5 <= wx0 + ww && miny - 3 >= wy0 && maxy + h + 4 <= wy0 + wh)
            return CF_WINDOW;
        if (!may_restage || maxx - minx > WIN_LOCAL || maxy - miny > WIN_LOCAL)
            return 0;
        D::post(base, D::OP_RESTAGE, minx, miny, maxx, maxy);
        if (!uni(D::restage(base, minx, miny, maxx, maxy)))
            return 0;
        wx0 = uni(s->wx0); wy0 = uni(s->wy0); ww = uni(s->ww); wh = uni(s->wh); wpitch = uni(s->wpitch);
        wvalid = true;
        return CF_WINDOW;
    }
    // offset of the block with origin (cx,cy) in the addressing `where` selected, and that addressing's pitch
    __device__ __forceinline__ int block_off(int where, int cx, int cy) const
    {
        return where ? (cy - wy0) * wpitch + (cx - wx0) * PB : cy * stride + cx;
    }
    // reduction slot parity for a call that performs `count` reductions
    __device__ __forceinline__ int take_par(int count)
    {
        const int p = par;
        if (NW > 1 && (count & 1))
            par ^= CF_PAR;
        return p;
    }

    // Costs of the first n of N candidate blocks with full-pel origins (sx,sy)[k], of kind `kind`
    // and phases frac[k] (NULL: all full-pel), plus (unless CF_NOMV) the MV costs of the q-pel
    // labels (qx,qy)[k].
    template <int N>
    __device__ __forceinline__ void eval_n(int n, const int *sx, const int *sy, const int *qx, const int *qy, int kind,
                                           const int *frac, int cflags, int *out)
    {
        // bounding box of the origins (entries >= n repeat candidate 0, also as padding for the callee)
        int minx = sx[0], maxx = sx[0], miny = sy[0], maxy = sy[0];
#pragma unroll
        for (int k = 1; k < N; k++)
        {
            const int x = k < n ? sx[k] : sx[0], y = k < n ? sy[k] : sy[0];
            minx = x < minx ? x : minx; maxx = x > maxx ? x : maxx;
            miny = y < miny ? y : miny; maxy = y > maxy ? y : maxy;
        }
#ifdef XA_PROFILE
        const long long t_a = (long long)__builtin_readcyclecounter();
#endif
        const bool multi = kind == X265AMD_SAD && !frac;
        // a lone full-pel SAD (predictor checks, clipped pattern points) does not move the window;
        // phases of a half-resolution set select other planes than the one the window caches
        const int where = (frac && lowres) ? 0 : place(minx, miny, maxx, maxy, !(N == 1 && multi));
        constexpr int NP = N == 1 ? 1 : N <= 4 ? 4 : 8;     // candidates the SAD pass measures
#pragma unroll
        for (int k = 0; k < NP; k++)
        {
            const bool real = k < N && k < n;
            s->pts[k] = block_off(where, real ? sx[k < N ? k : 0] : sx[0], real ? sy[k < N ? k : 0] : sy[0]);
            if (frac)
                s->frc[k] = real ? frac[k < N ? k : 0] : frac[0];       // (entries >= n repeat candidate 0, like pts)
            if (k < N)
                s->lbl[k] = (qy[k] << 16) | (qx[k] & 0xffff);
        }
        evals += (unsigned)n;
        Res8 r;
        if (multi)
        {
            const int cf = where | cflags | take_par(1);
#ifdef XA_PROFILE
            const long long t_b = (long long)__builtin_readcyclecounter();
#endif
            D::post(base, D::OP_SAD, n, cf);
            r = D::call_sad(base, n, cf);
#ifdef XA_PROFILE
            const long long t_c = (long long)__builtin_readcyclecounter() + (uni(r.c[0]) & 0);
            cgen[0] += 1; cgen[1] += t_b - t_a; cgen[2] += t_c - t_b;
#endif
        }
        else
        {
            // (a half-resolution batch reduces all its candidates at once)
            const int cf = where | cflags | take_par((frac && lowres) ? 1 : n);
            if (!frac)
            {
#pragma unroll
                for (int k = 0; k < N; k++)
                    s->frc[k] = 0;
            }
            D::post(base, D::OP_ANY, n, cf, kind);
#ifdef XA_PROFILE
            const long long t_b = (long long)__builtin_readcyclecounter();
#endif
            bool dealt = false;
            if constexpr (L::CP && N > 1)
            {
                if (n >= 2 && !where && !lowres)        // several candidates from the plane: side by side on the workgroup's wavefronts
                {
                    r = D::cp_round(base, n, cf, kind);
                    dealt = true;
                }
            }
            if (!dealt)
                r = D::call_any(base, n, cf, kind);
#ifdef XA_PROFILE
            {
                const long long t_c = (long long)__builtin_readcyclecounter() + (uni(r.c[0]) & 0);
                cgen[3] += 1; cgen[4] += t_b - t_a; cgen[5] += t_c - t_b;
            }
            if (frac && lowres)
            {
                const long long t_c = (long long)__builtin_readcyclecounter() + (uni(r.c[0]) & 0);
#pragma unroll
                for (int k = 0; k < N; k++)
                    out[k] = uni(r.c[k]);
                const long long t_d = (long long)__builtin_readcyclecounter() + (out[0] & 0);
                const long long e0 = s->cpt[0], e1 = s->cpt[1], e2 = s->cpt[2], e3 = s->cpt[3];
                cprof[0] += 1; cprof[1] += t_b - t_a; cprof[2] += e0 - t_b; cprof[3] += e1 - e0; cprof[4] += e2 - e1;
                cprof[5] += e3 - e2; cprof[6] += t_c - e3; cprof[7] += t_d - t_c;
            }
#endif
        }
#pragma unroll
        for (int k = 0; k < N; k++)
            out[k] = uni(r.c[k]);
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
        eval_n<N>(N, sx, sy, qx, qy, ikind, nullptr, pflags, out);
    }

    // COST_MV's value: cost at full-pel (mx,my) + mvcost(MV(mx,my) << 2)   (motion.cpp:173-179)
    __device__ __forceinline__ int cost_fpel(int mx, int my)
    {
        int c;
        eval_points<1>(&mx, &my, &mx, &my, &c);
        return c;
    }

    // phase word of q-pel position (qx,qy) (see LR_FLAG)
    __device__ __forceinline__ int phase(int qx, int qy) const
    {
        if (!lowres)
            return (qx & 3) | ((qy & 3) << 2);
        // motion.cpp:1117-1122: plane A from the half-pel bits; with an odd component the mean of A and
        // the plane B of the position rounded away (qmvB = qmv + (qx & 1, qy & 1) * 2)
        const int bx = (int16_t)(qx + (qx & 1) * 2), by = (int16_t)(qy + (qy & 1) * 2);
        const int ha = (qy & 2) | ((qx & 2) >> 1), hb = (by & 2) | ((bx & 2) >> 1);
        return LR_FLAG | ha | (hb << 2) | (((qx | qy) & 1) << 4) | (((bx >> 2) - (qx >> 2)) << 5) | (((by >> 2) - (qy >> 2)) << 6);
    }

    // subpelCompare (motion.cpp:1110-1166) at q-pel (qx,qy); with_mv: plus mvcost(MV(qx,qy))
    __device__ __forceinline__ int subpel_cost(int qx, int qy, int kind, bool with_mv)
    {
        const int cx = qx >> 2, cy = qy >> 2, fr = phase(qx, qy);
        int c;
        eval_n<1>(1, &cx, &cy, &qx, &qy, kind, &fr, with_mv ? 0 : CF_NOMV, &c);
        return c;
    }
    // two positions without their MV costs (the template costs of two predictor candidates): side by side on two wavefronts of a
    // candidate-parallel workgroup, else one after the other
    __device__ __forceinline__ void subpel_cost2(const int *qx, const int *qy, int kind, int *out)
    {
        if constexpr (L::CP)
        {
            if (!lowres)
            {
#pragma unroll
                for (int k = 0; k < 2; k++)
                {
                    s->pts[k] = block_off(0, qx[k] >> 2, qy[k] >> 2);
                    s->lbl[k] = (qy[k] << 16) | (qx[k] & 0xffff);
                    s->frc[k] = phase(qx[k], qy[k]);
                }
                evals += 2;
                const Res8 r = D::cp_round(base, 2, CF_NOMV, kind);
                out[0] = uni(r.c[0]);
                out[1] = uni(r.c[1]);
                return;
            }
        }
        out[0] = subpel_cost(qx[0], qy[0], kind, false);
        out[1] = subpel_cost(qx[1], qy[1], kind, false);
    }
    // the same for the first n of 8 q-pel positions at once, each plus its MV cost: one refinement
    // round (motion.cpp:801-845) in a single call
    __device__ __forceinline__ void subpel_cost8(int n, const int *qx, const int *qy, int kind, int *out, int cflags = 0)
    {
        int cx[8], cy[8], fr[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            cx[k] = qx[k] >> 2; cy[k] = qy[k] >> 2;
            fr[k] = phase(qx[k], qy[k]);
        }
        eval_n<8>(n, cx, cy, qx, qy, kind, fr, cflags, out);
    }

    // entry [i][c] of one of the search's small tables (me_search.h): uniform indices, scalar loads from constant memory
    template <int TAB> static __device__ __forceinline__ int tab(int i, int c);

    // One refinement round (motion.cpp:799-809, 813-845, 851-861): the costs of the first n of the 8 neighbours of
    // (bmx, bmy) at distance `step` (q-pel units), each subpelCompare + mvcost, in c[0 .. n).  Candidate data goes
    // straight into the job state - no per-candidate arrays are kept in scalar registers.
    __device__ __forceinline__ void round_costs(int bmx, int bmy, int step, int n, int kind, int *c)
    {
        // every candidate's block origin lies within [(bmv - step) >> 2, (bmv + step) >> 2]
        const int where = lowres ? 0 : place((int)(int16_t)(bmx - step) >> 2, (int)(int16_t)(bmy - step) >> 2,
                                             (int)(int16_t)(bmx + step) >> 2, (int)(int16_t)(bmy + step) >> 2, true);
        // (only the first n entries: every evaluation path reads candidates below n, or repeats candidate 0)
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            if (k >= 4 && n <= 4)
                break;
            const int qx = (int)(int16_t)(bmx + c_square1[k + 1][0] * step), qy = (int)(int16_t)(bmy + c_square1[k + 1][1] * step);
            s->pts[k] = block_off(where, qx >> 2, qy >> 2);
            s->lbl[k] = (qy << 16) | (qx & 0xffff);
            s->frc[k] = phase(qx, qy);
        }
        evals += (unsigned)n;
        const int cf = where | take_par(lowres ? 1 : n);
        if constexpr (L::CP)
        {
            if (!where && !lowres && n >= 4)
            {
                const Res8 r = D::cp_round(base, n, cf, kind);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    c[k] = uni(r.c[k]);
                return;
            }
        }
        D::post(base, D::OP_ANY, n, cf, kind);
        const Res8 r = D::call_any(base, n, cf, kind);
#pragma unroll
        for (int k = 0; k < 8; k++)
            c[k] = uni(r.c[k]);
    }

    // winner over the n (<= 16) slab partials of a job (combine_slabs), wave-uniform
    __device__ __forceinline__ SlabBest slab_best(const FullBest *fb, int n, int nxr, int minx, int miny) const
    {
        const SlabBest sb = combine_slabs(fb, n, nxr, minx, miny);
        SlabBest r;
        r.cost = uni(sb.cost);
        r.mv = uni(sb.mv);
        return r;
    }

    // plain block cost against the block at the reference origin
    __device__ __forceinline__ int cost_direct(int kind)
    {
        const int z = 0;
        int c;
        eval_n<1>(1, &z, &z, &z, &z, kind, &z, CF_NOMV, &c);
        return c;
    }
};

} // namespace xa
