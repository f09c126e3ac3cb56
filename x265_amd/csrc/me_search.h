// MotionEstimate::motionEstimate as wave-uniform control flow over the wave-cooperative
// cost functions of me_device.h.
//
// Reference: encoder/motion.cpp:277-866 (search), :868-1108 (StarPatternSearch),
// :1110-1191 (subpelCompare / subpelInterpolate).  Every comparison is a strict `<` in the
// reference's program order (common/common.h:71-92), so ties resolve identically.  The
// reference's packed direction tags in DIA / HEX (motion.cpp:349-365, 393-424) are
// order-equivalent to tracking the index of the last strict improvement.
#pragma once

#include "me_device.h"

namespace xa {

// The search's small tables (encoder/motion.cpp:51-60, 122-143; Lib/TLibEncoder/TEncSearch.cpp:59-83).  Two images of each:
// a __constant__ array for the wave-uniform kernels (32-bit entries: uniform lookups then compile to scalar loads, which
// have no byte form on this target) and a constexpr copy from which the per-lane kernel (me_lane.h) packs 4-bit
// immediates - a per-lane table lookup would be a memory round trip in the middle of a dependent chain.
// (c_square1 lives in me_device.h.)
#define XA_T_HEX2 { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } }
#define XA_T_HEX4 { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 }, \
                    { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } }
#define XA_T_STAR { { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 }, \
                    { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 } }
// workload[] (motion.cpp:51-60): hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd
#define XA_T_WORKLOAD { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 }, \
                        { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } }
// s_mvRefineHpel / s_mvRefineQPel (Lib/TLibEncoder/TEncSearch.cpp:59-83): the two 9-point orders differ
#define XA_T_REFINE_H { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } }
#define XA_T_REFINE_Q { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } }
#define XA_T_RANGE_MUL { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } }
static __constant__ int c_hex2[8][2] = XA_T_HEX2;
static __constant__ int c_hex4[16][2] = XA_T_HEX4;
static __constant__ int c_star_offsets[16][2] = XA_T_STAR;
static __constant__ int c_workload[8][5] = XA_T_WORKLOAD;
static __constant__ int c_refine_hpel[9][2] = XA_T_REFINE_H;
static __constant__ int c_refine_qpel[9][2] = XA_T_REFINE_Q;
static __constant__ int c_range_mul[4][4] = XA_T_RANGE_MUL;
enum { T_SQUARE1, T_HEX2, T_HEX4, T_STAR, T_WORKLOAD, T_REFINE_H, T_REFINE_Q, T_RANGE_MUL };

template <class L> template <int TAB>
__device__ __forceinline__ int Wave<L>::tab(int i, int c)
{
    return TAB == T_SQUARE1 ? c_square1[i][c] : TAB == T_HEX2 ? c_hex2[i][c] : TAB == T_HEX4 ? c_hex4[i][c] :
           TAB == T_STAR ? c_star_offsets[i][c] : TAB == T_WORKLOAD ? c_workload[i][c] : TAB == T_REFINE_H ? c_refine_hpel[i][c] :
           TAB == T_REFINE_Q ? c_refine_qpel[i][c] : c_range_mul[i][c];
}

#define XA_S16(v) ((int)(int16_t)(v))

// A job unpacked into scalars.  (A by-value copy of the packed struct would be indexed dynamically
// through mvc[] and therefore live in per-lane scratch memory; these fields stay in scalar registers.)
struct JobRegs
{
    int pu_x, pu_y, w, h, mvmin_x, mvmin_y, mvmax_x, mvmax_y, qmvp_x, qmvp_y;
    int merange, method, subme, qp, numc, ref_idx, icost;
    int mvc[X265AMD_MAX_MVC][2];

    __device__ __forceinline__ void load(const x265amd_me_job *p)
    {
        static_assert(sizeof(x265amd_me_job) == 48 && X265AMD_MAX_MVC == 4, "job layout");
        const uint32_t *q = (const uint32_t*)p;
        uint32_t d[12];
#pragma unroll
        for (int i = 0; i < 12; i++)
            d[i] = q[i];
        auto lo = [](uint32_t v) { return (int)(int16_t)(v & 0xffff); };
        auto hi = [](uint32_t v) { return (int)(int16_t)(v >> 16); };
        pu_x = (int)d[0]; pu_y = (int)d[1];
        w = lo(d[2]); h = hi(d[2]);
        mvmin_x = lo(d[3]); mvmin_y = hi(d[3]);
        mvmax_x = lo(d[4]); mvmax_y = hi(d[4]);
        qmvp_x = lo(d[5]); qmvp_y = hi(d[5]);
        merange = lo(d[6]); method = (d[6] >> 16) & 0xff; subme = d[6] >> 24;
        qp = d[7] & 0xff; numc = (d[7] >> 8) & 0xff; ref_idx = (d[7] >> 16) & 0xff; icost = d[7] >> 24;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            mvc[i][0] = lo(d[8 + i]);
            mvc[i][1] = hi(d[8 + i]);
        }
    }
};

// WV: what prices candidates - Wave<L> (one PU per workgroup, pixels spread over the lanes; every value here is then
// wave-uniform) or LaneWave<> of me_lane.h (one PU per lane; every value here is then per lane).
template <class L, class WV = Wave<L>>
struct Search
{
    WV &W;
    const JobRegs &j;
    int bmx, bmy, bcost;

    __device__ __forceinline__ Search(WV &w_, const JobRegs &j_) : W(w_), j(j_) {}

    // table entry [i][c] (c is a constant at every call site)
    template <int TAB> static __device__ __forceinline__ int tab(int i, int c) { return WV::template tab<TAB>(i, c); }

    __device__ __forceinline__ bool in_range(int x, int y) const   // MV::checkRange, common/mv.h:96-99
    {
        return x >= j.mvmin_x && x <= j.mvmax_x && y >= j.mvmin_y && y <= j.mvmax_y;
    }

    // strict-less update with a candidate labelled (lx,ly)
    __device__ __forceinline__ bool take(int c, int lx, int ly)
    {
        if (c < bcost)
        {
            bcost = c;
            bmx = XA_S16(lx);
            bmy = XA_S16(ly);
            return true;
        }
        return false;
    }

    // COST_MV (motion.cpp:173-179)
    __device__ __forceinline__ void try_mv(int mx, int my)
    {
        take(W.cost_fpel(mx, my), mx, my);
    }

    // N candidates measured at (sx,sy)[k], labelled and priced as (lx,ly)[k], taken in order
    template <int N>
    __device__ __forceinline__ int try_points(const int *sx, const int *sy, const int *lx, const int *ly)
    {
        int c[N], last = -1;
        W.template eval_points<N>(sx, sy, lx, ly, c);
#pragma unroll
        for (int k = 0; k < N; k++)
            if (take(c[k], lx[k], ly[k]))
                last = k;
        return last;
    }

    // N candidates (cx,cy) + d[k], all relative to one centre, in order; returns index of the last taken or -1.
    // Only the cost is updated; the caller moves bmv (pattern loops move after the whole set).
    template <int N, int TAB>
    __device__ __forceinline__ int probe(int cx, int cy, int first)
    {
        int x[N], y[N], c[N], last = -1;
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            x[k] = XA_S16(cx + tab<TAB>(first + k, 0));
            y[k] = XA_S16(cy + tab<TAB>(first + k, 1));
        }
        W.template eval_points<N>(x, y, x, y, c);
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            if (c[k] < bcost)
            {
                bcost = c[k];
                last = k;
            }
        }
        return last;
    }

    // ---------------------------------------------------------------- STAR (motion.cpp:868-1108)
    __device__ __forceinline__ void pt(int mx, int my, int point, int dist, int &bPointNr, int &bDistance)
    {
        if (take(W.cost_fpel(mx, my), mx, my))
        {
            bPointNr = point;
            bDistance = dist;
        }
    }

    template <int N>
    __device__ __forceinline__ void pts(const int *x, const int *y, const int *point, const int *dist, int &bPointNr, int &bDistance)
    {
        int c[N];
        W.template eval_points<N>(x, y, x, y, c);
#pragma unroll
        for (int k = 0; k < N; k++)
            if (take(c[k], x[k], y[k]))
            {
                bPointNr = point[k];
                bDistance = dist[k];
            }
    }

    __device__ __forceinline__ void star_pattern(int &bPointNr, int &bDistance, int earlyExit, int merange)
    {
        const int ox = bmx, oy = bmy;
        int saved = bcost, rounds = 0;
        {
            const int top = XA_S16(oy - 1), bottom = XA_S16(oy + 1), left = XA_S16(ox - 1), right = XA_S16(ox + 1);
            if (top >= j.mvmin_y && left >= j.mvmin_x && right <= j.mvmax_x && bottom <= j.mvmax_y)
            {
                int x[4] = { ox, left, right, ox }, y[4] = { top, oy, oy, bottom };
                int p[4] = { 2, 4, 5, 7 }, d[4] = { 1, 1, 1, 1 };
                pts<4>(x, y, p, d, bPointNr, bDistance);
            }
            else
            {
                if (top >= j.mvmin_y)    pt(ox, top, 2, 1, bPointNr, bDistance);
                if (left >= j.mvmin_x)   pt(left, oy, 4, 1, bPointNr, bDistance);
                if (right <= j.mvmax_x)  pt(right, oy, 5, 1, bPointNr, bDistance);
                if (bottom <= j.mvmax_y) pt(ox, bottom, 7, 1, bPointNr, bDistance);
            }
            if (bcost < saved)
                rounds = 0;
            else if (++rounds >= earlyExit)
                return;
        }
        for (int dist = 2; dist <= 8; dist <<= 1)
        {
            const int top = XA_S16(oy - dist), bottom = XA_S16(oy + dist), left = XA_S16(ox - dist), right = XA_S16(ox + dist);
            const int hd = dist >> 1;
            const int top2 = XA_S16(oy - hd), bottom2 = XA_S16(oy + hd), left2 = XA_S16(ox - hd), right2 = XA_S16(ox + hd);
            saved = bcost;
            if (top >= j.mvmin_y && left >= j.mvmin_x && right <= j.mvmax_x && bottom <= j.mvmax_y)
            {
                int x[8] = { ox, left2, right2, left, right, left2, right2, ox };
                int y[8] = { top, top2, top2, oy, oy, bottom2, bottom2, bottom };
                int p[8] = { 2, 1, 3, 4, 5, 6, 8, 7 };
                int d[8] = { dist, hd, hd, dist, dist, hd, hd, dist };
                pts<8>(x, y, p, d, bPointNr, bDistance);
            }
            else
            {
                if (top >= j.mvmin_y) pt(ox, top, 2, dist, bPointNr, bDistance);
                if (top2 >= j.mvmin_y)
                {
                    if (left2 >= j.mvmin_x)  pt(left2, top2, 1, hd, bPointNr, bDistance);
                    if (right2 <= j.mvmax_x) pt(right2, top2, 3, hd, bPointNr, bDistance);
                }
                if (left >= j.mvmin_x)  pt(left, oy, 4, dist, bPointNr, bDistance);
                if (right <= j.mvmax_x) pt(right, oy, 5, dist, bPointNr, bDistance);
                if (bottom2 <= j.mvmax_y)
                {
                    if (left2 >= j.mvmin_x)  pt(left2, bottom2, 6, hd, bPointNr, bDistance);
                    if (right2 <= j.mvmax_x) pt(right2, bottom2, 8, hd, bPointNr, bDistance);
                }
                if (bottom <= j.mvmax_y) pt(ox, bottom, 7, dist, bPointNr, bDistance);
            }
            if (bcost < saved)
                rounds = 0;
            else if (++rounds >= earlyExit)
                return;
        }
        for (int dist = 16; dist <= XA_S16(merange); dist = XA_S16(dist << 1))
        {
            const int top = XA_S16(oy - dist), bottom = XA_S16(oy + dist), left = XA_S16(ox - dist), right = XA_S16(ox + dist);
            const int q = dist >> 2;
            saved = bcost;
            if (top >= j.mvmin_y && left >= j.mvmin_x && right <= j.mvmax_x && bottom <= j.mvmax_y)
            {
                int zero[4] = { 0, 0, 0, 0 }, dd[4] = { dist, dist, dist, dist };
                {
                    int x[4] = { ox, left, right, ox }, y[4] = { top, oy, oy, bottom };
                    pts<4>(x, y, zero, dd, bPointNr, bDistance);
                }
                for (int idx = 1; idx < 4; idx++)
                {
                    int yt = XA_S16(top + q * idx), yb = XA_S16(bottom - q * idx);
                    int xl = XA_S16(ox - q * idx), xr = XA_S16(ox + q * idx);
                    int x[4] = { xl, xr, xl, xr }, y[4] = { yt, yt, yb, yb };
                    pts<4>(x, y, zero, dd, bPointNr, bDistance);
                }
            }
            else
            {
                if (top >= j.mvmin_y)    pt(ox, top, 0, dist, bPointNr, bDistance);
                if (left >= j.mvmin_x)   pt(left, oy, 0, dist, bPointNr, bDistance);
                if (right <= j.mvmax_x)  pt(right, oy, 0, dist, bPointNr, bDistance);
                if (bottom <= j.mvmax_y) pt(ox, bottom, 0, dist, bPointNr, bDistance);
                for (int idx = 1; idx < 4; idx++)
                {
                    int yt = XA_S16(top + q * idx), yb = XA_S16(bottom - q * idx);
                    int xl = XA_S16(ox - q * idx), xr = XA_S16(ox + q * idx);
                    if (yt >= j.mvmin_y)
                    {
                        if (xl >= j.mvmin_x) pt(xl, yt, 0, dist, bPointNr, bDistance);
                        if (xr <= j.mvmax_x) pt(xr, yt, 0, dist, bPointNr, bDistance);
                    }
                    if (yb <= j.mvmax_y)
                    {
                        if (xl >= j.mvmin_x) pt(xl, yb, 0, dist, bPointNr, bDistance);
                        if (xr <= j.mvmax_x) pt(xr, yb, 0, dist, bPointNr, bDistance);
                    }
                }
            }
            if (bcost < saved)
                rounds = 0;
            else if (++rounds >= earlyExit)
                return;
        }
    }

    __device__ __forceinline__ void star_two_points(int bPointNr)
    {
        const int x1 = XA_S16(bmx + tab<T_STAR>((bPointNr - 1) * 2, 0)), y1 = XA_S16(bmy + tab<T_STAR>((bPointNr - 1) * 2, 1));
        const int x2 = XA_S16(bmx + tab<T_STAR>((bPointNr - 1) * 2 + 1, 0)), y2 = XA_S16(bmy + tab<T_STAR>((bPointNr - 1) * 2 + 1, 1));
        if (in_range(x1, y1)) try_mv(x1, y1);
        if (in_range(x2, y2)) try_mv(x2, y2);
    }

    // ---------------------------------------------------------------- UMH helpers (motion.cpp:443-628)
    // Quirk kept from the reference: inside the UMH block every x4 evaluation measures its SADs
    // relative to the search START (pix_base is computed once, motion.cpp:451) while the candidates
    // are labelled and priced relative to the CURRENT omv.
    __device__ __forceinline__ void umh_x4(int basex, int basey, int omx, int omy,
                           int a0, int a1, int b0, int b1, int c0, int c1, int d0, int d1)
    {
        int sx[4] = { basex + a0, basex + b0, basex + c0, basex + d0 };
        int sy[4] = { basey + a1, basey + b1, basey + c1, basey + d1 };
        int lx[4] = { XA_S16(omx + a0), XA_S16(omx + b0), XA_S16(omx + c0), XA_S16(omx + d0) };
        int ly[4] = { XA_S16(omy + a1), XA_S16(omy + b1), XA_S16(omy + c1), XA_S16(omy + d1) };
        try_points<4>(sx, sy, lx, ly);
    }

    // CROSS (motion.cpp:251-275).  Second quirk: COST_MV(omv.x, omv.y + i) expands to
    // fref + omv.x + omv.y + i * stride, i.e. the vertical singles are measured at
    // (omv.x + omv.y, +-i) but labelled (omv.x, omv.y +- i).
    __device__ __forceinline__ void umh_cross(int basex, int basey, int omx, int omy, int start, int x_max, int y_max)
    {
        int i = XA_S16(start);
        int lim = j.mvmax_x - omx < omx - j.mvmin_x ? j.mvmax_x - omx : omx - j.mvmin_x;
        if (x_max <= lim)
            for (; i < x_max - 2; i = XA_S16(i + 4))
                umh_x4(basex, basey, omx, omy, i, 0, -i, 0, i + 2, 0, -i - 2, 0);
        for (; i < x_max; i = XA_S16(i + 2))
        {
            if (omx + i <= j.mvmax_x) try_mv(omx + i, omy);
            if (omx - i >= j.mvmin_x) try_mv(omx - i, omy);
        }
        i = XA_S16(start);
        lim = j.mvmax_y - omy < omy - j.mvmin_y ? j.mvmax_y - omy : omy - j.mvmin_y;
        if (y_max <= lim)
            for (; i < y_max - 2; i = XA_S16(i + 4))
                umh_x4(basex, basey, omx, omy, 0, i, 0, -i, 0, i + 2, 0, -i - 2);
        for (; i < y_max; i = XA_S16(i + 2))
        {
            if (omy + i <= j.mvmax_y)
            {
                int sx = omx + omy, sy = i, lx = omx, ly = omy + i;
                try_points<1>(&sx, &sy, &lx, &ly);
            }
            if (omy - i >= j.mvmin_y)
            {
                int sx = omx + omy, sy = -i, lx = omx, ly = omy - i;
                try_points<1>(&sx, &sy, &lx, &ly);
            }
        }
    }

    // The predictor phase (motion.cpp:302-342) with every candidate priced in ONE evaluation call:
    // the predictor, its full-pel rounding, the zero vector and the caller's candidates do not depend
    // on one another's costs, only the bookkeeping does, and that is replayed afterwards in the
    // reference's order.  Used by the kernels for half-resolution references, whose evaluations are
    // latency-bound plane reads.  Sets bmx / bmy / bcost; returns bprecost.
    __device__ __forceinline__ int predictors_batched(int pmx, int pmy, int &bprex, int &bprey,
                                                      int qminx, int qminy, int qmaxx, int qmaxy)
    {
        auto clipq = [](int v, int lo, int hi) { int c = v > hi ? hi : v; return c < lo ? lo : c; };
        const bool sub = ((pmx | pmy) & 3) != 0, nz = pmx || pmy;
        bmx = XA_S16(XA_S16(pmx + 2) >> 2);
        bmy = XA_S16(XA_S16(pmy + 2) >> 2);
        // fixed slots (static indices only: everything stays in scalar registers): 0 the predictor, 1 its rounded
        // position, 2 the zero vector, 3.. the candidates; a slot that is not wanted repeats the predictor and its cost
        // is ignored (the usual 5 to 7 candidates are two passes of the side-by-side forms either way)
        static_assert(X265AMD_MAX_MVC <= 4, "eight slots");
        int qx[8], qy[8];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            qx[k] = pmx;
            qy[k] = pmy;
        }
        if (sub)
        {
            qx[1] = XA_S16(bmx << 2);
            qy[1] = XA_S16(bmy << 2);
        }
        if (nz)
            qx[2] = qy[2] = 0;
        int cmx[X265AMD_MAX_MVC], cmy[X265AMD_MAX_MVC];
        bool use[X265AMD_MAX_MVC];
        int wanted = 1 + (sub ? 1 : 0) + (nz ? 1 : 0), ns = nz ? 3 : sub ? 2 : 1;     // slots in use: up to the last wanted one
#pragma unroll
        for (int i = 0; i < X265AMD_MAX_MVC; i++)
        {
            cmx[i] = clipq(j.mvc[i][0], qminx, qmaxx);
            cmy[i] = clipq(j.mvc[i][1], qminy, qmaxy);
            use[i] = i < j.numc && (cmx[i] || cmy[i]) && !(cmx[i] == pmx && cmy[i] == pmy);
            if (use[i])
            {
                qx[3 + i] = cmx[i];
                qy[3 + i] = cmy[i];
                wanted++;
                ns = 4 + i;
            }
        }
        int c[8];
        W.subpel_cost8(ns, qx, qy, W.ikind, c, CF_NOMV0);     // the predictor itself carries no MV cost (motion.cpp:305)
        W.evals -= (unsigned)(ns - wanted);
        int bprecost = c[0];
        bcost = bprecost;
        if (sub)
            bcost = c[1];
        if (nz)
        {
            const int cz = c[2];
            if (cz < bcost)
            {
                bcost = cz;
                bmx = bmy = 0;
            }
        }
#pragma unroll
        for (int i = 0; i < X265AMD_MAX_MVC; i++)
        {
            if (!use[i])
                continue;
            if (cmx[i] == bprex && cmy[i] == bprey)
            {
                W.evals--;              // the reference skips a candidate equal to the best so far (motion.cpp:325)
                continue;
            }
            const int cc = c[3 + i];
            if (cc < bprecost)
            {
                bprecost = cc;
                bprex = cmx[i];
                bprey = cmy[i];
            }
        }
        return bprecost;
    }

    // Exhaustive raster scan of [mvmin, mvmax] with strict `<` (motion.cpp:738-771 and
    // TEncSearch::xPatternSearch, TEncSearch.cpp:3026-3085), improving on the current bcost / bmv.
    __device__ __forceinline__ void raster_scan(const FullBest *fb, int fb_stride)
    {
        if (fb)
        {
            // the exhaustive raster scan ran in its own kernel; its winner is the first
            // minimum in raster order, which is what strict-`<` updates would have kept
            // combine the slabs: lowest cost, then lowest raster index
            const int nxr = j.mvmax_x - j.mvmin_x + 1;
            const SlabBest sb = W.slab_best(fb, fb_stride > 16 ? 16 : fb_stride, nxr, j.mvmin_x, j.mvmin_y);
            const int fcost = sb.cost, fmv = sb.mv, fx = XA_S16(fmv & 0xffff), fy = XA_S16(fmv >> 16);
            if (fcost < bcost)
            {
                bcost = fcost;
                bmx = fx;
                bmy = fy;
            }
            int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
            W.evals += (nx > 0 && ny > 0) ? (unsigned)(nx * ny) : 0u;
        }
        else
        {
            for (int ty = j.mvmin_y; ty <= j.mvmax_y; ty++)
            {
                int tx = j.mvmin_x;
                for (; tx + 3 <= j.mvmax_x; tx += 4)
                {
                    int x[4] = { tx, tx + 1, tx + 2, tx + 3 }, y[4] = { ty, ty, ty, ty };
                    try_points<4>(x, y, x, y);
                }
                for (; tx <= j.mvmax_x; tx++)
                    try_mv(tx, ty);
            }
        }
    }

    // The search of TEncSearch::xMotionEstimation (bidirectional refinement, TEncSearch.cpp:2993-3000):
    // xPatternSearch, then xPatternRefinement (TEncSearch.cpp:224-285) at half-pel and at quarter-pel -
    // 9 candidates each, SATD of the interpolated block + mvcost, first minimum in table order.
    __device__ __forceinline__ int run_bipred(const FullBest *fb, int fb_stride, int &outx, int &outy)
    {
        W.ikind = X265AMD_SAD;
        bcost = 0x7fffffff;
        bmx = bmy = 0;
        raster_scan(fb, fb_stride);
        int ux = bmx, uy = bmy;             // full-pel, then half-pel, then quarter-pel units
#pragma nounroll
        for (int pass = 0; pass < 2; pass++)
        {
            ux = XA_S16(ux << 1);
            uy = XA_S16(uy << 1);
            auto rt = [&](int i, int c) { return pass ? tab<T_REFINE_Q>(i, c) : tab<T_REFINE_H>(i, c); };
            const int scale = pass ? 1 : 2;
            int qx[8], qy[8], c[8];
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                qx[i] = XA_S16(XA_S16(ux + rt(i, 0)) * scale);
                qy[i] = XA_S16(XA_S16(uy + rt(i, 1)) * scale);
            }
            W.subpel_cost8(8, qx, qy, X265AMD_SATD, c);
            const int c8 = W.subpel_cost(XA_S16(XA_S16(ux + rt(8, 0)) * scale), XA_S16(XA_S16(uy + rt(8, 1)) * scale), X265AMD_SATD, true);
            int best = 0;
            bcost = c[0];
#pragma unroll
            for (int i = 1; i < 8; i++)
                if (c[i] < bcost)
                {
                    bcost = c[i];
                    best = i;
                }
            if (c8 < bcost)
            {
                bcost = c8;
                best = 8;
            }
            ux = XA_S16(ux + rt(best, 0));
            uy = XA_S16(uy + rt(best, 1));
        }
        outx = ux;
        outy = uy;
        return bcost;
    }

    // One refinement round (motion.cpp:799-809, 813-845, 851-861): the first n of the 8 neighbours of
    // bmv at distance `step` (q-pel units), each subpelCompare + mvcost; improves bcost and returns
    // the winning direction (0: none).  Candidate data goes straight into the job state - no
    // per-candidate arrays are kept in scalar registers.
    __device__ __forceinline__ int subpel_round(int step, int n, int kind)
    {
        int c[8];
        W.round_costs(bmx, bmy, step, n, kind, c);
        int bdir = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            if (k < n && c[k] < bcost)
            {
                bcost = c[k];
                bdir = k + 1;
            }
        }
        return bdir;
    }

    // ---------------------------------------------------------------- the search
    __device__ __forceinline__ int run(const FullBest *fb, int fb_stride, int &outx, int &outy)
    {
        int merange = j.merange;
        W.ikind = j.icost;
        W.set_mvp(j.qmvp_x, j.qmvp_y);      // setMVP (bitcost.h:43)
        W.stamp(2);
        if (j.method == X265AMD_BIPRED)
            return run_bipred(fb, fb_stride, outx, outy);
        const int qminx = XA_S16(j.mvmin_x << 2), qminy = XA_S16(j.mvmin_y << 2);
        const int qmaxx = XA_S16(j.mvmax_x << 2), qmaxy = XA_S16(j.mvmax_y << 2);
        auto clipq = [](int v, int lo, int hi) { int c = v > hi ? hi : v; return c < lo ? lo : c; };   // MV::clipped, mv.h:88-93

        // predictor phase (motion.cpp:302-342)
        int pmx = clipq(j.qmvp_x, qminx, qmaxx), pmy = clipq(j.qmvp_y, qminy, qmaxy);
        int bprex = pmx, bprey = pmy;
        int bprecost;
        if (L::LR || L::CP)        // (one call for all predictor candidates: side by side in a half-resolution or candidate-parallel kernel)
            bprecost = predictors_batched(pmx, pmy, bprex, bprey, qminx, qminy, qmaxx, qmaxy);
        else
        {
            bprecost = W.subpel_cost(pmx, pmy, W.ikind, false);      // no mvcost term (motion.cpp:305)
            bmx = XA_S16(XA_S16(pmx + 2) >> 2);
            bmy = XA_S16(XA_S16(pmy + 2) >> 2);
            bcost = bprecost;
            if ((pmx | pmy) & 3)
                bcost = W.cost_fpel(bmx, bmy);
            if (pmx || pmy)
            {
                int c = W.cost_fpel(0, 0);
                if (c < bcost)
                {
                    bcost = c;
                    bmx = bmy = 0;
                }
            }
#pragma unroll
            for (int i = 0; i < X265AMD_MAX_MVC; i++)
            {
                if (i >= j.numc)
                    break;
                int mx = clipq(j.mvc[i][0], qminx, qmaxx), my = clipq(j.mvc[i][1], qminy, qmaxy);
                if ((mx || my) && !(mx == pmx && my == pmy) && !(mx == bprex && my == bprey))
                {
                    int c = W.subpel_cost(mx, my, W.ikind, true);
                    if (c < bprecost)
                    {
                        bprecost = c;
                        bprex = mx;
                        bprey = my;
                    }
                }
            }
        }
        pmx = XA_S16(XA_S16(pmx + 2) >> 2);
        pmy = XA_S16(XA_S16(pmy + 2) >> 2);
        int omx = bmx, omy = bmy;
        bool do_hex = false;
        W.stamp(3);

        switch (j.method)
        {
        case X265AMD_DIA:   // motion.cpp:346-367
        {
            int i = merange;
            do
            {
                const int cx = bmx, cy = bmy;
                int k = probe<4, T_SQUARE1>(cx, cy, 1);
                if (k < 0)
                    break;
                bmx = XA_S16(cx + tab<T_SQUARE1>(k + 1, 0));
                bmy = XA_S16(cy + tab<T_SQUARE1>(k + 1, 1));
            }
            while (--i && in_range(bmx, bmy));
            break;
        }
        case X265AMD_HEX:
            do_hex = true;
            break;

        case X265AMD_UMH:
        {
            const int basex = bmx, basey = bmy;
            const int scale = (W.w * W.h) >> 4;           // size_scale[] (motion.cpp:64-80)
            int cross_start = 1;
#define SAD_THRESH(v) (bcost < (((v) >> 4) * scale))
            int ucost1 = bcost, ucost2;
            omx = pmx; omy = pmy;
            umh_x4(basex, basey, omx, omy, 0, -1, 0, 1, -1, 0, 1, 0);
            if (pmx || pmy)
            {
                omx = 0; omy = 0;
                umh_x4(basex, basey, omx, omy, 0, -1, 0, 1, -1, 0, 1, 0);
            }
            ucost2 = bcost;
            if ((bmx || bmy) && !(bmx == pmx && bmy == pmy))
            {
                omx = bmx; omy = bmy;
                umh_x4(basex, basey, omx, omy, 0, -1, 0, 1, -1, 0, 1, 0);
            }
            if (bcost == ucost2)
                cross_start = 3;
            omx = bmx; omy = bmy;
            bool terminated = false;
            if (bcost == ucost2 && SAD_THRESH(2000))
            {
                umh_x4(basex, basey, omx, omy, 0, -2, -1, -1, 1, -1, -2, 0);
                umh_x4(basex, basey, omx, omy, 2, 0, -1, 1, 1, 1, 0, 2);
                if (bcost == ucost1 && SAD_THRESH(500))
                    terminated = true;
                else if (bcost == ucost2)
                {
                    int range = XA_S16(XA_S16(merange >> 1) | 1);
                    umh_cross(basex, basey, omx, omy, 3, range, range);
                    umh_x4(basex, basey, omx, omy, -1, -2, 1, -2, -2, -1, 2, -1);
                    umh_x4(basex, basey, omx, omy, -2, 1, 2, 1, -1, 2, 1, 2);
                    if (bcost == ucost2)
                        terminated = true;
                    else
                        cross_start = range + 2;
                }
            }
            if (terminated)
                break;

            if (j.numc)   // adaptive range from predictor spread (motion.cpp:486-538)
            {
                int mvd, denom = 1;
                const bool is64 = (W.w == 64 && W.h == 64);
                auto iabs = [](int v) { return v < 0 ? -v : v; };
                if (j.numc == 1)
                {
                    if (is64)
                        mvd = 25;
                    else
                        mvd = iabs(j.qmvp_x - j.mvc[0][0]) + iabs(j.qmvp_y - j.mvc[0][1]);
                }
                else
                {
                    denom = j.numc - 1;
                    mvd = 0;
                    if (!is64)
                    {
                        mvd = iabs(j.qmvp_x - j.mvc[0][0]) + iabs(j.qmvp_y - j.mvc[0][1]);
                        denom++;
                    }
#pragma unroll
                    for (int i = 0; i < X265AMD_MAX_MVC - 1; i++)
                        if (i < j.numc - 1)
                            mvd += iabs(j.mvc[i][0] - j.mvc[i + 1][0]) + iabs(j.mvc[i][1] - j.mvc[i + 1][1]);
                }
                int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
                int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
                merange = (merange * tab<T_RANGE_MUL>(mvd_ctx, sad_ctx)) >> 2;
            }
#undef SAD_THRESH
            umh_cross(basex, basey, omx, omy, cross_start, merange, merange >> 1);
            umh_x4(basex, basey, omx, omy, -2, -2, -2, 2, 2, -2, 2, 2);

            // hexagon grid (motion.cpp:545-624)
            omx = bmx; omy = bmy;
            int i = 1;
            do
            {
                int m1 = j.mvmax_x - omx < omx - j.mvmin_x ? j.mvmax_x - omx : omx - j.mvmin_x;
                int m2 = j.mvmax_y - omy < omy - j.mvmin_y ? j.mvmax_y - omy : omy - j.mvmin_y;
                if (4 * i > (m1 < m2 ? m1 : m2))
                {
                    for (int k = 0; k < 16; k++)
                    {
                        int mx = XA_S16(omx + XA_S16(tab<T_HEX4>(k, 0) * i)), my = XA_S16(omy + XA_S16(tab<T_HEX4>(k, 1) * i));
                        if (in_range(mx, my))
                            try_mv(mx, my);
                    }
                }
                else
                {
                    // unclipped ring: 4 x sad_x4; mvcost read straight from the table rows at omv*4,
                    // not truncated to 16 bits (motion.cpp:547-548,575); bmv is set after all 16
                    int best = -1;
                    for (int q = 0; q < 4; q++)
                    {
                        int x[4], y[4], c[4];
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            x[k] = omx + tab<T_HEX4>(q * 4 + k, 0) * i;
                            y[k] = omy + tab<T_HEX4>(q * 4 + k, 1) * i;
                        }
                        W.template eval_points<4>(x, y, x, y, c, CF_NOTRUNC);
#pragma unroll
                        for (int k = 0; k < 4; k++)
                        {
                            if (c[k] < bcost)
                            {
                                bcost = c[k];
                                best = q * 4 + k;
                            }
                        }
                    }
                    if (best >= 0)
                    {
                        bmx = XA_S16(omx + i * tab<T_HEX4>(best, 0));
                        bmy = XA_S16(omy + i * tab<T_HEX4>(best, 1));
                    }
                }
                i = XA_S16(i + 1);
            }
            while (i <= (merange >> 2));
            if (in_range(bmx, bmy))
                do_hex = true;
            break;
        }

        case X265AMD_STAR:  // motion.cpp:630-737
        {
            // One loop holds the reference's first pattern pass (early exit after 3 rounds), its
            // two-point / raster refinements, and the repeated passes (early exit after 32 rounds,
            // motion.cpp:704-735), so the pattern code is instantiated once.
            int bPointNr = 0, bDistance = 0;
            for (bool first = true;; first = false)
            {
                star_pattern(bPointNr, bDistance, first ? 3 : 32, merange);
                if (bDistance == 1)
                {
                    // best distance 1: check the two missing outer points.  That ends the search,
                    // except after the first pass when one of them improved the cost.
                    if (!bPointNr)
                        break;
                    const int saved = bcost;
                    star_two_points(bPointNr);
                    if (!first || bcost == saved)
                        break;
                }
                if (first && bDistance > 5)      // raster refinement, step 5 (motion.cpp:667-701)
                {
                    for (int ty = j.mvmin_y; ty <= j.mvmax_y; ty = XA_S16(ty + 5))
                        for (int tx = j.mvmin_x; tx <= j.mvmax_x; tx = XA_S16(tx + 5))
                        {
                            if (tx + 15 <= j.mvmax_x)
                            {
                                int x[4] = { tx, tx + 5, tx + 10, tx + 15 }, y[4] = { ty, ty, ty, ty }, c[4];
                                // quirk: the 4th candidate is priced with mvcost(tmv << 3) (motion.cpp:694)
                                W.template eval_points<4>(x, y, x, y, c, 0, 3);
#pragma unroll
                                for (int k = 0; k < 4; k++)
                                    take(c[k], XA_S16(x[k]), ty);
                                tx = XA_S16(tx + 15);
                            }
                            else
                                try_mv(tx, ty);
                        }
                }
                if (bDistance <= 0)
                    break;
                bDistance = 0;
                bPointNr = 0;
            }
            break;
        }

        case X265AMD_FULL:  // motion.cpp:738-771
        default:
            raster_scan(fb, fb_stride);
            break;
        }

        if (do_hex)     // me_hex2 (motion.cpp:369-441)
        {
            int dir = probe<6, T_HEX2>(bmx, bmy, 1);
            if (dir >= 0)
            {
                bmx = XA_S16(bmx + tab<T_HEX2>(dir + 1, 0));
                bmy = XA_S16(bmy + tab<T_HEX2>(dir + 1, 1));
                for (int i = (merange >> 1) - 1; i > 0 && in_range(bmx, bmy); i--)
                {
                    int step = probe<3, T_HEX2>(bmx, bmy, dir);
                    if (step < 0)
                        break;
                    dir = (dir + step - 1 + 6) % 6;
                    bmx = XA_S16(bmx + tab<T_HEX2>(dir + 1, 0));
                    bmy = XA_S16(bmy + tab<T_HEX2>(dir + 1, 1));
                }
            }
            int sq = probe<8, T_SQUARE1>(bmx, bmy, 1);
            bmx = XA_S16(bmx + tab<T_SQUARE1>(sq + 1, 0));
            bmy = XA_S16(bmy + tab<T_SQUARE1>(sq + 1, 1));
        }

        W.stamp(4);
        // motion.cpp:778-784
        if (bprecost < bcost)
        {
            bmx = bprex;
            bmy = bprey;
            bcost = bprecost;
        }
        else
        {
            bmx = XA_S16(bmx << 2);
            bmy = XA_S16(bmy << 2);
        }

        // sub-pel refinement (motion.cpp:786-861).  The reference's half-pel loop interpolates one
        // (w+1)xh or wx(h+1) block per direction pair; both members share the fractional phase,
        // so each candidate equals an independent subpelCompare.
        const int wl[5] = { tab<T_WORKLOAD>(j.subme, 0), tab<T_WORKLOAD>(j.subme, 1), tab<T_WORKLOAD>(j.subme, 2),
                            tab<T_WORKLOAD>(j.subme, 3), tab<T_WORKLOAD>(j.subme, 4) };
        const int hkind = wl[4] ? X265AMD_SATD : W.ikind;
        if (wl[4])
            bcost = W.subpel_cost(bmx, bmy, X265AMD_SATD, true);
        // Half-pel rounds (step 2), then quarter-pel rounds (step 1), through one code path.  Each round
        // prices its 4 or 8 neighbours in one call; they are then taken in the reference's order with
        // its strict comparison.
#pragma nounroll
        for (int ph = 0; ph < 2; ph++)
        {
            const int iters = ph ? wl[2] : wl[0], dirs = ph ? wl[3] : wl[1], step = ph ? 1 : 2;
            const int kind = ph ? X265AMD_SATD : hkind;
            if (ph && !wl[4])
                bcost = W.subpel_cost(bmx, bmy, X265AMD_SATD, true);     // hpel used SAD: remeasure with SATD (motion.cpp:848)
#pragma nounroll
            for (int it = 0; it < iters; it++)
            {
                const int bdir = subpel_round(step, dirs, kind);
                bmx = XA_S16(bmx + tab<T_SQUARE1>(bdir, 0) * step);
                bmy = XA_S16(bmy + tab<T_SQUARE1>(bdir, 1) * step);
            }
        }
        W.stamp(5);
        outx = bmx;
        outy = bmy;
        return bcost;
    }
};

} // namespace xa
