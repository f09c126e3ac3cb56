/* TEST INFRASTRUCTURE ONLY -- never linked into or called from the product path.
 *
 * Thin extern "C" shim over the *unmodified* reference translation units
 * (compiled where they lie under /root/reference by oracle/Makefile into
 * oracle/_ref/).  It exposes the reference's own C primitives
 * (source/common/pixel.cpp, ipfilter.cpp), MV-cost table
 * (source/encoder/bitcost.cpp) and MotionEstimate::motionEstimate
 * (source/encoder/motion.cpp:277-866) through plain-C entry points so that
 * the CPU restatement in oracle/me_oracle.c can be validated against the real
 * reference and golden vectors can be generated from it.
 *
 * Built three times: -DHIGH_BIT_DEPTH=0 (pixel = uint8_t) and =1 (pixel = uint16_t) with the C
 * primitives only, and -DHIGH_BIT_DEPTH=0 -DREF_WITH_VEC=1 with the reference's intrinsic primitives
 * dispatched over them (the CPU-baseline build).
 */
#include "primitives.h"
#include "common.h"
#include "motion.h"
#include "bitcost.h"
#include "mv.h"
#include "TLibCommon/TComRom.h"

#include <string.h>
#include <stdint.h>

using namespace x265;

namespace {
/* BitCost keeps the per-QP table pointer protected (bitcost.h:57-65); a
 * subclass can read it after setQP(). */
struct CostPeek : public BitCost
{
    const uint16_t *table() const { return m_cost; }
};

struct RefME
{
    MotionEstimate me;
    ReferencePlanes planes;
};

bool g_init = false;
}

extern "C" {

int ref_pixel_bytes(void) { return (int)sizeof(pixel); }
int ref_with_vec(void) { return REF_WITH_VEC; }

int ref_init(void)
{
    if (!g_init)
    {
        x265_param_t param;
        memset(&param, 0, sizeof(param));
        param.logLevel = X265_LOG_NONE;
#if REF_WITH_VEC
        /* cpuid = 0: detect the CPU and layer the intrinsic primitives over the C ones, as the encoder
         * does (primitives.cpp:107-139) */
        x265_setup_primitives(&param, 0);
#else
        /* cpuid = 1: C primitives only (primitives.cpp:129-139 loops from 2) */
        x265_setup_primitives(&param, 1);
#endif
        g_init = true;
    }
    return 0;
}

void ref_set_bit_depth(int depth)
{
#if HIGH_BIT_DEPTH
    x265::g_bitDepth = depth;
#else
    (void)depth;
#endif
}

/* kind: 0 sad, 1 satd, 2 sa8d_inter (index = Partitions enum), 3 sa8d (index = SquareBlocks) */
int ref_cmp(int kind, int idx, pixel *fenc, intptr_t fencstride, pixel *fref, intptr_t frefstride)
{
    ref_init();
    switch (kind)
    {
    case 0: return primitives.sad[idx](fenc, fencstride, fref, frefstride);
    case 1: return primitives.satd[idx](fenc, fencstride, fref, frefstride);
    case 2: return primitives.sa8d_inter[idx](fenc, fencstride, fref, frefstride);
    case 3: return primitives.sa8d[idx](fenc, fencstride, fref, frefstride);
    }
    return -1;
}

void ref_sad_x3(int part, pixel *fenc, pixel *r0, pixel *r1, pixel *r2, intptr_t stride, int *res)
{
    ref_init();
    primitives.sad_x3[part](fenc, r0, r1, r2, stride, res);
}

void ref_sad_x4(int part, pixel *fenc, pixel *r0, pixel *r1, pixel *r2, pixel *r3, intptr_t stride, int *res)
{
    ref_init();
    primitives.sad_x4[part](fenc, r0, r1, r2, r3, stride, res);
}

int ref_partition_from_sizes(int w, int h) { return PartitionFromSizes(w, h); }

/* kind: 0 = ipfilter_pp[H_8], 1 = ipfilter_pp[V_8] (pixel -> pixel) */
void ref_ipfilter_pp(int kind, pixel *src, intptr_t srcStride, pixel *dst, intptr_t dstStride, int w, int h, int frac)
{
    ref_init();
    primitives.ipfilter_pp[kind ? FILTER_V_P_P_8 : FILTER_H_P_P_8](src, srcStride, dst, dstStride, w, h, g_lumaFilter[frac]);
}

void ref_ipfilter_ps_h(pixel *src, intptr_t srcStride, short *dst, intptr_t dstStride, int w, int h, int frac)
{
    ref_init();
    primitives.ipfilter_ps[FILTER_H_P_S_8](src, srcStride, dst, dstStride, w, h, g_lumaFilter[frac]);
}

void ref_ipfilter_sp_v(short *src, intptr_t srcStride, pixel *dst, intptr_t dstStride, int w, int h, int frac)
{
    ref_init();
    primitives.ipfilter_sp[FILTER_V_S_P_8](src, srcStride, dst, dstStride, w, h, g_lumaFilter[frac]);
}

void ref_lowres_core(pixel *src0, pixel *dst0, pixel *dsth, pixel *dstv, pixel *dstc,
                     intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    ref_init();
    primitives.frame_init_lowres_core(src0, dst0, dsth, dstv, dstc, src_stride, dst_stride, width, height);
}

/* out[i + 32768] = table[i] for i in [-32767, 32767]; out[0] (i = -32768) is
 * never written by the reference (bitcost.cpp:52-56) and is reported as 0. */
void ref_mvcost_table(int qp, uint16_t *out)
{
    CostPeek b;
    b.setQP(qp);
    const uint16_t *t = b.table();
    out[0] = 0;
    for (int i = -32767; i <= 32767; i++)
        out[i + 32768] = t[i];
}

/* mvcost / bitcost of q-pel mv relative to q-pel mvp at qp (bitcost.h:43-53) */
void ref_mvcost(int qp, int mvpx, int mvpy, int mvx, int mvy, int *lambda_cost, int *bits)
{
    CostPeek b;
    b.setQP(qp);
    b.setMVP(MV((int16_t)mvpx, (int16_t)mvpy));
    *lambda_cost = b.mvcost(MV((int16_t)mvx, (int16_t)mvy));
    *bits = b.bitcost(MV((int16_t)mvx, (int16_t)mvy));
}

void *ref_me_new(void)
{
    ref_init();
    return new RefME;
}

void ref_me_free(void *h) { delete (RefME*)h; }

/* One full reference call sequence:
 *   setQP; setSearchMethod; setSubpelRefine; setSourcePlane; setSourcePU; motionEstimate
 * enc / ref point at pixel (0,0) of border-extended planes.
 * lowres: if non-NULL, 4 plane pointers (F,H,V,C) and isLowres = true.
 * mv arguments are {x, y} int16 pairs.  Returns bcost; out_mv = q-pel MV. */
int ref_me_search(void *h,
                  pixel *enc, intptr_t encStride,
                  pixel *ref, intptr_t refStride, pixel **lowres,
                  int puOffset, int w, int hgt,
                  const int16_t *mvmin, const int16_t *mvmax, const int16_t *qmvp,
                  int numc, const int16_t *mvc, int merange,
                  int method, int subme, int qp, int16_t *out_mv)
{
    RefME *r = (RefME*)h;
    r->me.setQP(qp);
    r->me.setSearchMethod(method);
    r->me.setSubpelRefine(subme);
    r->me.setSourcePlane(enc, encStride);
    r->me.setSourcePU(puOffset, w, hgt);
    r->planes.fpelPlane = ref;
    r->planes.lumaStride = (int)refStride;
    r->planes.isLowres = lowres != NULL;
    if (lowres)
        for (int i = 0; i < 4; i++)
            r->planes.lowresPlane[i] = lowres[i];
    else
        r->planes.lowresPlane[0] = ref;
    MV cand[16];
    for (int i = 0; i < numc && i < 16; i++)
        cand[i] = MV(mvc[2 * i], mvc[2 * i + 1]);
    MV out;
    int cost = r->me.motionEstimate(&r->planes,
                                    MV(mvmin[0], mvmin[1]), MV(mvmax[0], mvmax[1]),
                                    MV(qmvp[0], qmvp[1]), numc, cand, merange, out);
    out_mv[0] = out.x;
    out_mv[1] = out.y;
    return cost;
}

/* The same over an array of jobs laid out like meo_job (oracle/me_oracle.h, 48 bytes each),
 * all against one reference plane: used to time the reference's CPU path without per-job FFI cost. */
struct ShimJob
{
    int32_t pu_x, pu_y;
    int16_t w, h, mvmin[2], mvmax[2], qmvp[2], merange;
    uint8_t method, subme, qp, numc, ref_idx, icost;
    int16_t mvc[4][2];
};

void ref_me_search_batch(void *h, pixel *enc, intptr_t encStride, pixel *ref, intptr_t refStride,
                         const void *jobs, int n, int16_t *out_mv, int *out_cost)
{
    const ShimJob *j = (const ShimJob*)jobs;
    for (int i = 0; i < n; i++)
        out_cost[i] = ref_me_search(h, enc, encStride, ref, refStride, NULL,
                                    (int)(j[i].pu_y * encStride + j[i].pu_x), j[i].w, j[i].h,
                                    j[i].mvmin, j[i].mvmax, j[i].qmvp, j[i].numc, &j[i].mvc[0][0],
                                    j[i].merange, j[i].method, j[i].subme, j[i].qp, out_mv + 2 * i);
}

/* Driver for the lookahead's motion search over one half-resolution frame: the loop order, search
 * bounds and neighbour candidates of Lookahead::estimateFrameCost / estimateCUCost
 * (encoder/slicetype.cpp:306-316, 361-407) around the reference's own MotionEstimate.
 * (slicetype.cpp itself needs the whole encoder to link; only its ME calls are replayed here.)
 * enc_planes / ref_planes: F,H,V,C plane pointers at pixel (0,0). */
static inline int16_t shim_median(int16_t a, int16_t b, int16_t c)
{
    int16_t lo = a < b ? a : b, hi = a < b ? b : a;
    int16_t m = hi < c ? hi : c;
    return lo > m ? lo : m;
}

void ref_lookahead_me(void *h, pixel **enc_planes, pixel **ref_planes, intptr_t stride, int cuw, int cuh,
                      int merange, int method, int subme, int qp, int16_t *out_mv, int *out_cost)
{
    RefME *r = (RefME*)h;
    r->me.setQP(qp);
    r->me.setSearchMethod(method);
    r->me.setSubpelRefine(subme);
    r->me.setSourcePlane(enc_planes[0], stride);
    r->planes.fpelPlane = ref_planes[0];
    r->planes.lumaStride = (int)stride;
    r->planes.isLowres = true;
    for (int i = 0; i < 4; i++)
        r->planes.lowresPlane[i] = ref_planes[i];
    MV *mvs = new MV[cuw * cuh];
    for (int cuy = cuh - 1; cuy >= 0; cuy--)
    {
        for (int cux = cuw - 1; cux >= 0; cux--)
        {
            const int xy = cux + cuy * cuw;
            r->me.setSourcePU(8 * cux + 8 * cuy * (int)stride, 8, 8);
            MV mvmin, mvmax, mvc[4], mvp;
            mvmin.x = (uint16_t)(-cux * 8 - 8);
            mvmin.y = (uint16_t)(-cuy * 8 - 8);
            mvmax.x = (uint16_t)((cuw - cux - 1) * 8 + 8);
            mvmax.y = (uint16_t)((cuh - cuy - 1) * 8 + 8);
            int numc = 0;
            mvc[0] = 0;
            mvc[2] = 0;
            if (cux < cuw - 1)
                mvc[numc++] = mvs[xy + 1];
            if (cuy < cuh - 1)
            {
                mvc[numc++] = mvs[xy + cuw];
                if (cux > 0)
                    mvc[numc++] = mvs[xy + cuw - 1];
                if (cux < cuw - 1)
                    mvc[numc++] = mvs[xy + cuw + 1];
            }
            if (numc <= 1)
                mvp = mvc[0];
            else
            {
                mvp.x = shim_median(mvc[0].x, mvc[1].x, mvc[2].x);
                mvp.y = shim_median(mvc[0].y, mvc[1].y, mvc[2].y);
            }
            out_cost[xy] = r->me.motionEstimate(&r->planes, mvmin, mvmax, mvp, numc, mvc, merange, mvs[xy]);
            out_mv[2 * xy] = mvs[xy].x;
            out_mv[2 * xy + 1] = mvs[xy].y;
        }
    }
    delete[] mvs;
}

/* Driver for the search of TEncSearch::xMotionEstimation (bidirectional refinement): the control flow
 * of xPatternSearch and xPatternRefinement (Lib/TLibEncoder/TEncSearch.cpp:224-285, 2993-3000,
 * 3026-3085; TEncSearch.cpp itself needs the whole encoder to link) around the reference's own
 * primitives, interpolation filters and MV cost.  enc holds the caller's modified block at the PU
 * position.  Returns the quarter-pel refinement's cost; out_mv = q-pel MV. */
unsigned ref_bipred_search(void *h, pixel *enc, intptr_t encStride, pixel *ref, intptr_t stride, int puOffset,
                           int width, int height, const int16_t *mvmin, const int16_t *mvmax, const int16_t *mvp,
                           int qp, int16_t *out_mv)
{
    static const MV refineHpel[9] = { MV(0, 0), MV(0, -1), MV(0, 1), MV(-1, 0), MV(1, 0), MV(-1, -1), MV(1, -1), MV(-1, 1), MV(1, 1) };
    static const MV refineQPel[9] = { MV(0, 0), MV(0, -1), MV(0, 1), MV(-1, -1), MV(1, -1), MV(-1, 0), MV(1, 0), MV(-1, 1), MV(1, 1) };
    RefME *r = (RefME*)h;
    r->me.setQP(qp);
    r->me.setSourcePlane(enc, encStride);
    r->me.setSourcePU(puOffset, width, height);
    r->me.setMVP(MV(mvp[0], mvp[1]));
    pixel *fenc = r->me.fenc;
    pixel *fref = ref + puOffset;
    int part = PartitionFromSizes(width, height);
    pixelcmp_t sad = primitives.sad[part];
    pixelcmp_x4_t sad_x4 = primitives.sad_x4[part];
    pixelcmp_t satd = primitives.satd[part];
    ALIGN_VAR_32(int, costs[16]);
    MV outmv(0, 0);
    unsigned bcost = ~0u;
    for (int y = mvmin[1]; y <= mvmax[1]; y++)
    {
        pixel *row = fref + y * stride;
        for (int x = mvmin[0]; x <= mvmax[0]; x++)
        {
            MV mv(x, y);
            if (x + 3 <= mvmax[0])
            {
                sad_x4(fenc, row + x, row + x + 1, row + x + 2, row + x + 3, stride, costs);
                for (int k = 0; k < 4; k++, mv.x++)
                {
                    unsigned c = costs[k] + r->me.mvcost(mv << 2);
                    if (c < bcost) { bcost = c; outmv = mv; }
                }
                x += 3;
            }
            else
            {
                unsigned c = sad(fenc, FENC_STRIDE, row + x, stride) + r->me.mvcost(mv << 2);
                if (c < bcost) { bcost = c; outmv = mv; }
            }
        }
    }
    ALIGN_VAR_32(pixel, qref[64 * 64]);
    short *tmp = new short[width * (height + 7)];
    for (int pass = 0; pass < 2; pass++)
    {
        const MV *tab = pass ? refineQPel : refineHpel;
        const int fracBits = pass ? 1 : 2;
        outmv <<= 1;
        unsigned best = 0;
        bcost = ~0u;
        for (int i = 0; i < 9; i++)
        {
            MV mv = (outmv + tab[i]) * (int16_t)fracBits;
            pixel *src = fref + (mv.x >> 2) + (mv.y >> 2) * stride;
            int xFrac = mv.x & 3, yFrac = mv.y & 3;
            unsigned cost = r->me.mvcost(mv);
            if (!yFrac && !xFrac)
                cost += satd(fenc, FENC_STRIDE, src, stride);
            else
            {
                if (!yFrac)
                    primitives.ipfilter_pp[FILTER_H_P_P_8](src, stride, qref, 64, width, height, g_lumaFilter[xFrac]);
                else if (!xFrac)
                    primitives.ipfilter_pp[FILTER_V_P_P_8](src, stride, qref, 64, width, height, g_lumaFilter[yFrac]);
                else
                {
                    primitives.ipfilter_ps[FILTER_H_P_S_8](src - 3 * stride, stride, tmp, width, width, height + 7, g_lumaFilter[xFrac]);
                    primitives.ipfilter_sp[FILTER_V_S_P_8](tmp + 3 * width, width, qref, 64, width, height, g_lumaFilter[yFrac]);
                }
                cost += satd(fenc, FENC_STRIDE, qref, 64);
            }
            if (cost < bcost) { bcost = cost; best = i; }
        }
        outmv += tab[best];
    }
    delete[] tmp;
    out_mv[0] = outmv.x;
    out_mv[1] = outmv.y;
    return bcost;
}

/* Driver for the motion searches of TEncSearch::predInterSearch over a uniform PU grid (TEncSearch.cpp itself
 * needs the whole encoder to link): the call sequence of :2364-2368, :2392, :2428-2429 around the reference's
 * own MotionEstimate, with xEstimateMvPredAMVP's choice (:2767-2819) made through the reference's bufSAD and
 * interpolation primitives exactly as xGetTemplateCost / xPredInterLumaBlk do (:2944-2958,
 * TComPrediction.cpp:464-496), TComDataCU::clipMv (TComDataCU.cpp:2724-2736) and xSetSearchRange (:3007-3024).
 * The AMVP list is the uniform-grid case: A = left MV, B = first of above-right / above / above-left MV, B
 * dropped if equal to A, zero-padded to two; both index costs are one bit (xGetMvpIdxBits), ties keep the first. */
static inline void shim_clip(MV &mv, int picw, int pich, int px, int py)
{
    int xmax = (picw + 8 - px - 1) << 2, xmin = (-64 - 8 - px + 1) << 2;
    int ymax = (pich + 8 - py - 1) << 2, ymin = (-64 - 8 - py + 1) << 2;
    mv.x = (int16_t)X265_MIN(xmax, X265_MAX(xmin, (int)mv.x));
    mv.y = (int16_t)X265_MIN(ymax, X265_MAX(ymin, (int)mv.y));
}

/* partMode 0: 2Nx2N PUs (every PU its own CU); 1: the parts of 2NxN CUs (puw x 2 puh), PU rows 2r / 2r + 1; 2: the parts of
 * Nx2N CUs (2 puw x puh), PU columns 2c / 2c + 1.  CUs in raster order, parts one after the other (the partIdx loop, :2341);
 * clipMv / xSetSearchRange from the CU's position; the upper part of a 2NxN CU takes candidate A from below-left (the lower
 * part of the CU to the left: TComDataCU::fillMvpCand asks below-left before left, TComDataCU.cpp:2574-2588), the lower part
 * has the upper part above it and nothing coded above-right.  4 / 5 / 6 / 7: the parts of 2NxnU / 2NxnD / nLx2N / nRx2N CUs of
 * puw x puh (sizes and corner positions: deriveLeftRightTopIdxGeneral / deriveLeftBottomIdxGeneral, TComDataCU.cpp:1837-1944);
 * their neighbours are those of partMode 1 / 2. */
struct ShimColMv { int16_t x, y; int32_t cost; };      /* one 16x16 block of the co-located picture; cost < 0: no motion */

/* colMvs (optional): the temporal candidate of fillMvpCand (TComDataCU.cpp:2635-2697) from the co-located picture's motion per
 * 16x16 block (xGetColMVP masks the partition address to that granularity, :3010), (picw + 15) / 16 entries per row, scaled
 * already: taken below-right of the PU unless that is outside the picture or in the next CTU row, else - and when that block
 * has no motion - at the PU's centre (xDeriveCenterIdx, :3104-3116); appended unpruned, the list is cut to two (:2699-2702). */
void ref_field_me_tmvp(void *h, pixel *enc, pixel *ref, intptr_t stride, int picw, int pich, int puw, int puh, int partMode,
                       const ShimColMv *colMvs, int merange, int method, int subme, int qp, int16_t *out_mv, int *out_cost,
                       int16_t *out_mvp)
{
    RefME *r = (RefME*)h;
    r->me.setQP(qp);
    r->me.setSearchMethod(method);
    r->me.setSubpelRefine(subme);
    r->me.setSourcePlane(enc, stride);
    r->planes.fpelPlane = ref;
    r->planes.lumaStride = (int)stride;
    r->planes.isLowres = false;
    r->planes.lowresPlane[0] = ref;
    // partMode 4..7: puw x puh is the CU of an asymmetric partition (PartSize numbering, TypeDef.h:100-111)
    const bool horiz = partMode == 1 || partMode == 4 || partMode == 5, vert = partMode == 2 || partMode == 6 || partMode == 7;
    const bool amp = partMode >= 4;
    int gw = picw / puw, gh = pich / puh;
    if (partMode == 1) gh &= ~1;
    if (partMode == 2) gw &= ~1;
    if (amp && horiz) gh *= 2;
    if (amp && vert) gw *= 2;
    const int cutH = partMode == 4 ? puh >> 2 : puh - (puh >> 2), cutW = partMode == 6 ? puw >> 2 : puw - (puw >> 2);
    const int parts = horiz ? 2 : 1;
    MV *mvs = new MV[gw * gh];
    ALIGN_VAR_32(pixel, pred[64 * 64]);
    short *tmp = new short[64 * (64 + 7)];
    for (int row = 0; row < gh / parts; row++)
    {
        for (int gx = 0; gx < gw; gx++)
          for (int part = 0; part < parts; part++)
          {
            const int gy = row * parts + part;
            const bool lower = part == 1;
            const int xy = gx + gy * gw;
            int px = gx * puw, py = gy * puh, bw = puw, bh = puh;
            int cux = partMode == 2 ? (gx & ~1) * puw : px, cuy = partMode == 1 ? py - part * puh : py;
            if (amp && horiz)
            {
                cuy = row * puh;
                py = cuy + (lower ? cutH : 0);
                bh = lower ? puh - cutH : cutH;
            }
            if (amp && vert)
            {
                cux = (gx >> 1) * puw;
                px = cux + ((gx & 1) ? cutW : 0);
                bw = (gx & 1) ? puw - cutW : cutW;
            }
            const int puOffset = px + py * (int)stride;
            r->me.setSourcePU(puOffset, bw, bh);
            const bool haveL = gx > 0, haveA = gy > 0, haveAR = !lower && gy > 0 && gx + 1 < gw, haveAL = gy > 0 && gx > 0;
            MV mvPredictors[3];                        // left, above, above-right (zero when absent)
            if (haveL)  mvPredictors[0] = mvs[xy - 1];
            if (haveA)  mvPredictors[1] = mvs[xy - gw];
            if (haveAR) mvPredictors[2] = mvs[xy - gw + 1];
            MV cand[2], b;
            bool haveB = haveAR || haveA || haveAL;
            if (haveAR) b = mvs[xy - gw + 1]; else if (haveA) b = mvs[xy - gw]; else if (haveAL) b = mvs[xy - gw - 1];
            int numCand = 0;
            if (haveL)
            {
                cand[numCand++] = (horiz && !lower) ? mvs[xy + gw - 1] : mvs[xy - 1];
                if (haveB && b != cand[0]) cand[numCand++] = b;
            }
            else if (haveB)
                cand[numCand++] = b;
            if (colMvs && numCand < 2)
            {
                const int cw = (picw + 15) >> 4;
                const int rbx = px + bw, rby = py + bh;
                const ShimColMv *t = NULL;
                if (rbx < picw && rby < pich && (rby & 63) && colMvs[(rby >> 4) * cw + (rbx >> 4)].cost >= 0)
                    t = &colMvs[(rby >> 4) * cw + (rbx >> 4)];
                else
                {
                    const int ctrx = px + (bw / 4 / 2) * 4, ctry = py + (bh / 4 / 2) * 4;
                    if (colMvs[(ctry >> 4) * cw + (ctrx >> 4)].cost >= 0)
                        t = &colMvs[(ctry >> 4) * cw + (ctrx >> 4)];
                }
                if (t)
                    cand[numCand++] = MV(t->x, t->y);
            }
            MV mvp = cand[0];
            if (cand[1] != cand[0])
            {
                unsigned bestCost = 0x7fffffff;
                for (int i = 0; i < 2; i++)
                {
                    MV mv = cand[i];
                    shim_clip(mv, picw, pich, cux, cuy);
                    pixel *src = ref + puOffset + (mv.x >> 2) + (mv.y >> 2) * stride;
                    int xFrac = mv.x & 3, yFrac = mv.y & 3;
                    unsigned cost;
                    if (!(xFrac | yFrac))
                    {
                        primitives.blockcpy_pp(bw, bh, pred, 64, src, stride);
                        cost = r->me.bufSAD(pred, 64);
                    }
                    else
                    {
                        if (!yFrac)
                            primitives.ipfilter_pp[FILTER_H_P_P_8](src, stride, pred, 64, bw, bh, g_lumaFilter[xFrac]);
                        else if (!xFrac)
                            primitives.ipfilter_pp[FILTER_V_P_P_8](src, stride, pred, 64, bw, bh, g_lumaFilter[yFrac]);
                        else
                        {
                            primitives.ipfilter_ps[FILTER_H_P_S_8](src - 3 * stride, stride, tmp, bw, bw, bh + 7, g_lumaFilter[xFrac]);
                            primitives.ipfilter_sp[FILTER_V_S_P_8](tmp + 3 * bw, bw, pred, 64, bw, bh, g_lumaFilter[yFrac]);
                        }
                        cost = r->me.bufSAD(pred, 64);
                    }
                    if (bestCost > cost)
                    {
                        bestCost = cost;
                        mvp = cand[i];
                    }
                }
            }
            MV cmvp = mvp;
            shim_clip(cmvp, picw, pich, cux, cuy);
            MV dist((int16_t)(merange << 2), (int16_t)(merange << 2));
            MV mvmin = cmvp - dist, mvmax = cmvp + dist;
            shim_clip(mvmin, picw, pich, cux, cuy);
            shim_clip(mvmax, picw, pich, cux, cuy);
            mvmin >>= 2;
            mvmax >>= 2;
            out_cost[xy] = r->me.motionEstimate(&r->planes, mvmin, mvmax, mvp, 3, mvPredictors, merange, mvs[xy]);
            out_mv[2 * xy] = mvs[xy].x;
            out_mv[2 * xy + 1] = mvs[xy].y;
            out_mvp[2 * xy] = mvp.x;
            out_mvp[2 * xy + 1] = mvp.y;
          }
    }
    delete[] tmp;
    delete[] mvs;
}

void ref_field_me_parts(void *h, pixel *enc, pixel *ref, intptr_t stride, int picw, int pich, int puw, int puh, int partMode,
                        int merange, int method, int subme, int qp, int16_t *out_mv, int *out_cost, int16_t *out_mvp)
{
    ref_field_me_tmvp(h, enc, ref, stride, picw, pich, puw, puh, partMode, NULL, merange, method, subme, qp, out_mv, out_cost, out_mvp);
}

void ref_field_me(void *h, pixel *enc, pixel *ref, intptr_t stride, int picw, int pich, int puw, int puh,
                  int merange, int method, int subme, int qp, int16_t *out_mv, int *out_cost, int16_t *out_mvp)
{
    ref_field_me_parts(h, enc, ref, stride, picw, pich, puw, puh, 0, merange, method, subme, qp, out_mv, out_cost, out_mvp);
}

/* bufSAD / bufSATD / bufSA8D of the cached PU (motion.h:89-93) */
int ref_me_buf(void *h, int kind, pixel *enc, intptr_t encStride, int puOffset, int w, int hgt,
               pixel *fref, intptr_t stride)
{
    RefME *r = (RefME*)h;
    r->me.setSourcePlane(enc, encStride);
    r->me.setSourcePU(puOffset, w, hgt);
    switch (kind)
    {
    case 0: return r->me.bufSAD(fref, stride);
    case 1: return r->me.bufSATD(fref, stride);
    case 2: return r->me.bufSA8D(fref, stride);
    }
    return -1;
}
} // extern "C"
