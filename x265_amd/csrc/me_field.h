// Whole-frame motion field with the predictor chain of the inter-prediction search (gfx950).
//
// TEncSearch::predInterSearch (Lib/TLibEncoder/TEncSearch.cpp:2341-2455) searches the PUs of a picture
// in coding order; each search is centred on a predictor that comes from the PU's already coded
// neighbours: the AMVP candidate whose motion-compensated block matches the source best
// (xEstimateMvPredAMVP / xGetTemplateCost, TEncSearch.cpp:2767-2819, 2944-2958), with the neighbours'
// MVs as extra start candidates (m_mvPredictors, :2366-2368), inside a range clipped around the
// predictor (xSetSearchRange, :3007-3024).  This kernel runs that chain for a uniform grid of pu_w x pu_h
// PUs over a whole picture, in raster order, so that a caller can ask for "the field the encoder would
// compute if every PU of this size were inter-coded with its ME result" ahead of the CU analysis.
//
// Per PU at grid position (gx, gy), with the final MVs of its left / above / above-right / above-left
// neighbours:
//   * candidate A = left MV; candidate B = the first that exists of above-right, above, above-left
//     (the spatial AMVP scan order of HEVC); B is dropped when equal to A; with a co-located motion field (`col`, one
//     entry per 16x16 block, scaled by the caller to this reference distance, cost < 0 = none) the temporal candidate
//     follows while the list holds fewer than two; the list is padded with zero vectors to AMVP_MAX_NUM_CANDS = 2
//     (Lib/TLibCommon/CommonDef.h:161).
//   * mvp = the candidate with the smaller template cost = SAD of the source PU against the
//     motion-compensated block at the candidate clipped by TComDataCU::clipMv
//     (Lib/TLibCommon/TComDataCU.cpp:2724-2736); xPredInterLumaBlk (TComPrediction.cpp:464-496) is the
//     interpolation of subpelCompare.  Both index costs are one bit (xGetMvpIdxBits with num = 2), so the
//     rate term cannot change the order; the first candidate wins a tie (`bestCost > cost`, :2803).
//   * mvmin / mvmax = xSetSearchRange(mvp, merange) without the frame-parallel row clamp;
//   * motionEstimate(ref, mvmin, mvmax, mvp, 3, {left, above, above-right MV or zero}, merange).
//
// One workgroup owns one PU row and walks it left to right; PU (gx, gy) starts when the row above has
// finished PU gx + 1 (its last neighbour to appear), so rows form a wavefront two PUs apart - the CTU-row
// wavefront of the reference's own encoder (encoder/frameencoder.cpp:899-930).  The hand-over between rows
// is the one of the lookahead kernel (me_lowres.h): one 64-bit device-scope store per result into an array
// pre-filled with an impossible MV, polled by the row below with device-scope loads; a row waits only on
// the row dispatched just before it, and a time limit turns a lost predecessor into an error code.
//
// A field's time is its chain of dependent PUs (PUs per row + twice the rows), each PU a chain of dependent evaluation calls, so what
// counts is the latency of one PU's search.  The classes of PUs of at most 256 and at most 1024 pixels therefore run the search and
// every single evaluation inline on the row's first wavefront and keep three or seven more wavefronts parked at a barrier
// (JobLds::CP, me_device.h): a call that prices several candidates read from the plane - the 4 or 8 neighbours of a sub-pel round,
// the two template costs, the predictor candidates - is dealt to all of them, each wavefront pricing its candidates over the whole
// PU, and the costs come back through LDS.  The class of the largest PUs splits the PU's pixels over eight wavefronts instead.
#pragma once

#include "me_lowres.h"

namespace xa {

// part_mode 0: every PU is the 2Nx2N PU of its own CU.  1 (2NxN): the CU is pu_w x 2 pu_h, PU rows 2r / 2r + 1 are its upper /
// lower part; 2 (Nx2N): the CU is 2 pu_w x pu_h, PU columns 2c / 2c + 1 its left / right part.  CUs are coded in raster order, the
// parts of a CU one after the other (the partIdx loop of predInterSearch, :2341), and what changes against part_mode 0 follows
// from where the neighbour positions of TComDataCU::fillMvpCand (Lib/TLibCommon/TComDataCU.cpp:2553-2633) fall:
//   * clipMv and xSetSearchRange use the CU's position, not the PU's (both parts of a CU share their bounds);
//   * upper part of a 2NxN CU: its below-left neighbour is the lower part of the CU to the left, already coded, and the
//     left candidate scan takes below-left before left - A = that PU's MV;
//   * lower part of a 2NxN CU: above is the upper part of the same CU; above-right lies in the CU to the right, not coded
//     yet - B = the upper part's MV, and the third start candidate is zero;
//   * Nx2N: the neighbours are those of the uniform grid (the left part's left neighbour is the right part of the CU to
//     the left, the right part's is the left part of its own CU).
// part_mode 4 / 5 / 6 / 7 (2NxnU / 2NxnD / nLx2N / nRx2N, numbered as PartSize, Lib/TLibCommon/TypeDef.h:100-111): pu_w x pu_h is the
// CU, cut a quarter (4, 6) or three quarters (5, 7) of the way down / across; PU sizes and corner positions are those of
// deriveLeftRightTopIdxGeneral / deriveLeftBottomIdxGeneral (TComDataCU.cpp:1837-1944).  The cut moves, the CUs that the corner
// positions fall into do not: the neighbours of the two parts are those of mode 1 (4, 5) / mode 2 (6, 7).
// With upper / lower parts (modes 1, 4, 5) one workgroup owns a CU row (two PU rows, walked upper part, lower part, next CU).
// One field of a launch.  The fields of a launch may differ in PU size, partition mode and planes (they share the kernel class,
// i.e. the range of PU areas); workgroups [first_wg, first_wg + wg_rows) are the PU (CU) rows of the field, top row first, so a
// row's predecessor always has the lower workgroup index and is dispatched before it.
struct FieldDesc
{
    int32_t enc_plane, ref_plane;
    int32_t pu_w, pu_h, gw, wg_rows, part_mode;
    int32_t first_wg;          // first workgroup of the field in its launch
    int32_t first_out;         // first result / predictor of the field in the call's output arrays
    int32_t first_col;         // first entry of the field's co-located motion, -1: none
};

template <class L>
__global__ void __launch_bounds__(L::NT * L::CP_WAVES) k_field(const PlaneDesc *planes, const uint16_t *cost, const FieldDesc *fields, int n_fields,
                                                 int pic_w, int pic_h, int max_cu, int merange, int method, int subme, int qp, int flags,
                                                 const x265amd_me_result *col, x265amd_me_result *out, int *out_mvp, int *error,
                                                 unsigned long long *evals)
{
    __shared__ __align__(16) L lds;
    // the field of this workgroup: the last one that starts at or before it
    int lo = 0, hi = n_fields - 1;
    while (lo < hi)
    {
        const int mid = (lo + hi + 1) >> 1;
        if (fields[mid].first_wg <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const FieldDesc F = fields[lo];
    const int pu_w = F.pu_w, pu_h = F.pu_h, gw = F.gw, part_mode = F.part_mode;
    const int row = (int)blockIdx.x - F.first_wg;
    const bool amp = part_mode >= 4;
    const bool horiz = part_mode == 1 || part_mode == 4 || part_mode == 5, vert = part_mode == 2 || part_mode == 6 || part_mode == 7;
    const int parts = horiz ? 2 : 1;                                     // PU rows per workgroup
    const int cut_h = part_mode == 4 ? pu_h >> 2 : pu_h - (pu_h >> 2), cut_w = part_mode == 6 ? pu_w >> 2 : pu_w - (pu_w >> 2);
    out += F.first_out;
    if (out_mvp)
        out_mvp += F.first_out;
    const int col_w = (pic_w + 15) >> 4;                                 // 16x16 blocks per row of the co-located motion
    col = F.first_col >= 0 ? col + F.first_col : nullptr;
    if (L::NW > 1 && Dev<L>::wave_id() != 0)
    {
        Dev<L>::worker((uint32_t)(uintptr_t)(XA_LDS L*)&lds);
        return;
    }
    if (L::CP && threadIdx.x >= 64)
    {
        // (the candidate-parallel workers wait for the leader's first command: everything they read is written before it)
        Dev<L>::cp_worker((uint32_t)(uintptr_t)(XA_LDS L*)&lds, __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)));
        return;
    }
    const PlaneDesc &enc = planes[F.enc_plane];
    const PlaneDesc &ref = planes[F.ref_plane];
    unsigned n_evals = 0;
    int failed = 0;
    int last[2] = { 0, 0 };                                              // MV word of the PU to the left, per PU row of the workgroup
    int cur_top = 0;                                                     // 2NxN: MV word of the upper part of the current CU
    Wave<L> W;
    W.sliced = false;
#ifdef XA_PROFILE
    long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long prof_sq[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, prof_max[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
#endif
    auto clipq = [](int v, int lo, int hi) { int c = v > hi ? hi : v; return c < lo ? lo : c; };

    for (int gx = 0; gx < gw; gx++)
#pragma nounroll
      for (int part = 0; part < parts; part++)
      {
        const int gy = row * parts + part;
        const bool lower = part == 1;
        int px = gx * pu_w, py = gy * pu_h, bw = pu_w, bh = pu_h;
        int cux = part_mode == 2 ? (gx & ~1) * pu_w : px, cuy = part_mode == 1 ? py - part * pu_h : py;
        if (amp && horiz)
        {
            cuy = row * pu_h;
            py = cuy + (lower ? cut_h : 0);
            bh = lower ? pu_h - cut_h : cut_h;
        }
        if (amp && vert)
        {
            cux = (gx >> 1) * pu_w;
            px = cux + ((gx & 1) ? cut_w : 0);
            bw = (gx & 1) ? pu_w - cut_w : cut_w;
        }
        const bool have_l = gx > 0, have_a = gy > 0, have_ar = !lower && gy > 0 && gx + 1 < gw, have_al = gy > 0 && gx > 0;
        // (the PU is cached while the row above may still be busy)
        W.stamp(0);
        W.bind(lds, enc, ref, px, py, px, py, bw, bh, cost, flags, true);
        W.stamp(1);
        int w_al = 0, w_a = 0, w_ar = 0;
        if (lower)
            w_a = cur_top;                                              // (B = above exists: above-left is never asked for)
        else if (have_a && !failed)
        {
            const x265amd_me_result *above = out + (size_t)(gy - 1) * gw + gx;
            const int v = LookLeaf::wait_mv3(above, (have_al ? 1 : 0) | 2 | (have_ar ? 4 : 0));
            w_al = __builtin_amdgcn_readlane(v, 0);
            w_a = __builtin_amdgcn_readlane(v, 1);
            w_ar = __builtin_amdgcn_readlane(v, 2);
            failed |= (w_al == LOOK_UNSET) | (w_a == LOOK_UNSET) | (w_ar == LOOK_UNSET);
        }
        W.stamp(6);
        auto mvx = [](int word) { return XA_S16(word & 0xffff); };
        auto mvy = [](int word) { return XA_S16(word >> 16); };
        // the two AMVP candidates
        const int left = lower ? last[1] : last[0];                     // MV word of the PU to the left
        const int wa_cand = parts == 2 ? last[1] : left;                // candidate A: 2NxN upper part - below-left, else left
        const int wb = have_ar ? w_ar : have_a ? w_a : w_al;
        const bool have_b = have_a;                                     // (above exists whenever any of the three does)
        int c0x = 0, c0y = 0, c1x = 0, c1y = 0, n_cand = 0;
        if (have_l)
        {
            c0x = mvx(wa_cand); c0y = mvy(wa_cand);
            n_cand = 1;
            if (have_b && wb != wa_cand)
            {
                c1x = mvx(wb); c1y = mvy(wb);
                n_cand = 2;
            }
        }
        else if (have_b)
        {
            c0x = mvx(wb); c0y = mvy(wb);
            n_cand = 1;
        }
        if (col && n_cand < 2)
        {
            // The temporal candidate (fillMvpCand, TComDataCU.cpp:2635-2697): the co-located motion below-right of the PU unless that
            // lies outside the picture or in the next CTU row, else - also when that block has none - at the PU's centre
            // (xDeriveCenterIdx, :3104-3116); motion is kept per 16x16 block (xGetColMVP masks the address, :3010).  It goes in
            // unpruned and the list is cut to two (:2699-2702), so it counts only while there is room.
            const int rbx = px + bw, rby = py + bh;
            const int ctx = px + ((bw >> 3) << 2), cty = py + ((bh >> 3) << 2);
            const bool rb_ok = rbx < pic_w && rby < pic_h && (rby & (max_cu - 1)) != 0;
            x265amd_me_result t = col[(cty >> 4) * col_w + (ctx >> 4)];
            if (rb_ok)
            {
                const x265amd_me_result rb = col[(rby >> 4) * col_w + (rbx >> 4)];
                if (rb.cost >= 0)
                    t = rb;
            }
            if (t.cost >= 0)
            {
                if (n_cand == 0) { c0x = t.mvx; c0y = t.mvy; }
                else             { c1x = t.mvx; c1y = t.mvy; }
            }
        }
        // TComDataCU::clipMv for a CU at (cux, cuy)
        const int xmax = (pic_w + 8 - cux - 1) << 2, xmin = (-max_cu - 8 - cux + 1) << 2;
        const int ymax = (pic_h + 8 - cuy - 1) << 2, ymin = (-max_cu - 8 - cuy + 1) << 2;
        W.ikind = X265AMD_SAD;
        int mvpx = c0x, mvpy = c0y;
        if (c1x != c0x || c1y != c0y)
        {
            // xGetTemplateCost of both candidates (equal candidates cost the same and the first one stays)
            const int tqx[2] = { clipq(c0x, xmin, xmax), clipq(c1x, xmin, xmax) }, tqy[2] = { clipq(c0y, ymin, ymax), clipq(c1y, ymin, ymax) };
            int t[2];
            W.subpel_cost2(tqx, tqy, X265AMD_SAD, t);
            const int t0 = t[0], t1 = t[1];
            if (t1 < t0)
            {
                mvpx = c1x; mvpy = c1y;
            }
        }
        JobRegs j;
        j.pu_x = px; j.pu_y = py; j.w = bw; j.h = bh;
        // xSetSearchRange
        const int cpx = clipq(mvpx, xmin, xmax), cpy = clipq(mvpy, ymin, ymax);
        j.mvmin_x = XA_S16(clipq(XA_S16(cpx - (merange << 2)), xmin, xmax)) >> 2;
        j.mvmin_y = XA_S16(clipq(XA_S16(cpy - (merange << 2)), ymin, ymax)) >> 2;
        j.mvmax_x = XA_S16(clipq(XA_S16(cpx + (merange << 2)), xmin, xmax)) >> 2;
        j.mvmax_y = XA_S16(clipq(XA_S16(cpy + (merange << 2)), ymin, ymax)) >> 2;
        j.qmvp_x = mvpx; j.qmvp_y = mvpy;
        j.merange = merange; j.method = method; j.subme = subme; j.qp = qp;
        j.numc = 3; j.ref_idx = 0; j.icost = X265AMD_SAD;
        j.mvc[0][0] = have_l ? mvx(left) : 0;  j.mvc[0][1] = have_l ? mvy(left) : 0;
        j.mvc[1][0] = have_a ? mvx(w_a) : 0;   j.mvc[1][1] = have_a ? mvy(w_a) : 0;
        j.mvc[2][0] = have_ar ? mvx(w_ar) : 0; j.mvc[2][1] = have_ar ? mvy(w_ar) : 0;
        j.mvc[3][0] = j.mvc[3][1] = 0;

        Search<L> S(W, j);
        int ox, oy;
        const int c = S.run(nullptr, 0, ox, oy);
        n_evals += W.evals;
        const int word = (oy << 16) | (ox & 0xffff);
        if (lower) last[1] = word; else last[0] = word;
        if (!lower)
            cur_top = word;
        if (out_mvp)
            LookLeaf::store_word(out_mvp + (size_t)gy * gw + gx, (mvpy << 16) | (mvpx & 0xffff));
        LookLeaf::publish(out + (size_t)gy * gw + gx, ox, oy, c);
#ifdef XA_PROFILE       // developer aid: clocks per phase, summed over the PUs of every 8th row
        W.stamp(7);
        if ((blockIdx.x & 7) == 0)
        {
            const int from[8] = { 0, 1, 6, 2, 3, 4, 5, 0 }, to[8] = { 1, 6, 2, 3, 4, 5, 7, 7 };
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                // (slot 7 of the squares and maxima: the PU's own work, i.e. the loop without the wait)
                const long long d = W.tstamp[to[k]] - W.tstamp[from[k]] - (k == 7 ? W.tstamp[6] - W.tstamp[1] : 0);
                prof[k] += W.tstamp[to[k]] - W.tstamp[from[k]];
                prof_sq[k] += (unsigned long long)((d >> 6) * (d >> 6));
                prof_max[k] = (unsigned long long)d > prof_max[k] ? (unsigned long long)d : prof_max[k];
            }
        }
#endif
      }
    Dev<L>::post(W.base, Dev<L>::OP_EXIT);
    Dev<L>::cp_exit(W.base);
    LookLeaf::finish(evals + (blockIdx.x % EVAL_SLOTS) * 16, n_evals, error, failed);
#ifdef XA_PROFILE
    if (threadIdx.x == 0 && (blockIdx.x & 7) == 0)
        for (int k = 0; k < 8; k++)
        {
            atomicAdd(evals + EVAL_SLOTS * 16 + 8 + k, (unsigned long long)prof[k]);
            atomicAdd(evals + (EVAL_SLOTS + 2) * 16 + k, prof_sq[k]);
            atomicMax(evals + (EVAL_SLOTS + 2) * 16 + 8 + k, prof_max[k]);
            if (k < 6)
                atomicAdd(evals + (EVAL_SLOTS + 1) * 16 + 8 + k, (unsigned long long)W.cgen[k]);
        }
#endif
}

} // namespace xa
