// Host-side C++ mirror of the reference's motion-estimation interface, on top of the C ABI
// (include/x265_amd.h).  Same method names, argument meaning and call order as
//   class MotionEstimate        (reference: source/encoder/motion.h:35-112)
//   struct MV                   (reference: source/common/mv.h:36-100)
//   Setup_*_Primitives(p, ...)  (reference: source/common/primitives.h:306-308)
// so that the call sites in TEncSearch::predInterSearch (Lib/TLibEncoder/TEncSearch.cpp:2332-2451)
// and Lookahead::estimateCUCost (encoder/slicetype.cpp:354-407) can switch over one call at a time.
//
// Two ways to search:
//   * motionEstimate(...)  -- the reference's synchronous per-PU call; runs a batch of one job;
//   * queue(...) / flush() -- collect the jobs of a row / frame, run them as one batch, read the
//     per-PU MV/cost field back.  This is the path that makes the GPU worthwhile.
#pragma once

#include <stdint.h>
#include <stdexcept>
#include <string>
#include <string.h>
#include <map>
#include <tuple>
#include <vector>

#include "../../include/x265_amd.h"

namespace x265amd {

struct MV
{
    int16_t x, y;
    MV() : x(0), y(0) {}
    MV(int16_t x_, int16_t y_) : x(x_), y(y_) {}
    bool operator==(const MV &o) const { return x == o.x && y == o.y; }
    bool operator!=(const MV &o) const { return !(*this == o); }
    MV operator<<(int i) const { return MV((int16_t)(x << i), (int16_t)(y << i)); }
    MV operator>>(int i) const { return MV((int16_t)(x >> i), (int16_t)(y >> i)); }
    MV roundToFPel() const { return MV((int16_t)(x + 2), (int16_t)(y + 2)) >> 2; }
    MV toQPel() const { return *this << 2; }
};

// A registered plane stands in for ReferencePlanes::fpelPlane / lumaStride (common/reference.h:36-53)
struct ReferencePlanes
{
    int plane_id = -1;
    int lumaStride = 0;
    // picture size and margins (TComPicYuv::getWidth / getLumaMarginX ...): what turns the reference's
    // linear PU offsets back into coordinates (MotionEstimate::setSourcePU)
    int width = 0, height = 0, marginX = 0, marginY = 0;
};

// Lowres (common/lowres.h:41-72), picture part: the F/H/V/C half-resolution planes live on the device;
// as a search reference it behaves as ReferencePlanes with isLowres set
struct Lowres : ReferencePlanes
{
    int lines = 0;                          // width (inherited) and lines, rounded up to whole 8x8 CUs (lowres.cpp:44-46)
    int widthInCU = 0, heightInCU = 0;      // slicetype.cpp:74-75
};

class Context
{
public:
    explicit Context(int device = 0)
    {
        int rc = x265amd_create(device, &m_ctx);
        if (rc)
            throw std::runtime_error("x265amd_create failed with " + std::to_string(rc));
    }
    ~Context() { x265amd_destroy(m_ctx); }
    Context(const Context&) = delete;
    Context &operator=(const Context&) = delete;

    x265amd_ctx *get() const { return m_ctx; }

    // Upload a TComPicYuv-style padded luma plane (host memory).
    ReferencePlanes uploadPlane(const void *base, int width, int height, int marginX, int marginY,
                                intptr_t stride, int bytesPerPixel, int bitDepth)
    {
        ReferencePlanes r;
        check(x265amd_plane_upload(m_ctx, base, width, height, marginX, marginY, stride, bytesPerPixel, bitDepth, &r.plane_id));
        return describe(r.plane_id);
    }
    // the ReferencePlanes of a plane registered through the C ABI (wrapped device memory, row-wise planes)
    ReferencePlanes describe(int plane_id)
    {
        ReferencePlanes r;
        intptr_t stride = 0;
        check(x265amd_plane_geometry(m_ctx, plane_id, &r.width, &r.height, &r.marginX, &r.marginY, &stride, NULL, NULL));
        r.plane_id = plane_id;
        r.lumaStride = (int)stride;
        return r;
    }
    void releasePlane(const ReferencePlanes &p) { check(x265amd_plane_release(m_ctx, p.plane_id)); }

    // Lowres::create + Lowres::init (common/lowres.cpp:30-58, 109-186): the half-resolution plane set
    // of an uploaded picture (margins holding replicated borders), built on the device
    Lowres createLowres(const ReferencePlanes &src)
    {
        Lowres l;
        check(x265amd_lowres_create(m_ctx, src.plane_id, &l.plane_id));
        intptr_t stride = 0;
        check(x265amd_lowres_geometry(m_ctx, l.plane_id, &l.width, &l.lines, &stride, &l.widthInCU, &l.heightInCU));
        l.lumaStride = (int)stride;
        l.height = l.lines;
        l.marginX = src.marginX;
        l.marginY = src.marginY;
        return l;
    }
    // Lowres::init for the next picture held by an existing set
    void initLowres(const Lowres &l, const ReferencePlanes &src) { check(x265amd_lowres_update(m_ctx, l.plane_id, src.plane_id)); }

    void check(int rc) const
    {
        if (rc)
            throw std::runtime_error(std::string("x265_amd: ") + x265amd_last_error(m_ctx));
    }

private:
    x265amd_ctx *m_ctx = nullptr;
};

// TComDataCU::clipMv (Lib/TLibCommon/TComDataCU.cpp:2724-2736) for a CU at (cuX, cuY) of a picW x picH picture
// and TEncSearch::xSetSearchRange (Lib/TLibEncoder/TEncSearch.cpp:3007-3024, without the frame-parallel row
// clamp): the two rules a call site applies between the AMVP predictor and motionEstimate().
inline MV clipMv(MV mv, int picW, int picH, int cuX, int cuY, int maxCU = 64)
{
    int xmax = (picW + 8 - cuX - 1) << 2, xmin = (-maxCU - 8 - cuX + 1) << 2;
    int ymax = (picH + 8 - cuY - 1) << 2, ymin = (-maxCU - 8 - cuY + 1) << 2;
    int x = mv.x > xmax ? xmax : mv.x, y = mv.y > ymax ? ymax : mv.y;
    return MV((int16_t)(x < xmin ? xmin : x), (int16_t)(y < ymin ? ymin : y));
}
inline void setSearchRange(MV mvp, int merange, int picW, int picH, int cuX, int cuY, MV &mvmin, MV &mvmax)
{
    MV c = clipMv(mvp, picW, picH, cuX, cuY);
    int16_t d = (int16_t)(merange << 2);
    mvmin = clipMv(MV((int16_t)(c.x - d), (int16_t)(c.y - d)), picW, picH, cuX, cuY) >> 2;
    mvmax = clipMv(MV((int16_t)(c.x + d), (int16_t)(c.y + d)), picW, picH, cuX, cuY) >> 2;
}

// The scaling xGetColMVP applies to a co-located MV (Lib/TLibCommon/TComDataCU.cpp:3047-3075: xGetDistScaleFactor :3079-3096,
// scaleMv :45-51): what a caller runs over a coded picture's motion, one MV per 16x16 block, before handing it to
// x265amd_field_me_multi_tmvp as the source of the temporal AMVP candidate.
inline MV scaleColMv(MV mv, int curPOC, int curRefPOC, int colPOC, int colRefPOC)
{
    const int d = colPOC - colRefPOC, b = curPOC - curRefPOC;
    if (d == b)
        return mv;
    const int tdb = b < -128 ? -128 : b > 127 ? 127 : b, tdd = d < -128 ? -128 : d > 127 ? 127 : d;
    const int x = (0x4000 + (tdd / 2 < 0 ? -(tdd / 2) : tdd / 2)) / tdd;
    int scale = (tdb * x + 32) >> 6;
    scale = scale < -4096 ? -4096 : scale > 4095 ? 4095 : scale;
    const int vx = (scale * mv.x + 127 + (scale * mv.x < 0)) >> 8, vy = (scale * mv.y + 127 + (scale * mv.y < 0)) >> 8;
    return MV((int16_t)(vx < -32768 ? -32768 : vx > 32767 ? 32767 : vx), (int16_t)(vy < -32768 ? -32768 : vy > 32767 ? 32767 : vy));
}

// A motion field computed AHEAD of the CU analysis: the searches TEncSearch::predInterSearch is
// expected to ask for (Lib/TLibEncoder/TEncSearch.cpp:2392-2451: one motionEstimate per PU, list and
// reference, with a predictor the caller can often foresee, e.g. from the lookahead's half-resolution
// MVs) are described once, run as ONE batch, and served from memory when the call site asks.  A
// request is a hit only if every argument equals the planned one, so a hit returns exactly what the
// call would have computed; anything else is a miss and is searched on the spot.
class MotionField
{
public:
    explicit MotionField(Context &ctx) : m_ctx(ctx) {}

    // plan one search; `job` as MotionEstimate::makeJob builds it (ref_idx is assigned here)
    int add(const ReferencePlanes &ref, x265amd_me_job job)
    {
        int slot = -1;
        for (size_t i = 0; i < m_refs.size(); i++)
            if (m_refs[i] == ref.plane_id)
                slot = (int)i;
        if (slot < 0)
        {
            if (m_refs.size() >= X265AMD_MAX_REFS)
                throw std::runtime_error("x265_amd: too many reference planes in one motion field");
            slot = (int)m_refs.size();
            m_refs.push_back(ref.plane_id);
        }
        job.ref_idx = (uint8_t)slot;
        m_index[key(ref.plane_id, job)] = (int)m_jobs.size();
        m_jobs.push_back(job);
        return (int)m_jobs.size() - 1;
    }
    // search everything planned against the current picture `enc`
    void run(const ReferencePlanes &enc)
    {
        m_results.resize(m_jobs.size());
        m_enc = enc.plane_id;
        if (!m_jobs.empty())
            m_ctx.check(x265amd_me_search_batch(m_ctx.get(), enc.plane_id, m_refs.data(), (int)m_refs.size(),
                                                m_jobs.data(), (int)m_jobs.size(), m_results.data()));
        m_valid = true;
    }
    // The whole-frame field with the encoder's own predictor chain (x265amd_field_me): every pu x pu PU of the
    // picture searched in raster order on the device, each centred on the AMVP predictor derived from its
    // neighbours' results.  The searches are entered into this field under the exact arguments
    // predInterSearch would pass (predictor, xSetSearchRange bounds, the three neighbour candidates), so a
    // call site whose neighbours carry the field's MVs - every PU coded inter with its ME result - hits.
    void runChained(const ReferencePlanes &enc, const ReferencePlanes &ref, int pu, int merange, int method, int subme, int qp)
    {
        runChained(enc, std::vector<ReferencePlanes>(1, ref), pu, merange, method, subme, qp);
    }
    // The same against several references (the loop over lists and references around TEncSearch.cpp:2425-2451) in
    // ONE device call: a field's time is its chain of dependent PUs, so the fields run side by side.
    void runChained(const ReferencePlanes &enc, const std::vector<ReferencePlanes> &refs, int pu, int merange, int method, int subme, int qp)
    {
        const int gw = enc.width / pu, gh = enc.height / pu, nf = (int)refs.size();
        const size_t per = (size_t)gw * gh;
        std::vector<x265amd_me_result> all(per * nf);
        std::vector<int16_t> allmvp(per * nf * 2);
        std::vector<int> encs(nf, enc.plane_id), refids(nf);
        for (int f = 0; f < nf; f++)
            refids[f] = refs[f].plane_id;
        m_ctx.check(x265amd_field_me_batch(m_ctx.get(), nf, encs.data(), refids.data(), pu, pu, merange, method, subme, qp,
                                           all.data(), allmvp.data()));
        if (m_valid && m_enc != enc.plane_id)
            clear();
        for (int f = 0; f < nf; f++)
        {
            const x265amd_me_result *res = all.data() + per * f;
            const int16_t *mvp = allmvp.data() + per * f * 2;
            for (int gy = 0; gy < gh; gy++)
                for (int gx = 0; gx < gw; gx++)
                {
                    const int xy = gx + gy * gw;
                    x265amd_me_job j = x265amd_me_job();
                    j.pu_x = gx * pu; j.pu_y = gy * pu; j.w = j.h = (int16_t)pu;
                    MV p(mvp[2 * xy], mvp[2 * xy + 1]), lo, hi;
                    setSearchRange(p, merange, enc.width, enc.height, j.pu_x, j.pu_y, lo, hi);
                    j.mvmin_x = lo.x; j.mvmin_y = lo.y; j.mvmax_x = hi.x; j.mvmax_y = hi.y;
                    j.qmvp_x = p.x; j.qmvp_y = p.y;
                    j.merange = (int16_t)merange; j.method = (uint8_t)method; j.subme = (uint8_t)subme; j.qp = (uint8_t)qp;
                    j.numc = 3; j.icost = X265AMD_SAD;
                    if (gx > 0)                { j.mvc[0][0] = res[xy - 1].mvx;      j.mvc[0][1] = res[xy - 1].mvy; }
                    if (gy > 0)                { j.mvc[1][0] = res[xy - gw].mvx;     j.mvc[1][1] = res[xy - gw].mvy; }
                    if (gy > 0 && gx + 1 < gw) { j.mvc[2][0] = res[xy - gw + 1].mvx; j.mvc[2][1] = res[xy - gw + 1].mvy; }
                    const int at = add(refs[f], j);
                    if ((size_t)at >= m_results.size())
                        m_results.resize(at + 1);
                    m_results[at] = res[xy];
                }
        }
        m_enc = enc.plane_id;
        m_valid = true;
    }
    // One field per entry of `shapes` - a PU size and a partition mode each (x265amd_field_me_multi): the 2Nx2N fields of
    // several quadtree depths and the 2NxN / Nx2N and asymmetric fields beside them, searched side by side in ONE device call
    // and entered under the arguments predInterSearch would pass for each part (bounds from the CU's position).
    // part_mode is a PartSize (Lib/TLibCommon/TypeDef.h:100-111): 0 2Nx2N, 1 2NxN (pu = the part), 2 Nx2N (pu = the part),
    // 4 2NxnU, 5 2NxnD, 6 nLx2N, 7 nRx2N (pu = the CU, square, 16 / 32 / 64).
    struct Shape { int pu_w, pu_h, part_mode; };
    static bool upperLower(int pm) { return pm == 1 || pm == 4 || pm == 5; }
    static void shapeGrid(const ReferencePlanes &enc, const Shape &s, int &gw, int &gh)
    {
        gw = enc.width / s.pu_w; gh = enc.height / s.pu_h;
        if (s.part_mode == 1) gh &= ~1;
        if (s.part_mode == 2) gw &= ~1;
        if (s.part_mode == 4 || s.part_mode == 5) gh *= 2;
        if (s.part_mode == 6 || s.part_mode == 7) gw *= 2;
    }
    // PU (gx, gy) of a shape's grid: its rectangle and the position of its CU
    struct PuRect { int x, y, w, h, cuX, cuY; bool lower; };
    static PuRect shapePu(const Shape &s, int gx, int gy)
    {
        PuRect r;
        const int pm = s.part_mode;
        r.lower = upperLower(pm) && (gy & 1);
        r.x = gx * s.pu_w; r.y = gy * s.pu_h; r.w = s.pu_w; r.h = s.pu_h;
        r.cuX = pm == 2 ? (gx & ~1) * s.pu_w : r.x;
        r.cuY = pm == 1 && r.lower ? r.y - s.pu_h : r.y;
        if (pm == 4 || pm == 5)
        {
            const int cut = pm == 4 ? s.pu_h >> 2 : s.pu_h - (s.pu_h >> 2);
            r.cuY = (gy >> 1) * s.pu_h;
            r.y = r.cuY + (r.lower ? cut : 0);
            r.h = r.lower ? s.pu_h - cut : cut;
        }
        if (pm == 6 || pm == 7)
        {
            const int cut = pm == 6 ? s.pu_w >> 2 : s.pu_w - (s.pu_w >> 2);
            r.cuX = (gx >> 1) * s.pu_w;
            r.x = r.cuX + ((gx & 1) ? cut : 0);
            r.w = (gx & 1) ? s.pu_w - cut : cut;
        }
        return r;
    }
    // col (optional): the co-located motion for the temporal AMVP candidate, per shape one grid of ((width + 15) / 16) x
    // ((height + 15) / 16) entries (one per 16x16 block, scaled with scaleColMv, cost < 0 = no motion), one grid after the other.
    void runChained(const ReferencePlanes &enc, const ReferencePlanes &ref, const std::vector<Shape> &shapes, int merange, int method,
                    int subme, int qp, const std::vector<x265amd_me_result> *col = nullptr)
    {
        const int nf = (int)shapes.size();
        std::vector<int> encs(nf, enc.plane_id), refids(nf, ref.plane_id), pw(nf), ph(nf), pm(nf);
        std::vector<size_t> first(nf + 1, 0);
        for (int f = 0; f < nf; f++)
        {
            int gw, gh;
            shapeGrid(enc, shapes[f], gw, gh);
            pw[f] = shapes[f].pu_w; ph[f] = shapes[f].pu_h; pm[f] = shapes[f].part_mode;
            first[f + 1] = first[f] + (size_t)gw * gh;
        }
        std::vector<x265amd_me_result> all(first[nf]);
        std::vector<int16_t> allmvp(first[nf] * 2);
        if (col && col->size() != (size_t)nf * ((enc.width + 15) / 16) * ((enc.height + 15) / 16))
            throw std::invalid_argument("runChained: col needs one entry per 16x16 block and shape");
        m_ctx.check(x265amd_field_me_multi_tmvp(m_ctx.get(), nf, encs.data(), refids.data(), pw.data(), ph.data(), pm.data(),
                                                col ? col->data() : nullptr, merange, method, subme, qp, all.data(), allmvp.data()));
        if (m_valid && m_enc != enc.plane_id)
            clear();
        for (int f = 0; f < nf; f++)
        {
            const Shape &sh = shapes[f];
            int gw, gh;
            shapeGrid(enc, sh, gw, gh);
            const x265amd_me_result *res = all.data() + first[f];
            const int16_t *mvp = allmvp.data() + first[f] * 2;
            for (int gy = 0; gy < gh; gy++)
                for (int gx = 0; gx < gw; gx++)
                {
                    const int xy = gx + gy * gw;
                    const PuRect pu = shapePu(sh, gx, gy);
                    const bool lower = pu.lower;
                    x265amd_me_job j = x265amd_me_job();
                    j.pu_x = pu.x; j.pu_y = pu.y; j.w = (int16_t)pu.w; j.h = (int16_t)pu.h;
                    const int cuX = pu.cuX, cuY = pu.cuY;
                    MV p(mvp[2 * xy], mvp[2 * xy + 1]), lo, hi;
                    setSearchRange(p, merange, enc.width, enc.height, cuX, cuY, lo, hi);
                    j.mvmin_x = lo.x; j.mvmin_y = lo.y; j.mvmax_x = hi.x; j.mvmax_y = hi.y;
                    j.qmvp_x = p.x; j.qmvp_y = p.y;
                    j.merange = (int16_t)merange; j.method = (uint8_t)method; j.subme = (uint8_t)subme; j.qp = (uint8_t)qp;
                    j.numc = 3; j.icost = X265AMD_SAD;
                    if (gx > 0)                          { j.mvc[0][0] = res[xy - 1].mvx;      j.mvc[0][1] = res[xy - 1].mvy; }
                    if (gy > 0)                          { j.mvc[1][0] = res[xy - gw].mvx;     j.mvc[1][1] = res[xy - gw].mvy; }
                    if (!lower && gy > 0 && gx + 1 < gw) { j.mvc[2][0] = res[xy - gw + 1].mvx; j.mvc[2][1] = res[xy - gw + 1].mvy; }
                    const int at = add(ref, j);
                    if ((size_t)at >= m_results.size())
                        m_results.resize(at + 1);
                    m_results[at] = res[xy];
                }
        }
        m_enc = enc.plane_id;
        m_valid = true;
    }
    // the stored result of a request identical to a planned one
    bool lookup(const ReferencePlanes &enc, const ReferencePlanes &ref, x265amd_me_job want, x265amd_me_result &out)
    {
        if (m_valid && enc.plane_id == m_enc)
        {
            std::map<Key, int>::const_iterator it = m_index.find(key(ref.plane_id, want));
            if (it != m_index.end())
            {
                want.ref_idx = m_jobs[it->second].ref_idx;
                if (!memcmp(&want, &m_jobs[it->second], sizeof(want)))
                {
                    out = m_results[it->second];
                    m_hits++;
                    return true;
                }
            }
        }
        m_misses++;
        return false;
    }
    void clear() { m_jobs.clear(); m_refs.clear(); m_index.clear(); m_results.clear(); m_valid = false; }
    size_t size() const { return m_jobs.size(); }
    unsigned hits() const { return m_hits; }
    unsigned misses() const { return m_misses; }
    const std::vector<x265amd_me_result> &results() const { return m_results; }

private:
    typedef std::tuple<int, int, int, int, int> Key;       // reference plane, PU x, y, w, h
    static Key key(int ref, const x265amd_me_job &j) { return Key(ref, j.pu_x, j.pu_y, j.w, j.h); }

    Context &m_ctx;
    std::vector<x265amd_me_job> m_jobs;
    std::vector<int> m_refs;
    std::vector<x265amd_me_result> m_results;
    std::map<Key, int> m_index;
    int m_enc = -1;
    bool m_valid = false;
    unsigned m_hits = 0, m_misses = 0;
};

class MotionEstimate
{
public:
    explicit MotionEstimate(Context &ctx) : m_ctx(ctx) {}

    // consult `field` (may be NULL) before searching: see MotionField
    void attachField(MotionField *field) { m_field = field; }
    // the job the next motionEstimate() with these arguments would run (for MotionField::add)
    x265amd_me_job plan(const MV &mvmin, const MV &mvmax, const MV &qmvp, int numCandidates, const MV *mvc, int merange)
    {
        return makeJob(mvmin, mvmax, qmvp, numCandidates, mvc, merange);
    }

    void setSearchMethod(int i) { m_method = i; }                 // motion.h:72
    void setSubpelRefine(int i) { m_subme = i; }                  // motion.h:74
    void setQP(unsigned qp) { m_qp = qp; }                        // bitcost.h:41
    void setSourcePlane(const ReferencePlanes &enc) { m_enc = enc; }   // motion.h:78-82
    void setSourcePU(int offset, int pwidth, int pheight)         // motion.h:84: offset = y*stride + x
    {
        // y = floor(offset / stride) (PUs in the top margin have negative offsets), x in [0, stride).
        // A column at or beyond width + marginX does not exist in row y: it is a column of the LEFT
        // margin of row y + 1 (x - stride in [-marginX, 0)).
        int y = offset >= 0 ? offset / m_enc.lumaStride : -((-offset + m_enc.lumaStride - 1) / m_enc.lumaStride);
        int x = offset - y * m_enc.lumaStride;
        if (x >= m_enc.width + m_enc.marginX) { x -= m_enc.lumaStride; y += 1; }
        m_puX = x; m_puY = y; m_w = pwidth; m_h = pheight;
    }
    void setSourcePUxy(int x, int y, int pwidth, int pheight) { m_puX = x; m_puY = y; m_w = pwidth; m_h = pheight; }

    // bufSAD / bufSATD / bufSA8D (motion.h:89-93): cost of the current PU against a block of `ref`
    int bufSAD(const ReferencePlanes &ref, int x, int y)  { return bufCost(X265AMD_SAD, ref, x, y); }
    int bufSATD(const ReferencePlanes &ref, int x, int y) { return bufCost(X265AMD_SATD, ref, x, y); }
    int bufSA8D(const ReferencePlanes &ref, int x, int y) { return bufCost(X265AMD_SA8D, ref, x, y); }

    // cost of the current PU against the motion-compensated block at q-pel `mv` of `ref`: the distortion term
    // of xGetTemplateCost (kind SAD, TEncSearch.cpp:2944-2958) and of the merge / inter-prediction error checks
    // (kind SA8D, :2182-2190); `n` candidates in one call
    void mcCost(int kind, const ReferencePlanes &ref, const MV *mv, int n, int32_t *out)
    {
        std::vector<x265amd_mc_pair> p(n);
        for (int i = 0; i < n; i++)
        {
            x265amd_mc_pair q = { m_enc.plane_id, m_puX, m_puY, ref.plane_id, mv[i].x, mv[i].y };
            p[i] = q;
        }
        m_ctx.check(x265amd_mc_cost_batch(m_ctx.get(), kind, m_w, m_h, n, p.data(), out));
    }

    // motion.h:95 -- synchronous, one job
    int motionEstimate(const ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &qmvp,
                       int numCandidates, const MV *mvc, int merange, MV &outQMv)
    {
        x265amd_me_job j = makeJob(mvmin, mvmax, qmvp, numCandidates, mvc, merange);
        x265amd_me_result r;
        int refs[1] = { ref->plane_id };
        if (!m_field || !m_field->lookup(m_enc, *ref, j, r))
            m_ctx.check(x265amd_me_search_batch(m_ctx.get(), m_enc.plane_id, refs, 1, &j, 1, &r));
        outQMv = MV(r.mvx, r.mvy);
        return r.cost;
    }

    // The search of TEncSearch::xMotionEstimation (bidirectional refinement, TEncSearch.cpp:2993-3000):
    // xPatternSearch over [mvmin, mvmax] + xPatternRefinement at half- and quarter-pel.  The source
    // plane set by setSourcePlane must hold the modified block (2*org - otherPred) at the PU position.
    // Returns the quarter-pel refinement's cost (outCost of TEncSearch.cpp:3000); outQMv in q-pel.
    int bipredSearch(const ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &mvp, MV &outQMv)
    {
        x265amd_me_job j = makeJob(mvmin, mvmax, mvp, 0, NULL, 0);
        j.method = X265AMD_BIPRED;
        j.subme = 0;
        x265amd_me_result r;
        int refs[1] = { ref->plane_id };
        m_ctx.check(x265amd_me_search_batch(m_ctx.get(), m_enc.plane_id, refs, 1, &j, 1, &r));
        outQMv = MV(r.mvx, r.mvy);
        return r.cost;
    }

    // batched form: same arguments, result delivered by flush(); returns the job's index
    int queue(const ReferencePlanes *ref, const MV &mvmin, const MV &mvmax, const MV &qmvp,
              int numCandidates, const MV *mvc, int merange)
    {
        x265amd_me_job j = makeJob(mvmin, mvmax, qmvp, numCandidates, mvc, merange);
        int slot = -1;
        for (size_t i = 0; i < m_refs.size(); i++)
            if (m_refs[i] == ref->plane_id)
                slot = (int)i;
        if (slot < 0)
        {
            if (m_refs.size() >= X265AMD_MAX_REFS)
                throw std::runtime_error("x265_amd: too many reference planes in one batch");
            slot = (int)m_refs.size();
            m_refs.push_back(ref->plane_id);
        }
        j.ref_idx = (uint8_t)slot;
        m_jobs.push_back(j);
        return (int)m_jobs.size() - 1;
    }
    // run everything queued since the last flush (all against the current source plane)
    const std::vector<x265amd_me_result> &flush()
    {
        m_results.resize(m_jobs.size());
        if (!m_jobs.empty())
            m_ctx.check(x265amd_me_search_batch(m_ctx.get(), m_enc.plane_id, m_refs.data(), (int)m_refs.size(),
                                                m_jobs.data(), (int)m_jobs.size(), m_results.data()));
        m_jobs.clear();
        m_refs.clear();
        return m_results;
    }

    // BitCost::bitcost(mv) relative to the last predictor (bitcost.h:49-53)
    int bitcost(const MV &mv) const { return x265amd_bitcost(m_ctx.get(), m_mvp.x, m_mvp.y, mv.x, mv.y); }

private:
    int bufCost(int kind, const ReferencePlanes &ref, int x, int y)
    {
        x265amd_block_pair p = { m_enc.plane_id, m_puX, m_puY, ref.plane_id, x, y };
        int32_t out;
        m_ctx.check(x265amd_cost_batch(m_ctx.get(), kind, m_w, m_h, 1, &p, &out));
        return out;
    }
    x265amd_me_job makeJob(const MV &mvmin, const MV &mvmax, const MV &qmvp, int numc, const MV *mvc, int merange)
    {
        x265amd_me_job j = x265amd_me_job();
        j.pu_x = m_puX; j.pu_y = m_puY; j.w = (int16_t)m_w; j.h = (int16_t)m_h;
        j.mvmin_x = mvmin.x; j.mvmin_y = mvmin.y; j.mvmax_x = mvmax.x; j.mvmax_y = mvmax.y;
        j.qmvp_x = qmvp.x; j.qmvp_y = qmvp.y;
        j.merange = (int16_t)merange;
        j.method = (uint8_t)m_method; j.subme = (uint8_t)m_subme; j.qp = (uint8_t)m_qp;
        j.numc = (uint8_t)(numc > X265AMD_MAX_MVC ? X265AMD_MAX_MVC : numc);
        for (int i = 0; i < j.numc; i++) { j.mvc[i][0] = mvc[i].x; j.mvc[i][1] = mvc[i].y; }
        j.icost = X265AMD_SAD;
        m_mvp = qmvp;
        return j;
    }

    Context &m_ctx;
    MotionField *m_field = nullptr;
    ReferencePlanes m_enc;
    int m_method = 3, m_subme = 5;          // the reference's defaults (motion.cpp:82-85)
    unsigned m_qp = 32;
    int m_puX = 0, m_puY = 0, m_w = 0, m_h = 0;
    MV m_mvp;
    std::vector<x265amd_me_job> m_jobs;
    std::vector<int> m_refs;
    std::vector<x265amd_me_result> m_results;
};

// The motion-search part of Lookahead::estimateFrameCost / estimateCUCost (encoder/slicetype.cpp:283-410)
// with the reference's lookahead settings (slicetype.cpp:70-73).
class Lookahead
{
public:
    explicit Lookahead(Context &ctx, int qp = 12) : m_ctx(ctx), m_qp(qp) {}

    // Search frame `fenc` against `fref` for every listed pair; field[pair * cuCount + cux + cuy * widthInCU]
    // holds what the reference stores in lowresMvs / lowresMvCosts (slicetype.cpp:407).
    const std::vector<x265amd_me_result> &estimateFrameMotion(const Lowres *const *fenc, const Lowres *const *fref, int pairs)
    {
        std::vector<int> e(pairs), r(pairs);
        for (int i = 0; i < pairs; i++)
        {
            e[i] = fenc[i]->plane_id;
            r[i] = fref[i]->plane_id;
        }
        m_field.resize(pairs ? (size_t)pairs * fenc[0]->widthInCU * fenc[0]->heightInCU : 0);
        if (pairs)
            m_ctx.check(x265amd_lookahead_me(m_ctx.get(), pairs, e.data(), r.data(), merange, X265AMD_HEX, 1, m_qp, m_field.data()));
        return m_field;
    }

    int merange = 16;

private:
    Context &m_ctx;
    int m_qp;
    std::vector<x265amd_me_result> m_field;
};

// Overwrite the ME-related slots of a primitive table laid out like the reference's
// EncoderPrimitives (sad/sad_x3/sad_x4/satd/sa8d_inter [64], sa8d[5]) the way
// Setup_Vector_Primitives / Setup_Assembly_Primitives do (common/primitives.cpp:129-139).
// `Table` is the encoder's struct; its slot types only differ from ours in `pixel*` vs `void*`.
template <class Table>
int Setup_HIP_Primitives(Table &p, int device, int bytesPerPixel)
{
    static x265amd_primitives t;
    int rc = x265amd_setup_primitives(&t, device, bytesPerPixel);
    if (rc)
        return rc;
    for (int i = 0; i < 64; i++)
    {
        p.sad[i] = reinterpret_cast<decltype(p.sad[i])>(t.sad[i]);
        p.sad_x3[i] = reinterpret_cast<decltype(p.sad_x3[i])>(t.sad_x3[i]);
        p.sad_x4[i] = reinterpret_cast<decltype(p.sad_x4[i])>(t.sad_x4[i]);
        p.satd[i] = reinterpret_cast<decltype(p.satd[i])>(t.satd[i]);
        p.sa8d_inter[i] = reinterpret_cast<decltype(p.sa8d_inter[i])>(t.sa8d_inter[i]);
    }
    for (int i = 0; i < 5; i++)
        p.sa8d[i] = reinterpret_cast<decltype(p.sa8d[i])>(t.sa8d[i]);
    return 0;
}

} // namespace x265amd
