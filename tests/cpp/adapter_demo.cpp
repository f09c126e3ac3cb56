// Drives the C++ adapter the way an encoder call site would and prints the results as text:
//   adapter_demo <width> <height> <margin> <enc.raw> <ref.raw>
// Planes are padded 8-bit luma (stride = width + 2*margin).  Used by tests/test_adapter_cpp.py.
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../../x265_amd/host/x265_amd_host.hpp"

typedef unsigned char pixel;
// a primitive table with the reference's slot types (common/primitives.h:180-184, 240-250)
typedef int  (*pixelcmp_t)(pixel *fenc, intptr_t fencstride, pixel *fref, intptr_t frefstride);
typedef void (*pixelcmp_x4_t)(pixel *fenc, pixel *fref0, pixel *fref1, pixel *fref2, pixel *fref3, intptr_t frefstride, int *res);
typedef void (*pixelcmp_x3_t)(pixel *fenc, pixel *fref0, pixel *fref1, pixel *fref2, intptr_t frefstride, int *res);
struct EncoderPrimitives
{
    pixelcmp_t sad[64];
    pixelcmp_x3_t sad_x3[64];
    pixelcmp_x4_t sad_x4[64];
    pixelcmp_t satd[64];
    pixelcmp_t sa8d_inter[64];
    pixelcmp_t sa8d[5];
};

static std::vector<pixel> slurp(const char *path, size_t n)
{
    std::vector<pixel> v(n);
    FILE *f = fopen(path, "rb");
    if (!f || fread(v.data(), 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", path); exit(2); }
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage\n"); return 2; }
    int W = atoi(argv[1]), H = atoi(argv[2]), M = atoi(argv[3]);
    int stride = W + 2 * M;
    size_t n = (size_t)stride * (H + 2 * M);
    std::vector<pixel> enc = slurp(argv[4], n), ref = slurp(argv[5], n);
    try
    {
        x265amd::Context ctx(0);
        x265amd::ReferencePlanes pe = ctx.uploadPlane(enc.data(), W, H, M, M, stride, 1, 8);
        x265amd::ReferencePlanes pr = ctx.uploadPlane(ref.data(), W, H, M, M, stride, 1, 8);
        x265amd::MotionEstimate me(ctx);
        me.setSourcePlane(pe);
        me.setQP(30);
        // (1) the reference's per-PU call sequence, all five methods
        for (int method = 0; method < 5; method++)
        {
            me.setSearchMethod(method);
            me.setSubpelRefine(5);
            me.setSourcePU(32 * stride + 48, 16, 16);
            x265amd::MV mvc[2] = { x265amd::MV(8, -4), x265amd::MV(-12, 6) }, out;
            int cost = me.motionEstimate(&pr, x265amd::MV(-16, -16), x265amd::MV(16, 16), x265amd::MV(5, -3), 2, mvc, 16, out);
            printf("single %d %d %d %d\n", method, out.x, out.y, cost);
        }
        // (1b) linear PU offsets (the reference's setSourcePU argument) all over the padded plane: right half,
        //      flush with the right edge, inside the left / right / top margins; each against the (x, y) form
        {
            me.setSearchMethod(1);
            me.setSubpelRefine(2);
            const int px[6] = { W - 48, W - 16, -8, W + 8 - 16, 160, 16 }, py[6] = { 40, 64, 24, 8, -8, H - 16 };
            for (int k = 0; k < 6; k++)
            {
                x265amd::MV a, b;
                me.setSourcePU(py[k] * stride + px[k], 16, 16);
                int ca = me.motionEstimate(&pr, x265amd::MV(-8, -8), x265amd::MV(8, 8), x265amd::MV(2, 1), 0, NULL, 8, a);
                me.setSourcePUxy(px[k], py[k], 16, 16);
                int cb = me.motionEstimate(&pr, x265amd::MV(-8, -8), x265amd::MV(8, 8), x265amd::MV(2, 1), 0, NULL, 8, b);
                printf("srcpu %d %d %d %d %d %d %d %d\n", px[k], py[k], a.x, a.y, ca, b.x, b.y, cb);
            }
        }
        // (2) batched: one row of 16x16 PUs queued, then flushed
        me.setSearchMethod(1);
        me.setSubpelRefine(2);
        int cnt = 0;
        for (int x = 0; x + 16 <= W; x += 16, cnt++)
        {
            me.setSourcePUxy(x, 16, 16, 16);
            me.queue(&pr, x265amd::MV(-24, -16), x265amd::MV(24, 16), x265amd::MV(0, 0), 0, NULL, 24);
        }
        const std::vector<x265amd_me_result> &r = me.flush();
        for (int i = 0; i < cnt; i++)
            printf("batch %d %d %d %d\n", i, r[i].mvx, r[i].mvy, r[i].cost);
        // (3) bufSAD / bufSATD / bufSA8D
        me.setSourcePUxy(64, 32, 32, 32);
        printf("buf %d %d %d\n", me.bufSAD(pr, 60, 35), me.bufSATD(pr, 60, 35), me.bufSA8D(pr, 60, 35));
        // (4) the primitive table hook
        EncoderPrimitives prim;
        if (x265amd::Setup_HIP_Primitives(prim, 0, 1)) { fprintf(stderr, "setup primitives failed\n"); return 3; }
        pixel *e0 = enc.data() + (size_t)M * stride + M, *r0 = ref.data() + (size_t)M * stride + M;
        int res[4];
        prim.sad_x4[27](e0, r0, r0 + 1, r0 + 2, r0 + 3, stride, res);   // PARTITION_16x16 == 27; fenc stride is 64 by contract
        printf("prim %d %d %d %d %d\n", prim.sad[27](e0, stride, r0 + 5, stride), prim.satd[27](e0, stride, r0 + 5, stride),
               prim.sa8d_inter[27](e0, stride, r0 + 5, stride), res[0], res[3]);
        // (4b) bidirectional refinement: +-4 around (3,-2), predictor (9,-7), PU (96,64) 32x16
        {
            me.setSourcePUxy(96, 64, 32, 16);
            x265amd::MV out;
            int cost = me.bipredSearch(&pr, x265amd::MV(-1, -6), x265amd::MV(7, 2), x265amd::MV(9, -7), out);
            printf("bipred %d %d %d\n", out.x, out.y, cost);
        }
        // (5) lookahead: half-resolution sets of both pictures, one whole-frame search, a few CUs printed
        if (M >= 16)
        {
            x265amd::Lowres le = ctx.createLowres(pe), lf = ctx.createLowres(pr);
            x265amd::Lookahead look(ctx);
            const x265amd::Lowres *fe[1] = { &le }, *ff[1] = { &lf };
            const std::vector<x265amd_me_result> &f = look.estimateFrameMotion(fe, ff, 1);
            printf("look %d %d %d\n", le.widthInCU, le.heightInCU, (int)f.size());
            for (size_t i = 0; i < f.size(); i++)
                printf("cu %d %d %d %d\n", (int)i, f[i].mvx, f[i].mvy, f[i].cost);
            ctx.releasePlane(le);
            ctx.releasePlane(lf);
        }
        // (6) motion field planned ahead of the call sites: a row of 16x16 PUs with foreseen predictors;
        //     identical requests are served from the field, a request with another predictor is searched
        {
            x265amd::MotionField field(ctx);
            me.setSearchMethod(3);
            me.setSubpelRefine(3);
            for (int x = 0; x + 16 <= W; x += 16)
            {
                me.setSourcePUxy(x, 48, 16, 16);
                x265amd::MV c[1] = { x265amd::MV((int16_t)(x / 8), 2) };
                field.add(pr, me.plan(x265amd::MV(-20, -20), x265amd::MV(20, 20), x265amd::MV(4, (int16_t)-(x / 16)), 1, c, 20));
            }
            field.run(pe);
            me.attachField(&field);
            for (int x = 0; x + 16 <= W; x += 16)
            {
                me.setSourcePUxy(x, 48, 16, 16);
                x265amd::MV c[1] = { x265amd::MV((int16_t)(x / 8), 2) }, out;
                bool foreseen = (x / 16) % 3 != 2;       // every third request arrives with another predictor
                x265amd::MV mvp(4, (int16_t)(foreseen ? -(x / 16) : 7));
                int cost = me.motionEstimate(&pr, x265amd::MV(-20, -20), x265amd::MV(20, 20), mvp, 1, c, 20, out);
                printf("field %d %d %d %d %d\n", x, foreseen ? 1 : 0, out.x, out.y, cost);
            }
            printf("fieldstats %u %u %d\n", field.hits(), field.misses(), (int)field.size());
            me.attachField(NULL);
        }
        // (7) the chained field: the device searches every 16x16 PU of the picture with the encoder's predictor chain;
        //     then an "encoder" walks the PUs in coding order, derives each AMVP predictor from ITS OWN decisions
        //     (template costs through mcCost, as xEstimateMvPredAMVP does) and calls motionEstimate at the unchanged
        //     call site.  One PU (the 46th) is "coded intra" (its MV is zero for its neighbours), so the chains differ
        //     around it and the requests that depend on it miss (and are searched on the spot); everything else is served from the field.
        {
            const int pu = 16, merange = 32, gw = W / pu, gh = H / pu;
            x265amd::MotionField field(ctx);
            me.setSearchMethod(1);
            me.setSubpelRefine(5);
            me.setQP(30);
            // (two references in one device call; the walk below asks for the first one only)
            field.runChained(pe, std::vector<x265amd::ReferencePlanes>{ pr, pe }, pu, merange, 1, 5, 30);
            me.attachField(&field);
            std::vector<x265amd::MV> dec(gw * gh);
            for (int gy = 0; gy < gh; gy++)
                for (int gx = 0; gx < gw; gx++)
                {
                    const int xy = gx + gy * gw;
                    me.setSourcePUxy(gx * pu, gy * pu, pu, pu);
                    const bool haveL = gx > 0, haveA = gy > 0, haveAR = gy > 0 && gx + 1 < gw, haveAL = gy > 0 && gx > 0;
                    x265amd::MV pred[3], cand[2], b;
                    if (haveL) pred[0] = dec[xy - 1];
                    if (haveA) pred[1] = dec[xy - gw];
                    if (haveAR) pred[2] = dec[xy - gw + 1];
                    const bool haveB = haveAR || haveA || haveAL;
                    if (haveAR) b = dec[xy - gw + 1]; else if (haveA) b = dec[xy - gw]; else if (haveAL) b = dec[xy - gw - 1];
                    if (haveL) { cand[0] = dec[xy - 1]; if (haveB && b != cand[0]) cand[1] = b; }
                    else if (haveB) cand[0] = b;
                    x265amd::MV mvp = cand[0];
                    if (cand[1] != cand[0])
                    {
                        x265amd::MV cl[2] = { x265amd::clipMv(cand[0], W, H, gx * pu, gy * pu), x265amd::clipMv(cand[1], W, H, gx * pu, gy * pu) };
                        int32_t t[2];
                        me.mcCost(X265AMD_SAD, pr, cl, 2, t);
                        if (t[1] < t[0]) mvp = cand[1];
                    }
                    x265amd::MV lo, hi, out;
                    x265amd::setSearchRange(mvp, merange, W, H, gx * pu, gy * pu, lo, hi);
                    int cost = me.motionEstimate(&pr, lo, hi, mvp, 3, pred, merange, out);
                    const bool intra = xy == 45;
                    dec[xy] = intra ? x265amd::MV(0, 0) : out;
                    printf("chain %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", gx, gy, mvp.x, mvp.y, lo.x, lo.y, hi.x, hi.y,
                           pred[0].x, pred[0].y, pred[1].x, pred[1].y, pred[2].x, pred[2].y, out.x, out.y, cost, intra ? 1 : 0);
                }
            printf("chainstats %u %u %d\n", field.hits(), field.misses(), gw * gh);
            me.attachField(NULL);
        }
        // (8) fields of several PU sizes and partition modes from ONE device call: the 2Nx2N fields of the 16x16 and 32x32 CUs, the
        //     2NxN / Nx2N fields of the 16x16 CUs, the 2NxnU field of the 32x32 and the nRx2N field of the 16x16 CUs.  For each an "encoder" that codes every CU in that mode walks the CUs in
        //     raster order, part by part, derives the predictors from its own decisions by the rules of the partition mode and
        //     calls motionEstimate at the unchanged call site.  One part per field is "coded intra".  The AMVP lists take the
        //     temporal candidate from a co-located motion field, on the device and in the walk alike.
        {
            typedef x265amd::MotionField::Shape Shape;
            const std::vector<Shape> shapes = { Shape{ 16, 16, 0 }, Shape{ 32, 32, 0 }, Shape{ 16, 8, 1 }, Shape{ 8, 16, 2 },
                                                Shape{ 32, 32, 4 }, Shape{ 16, 16, 7 } };
            const int merange = 24;
            x265amd::MotionField field(ctx);
            me.setSearchMethod(1);
            me.setSubpelRefine(3);
            me.setQP(28);
            // The co-located motion (the temporal AMVP candidate's source): one MV per 16x16 block of a "previous picture" twice as
            // far from its reference as this one is, so every MV is scaled to half; every ninth block was "coded intra".
            const int colW = (W + 15) / 16, colH = (H + 15) / 16;
            std::vector<x265amd_me_result> colOne(colW * colH);
            unsigned lcg = 12345;
            for (size_t i = 0; i < colOne.size(); i++)
            {
                lcg = lcg * 1664525u + 1013904223u;
                const x265amd::MV m = x265amd::scaleColMv(x265amd::MV((int16_t)((lcg >> 8) % 41) - 20, (int16_t)((lcg >> 16) % 25) - 12), 2, 1, 1, -1);
                colOne[i].mvx = m.x; colOne[i].mvy = m.y; colOne[i].cost = i % 9 == 4 ? -1 : 0;
            }
            std::vector<x265amd_me_result> col;
            for (size_t f = 0; f < shapes.size(); f++)
                col.insert(col.end(), colOne.begin(), colOne.end());
            field.runChained(pe, pr, shapes, merange, 1, 3, 28, &col);
            me.attachField(&field);
            int total = 0;
            for (size_t f = 0; f < shapes.size(); f++)
            {
                const Shape &sh = shapes[f];
                int gw, gh;
                x265amd::MotionField::shapeGrid(pe, sh, gw, gh);
                const unsigned hits0 = field.hits();
                const int total0 = total;
                const bool twoRows = x265amd::MotionField::upperLower(sh.part_mode);
                const int parts = twoRows ? 2 : 1;
                std::vector<x265amd::MV> dec(gw * gh);
                for (int row = 0; row < gh / parts; row++)
                    for (int gx = 0; gx < gw; gx++)
                        for (int part = 0; part < parts; part++)
                        {
                            const int gy = row * parts + part, xy = gx + gy * gw;
                            const bool lower = part == 1;
                            const x265amd::MotionField::PuRect pu = x265amd::MotionField::shapePu(sh, gx, gy);
                            const int px = pu.x, py = pu.y, cuX = pu.cuX, cuY = pu.cuY;
                            me.setSourcePUxy(px, py, pu.w, pu.h);
                            const bool haveL = gx > 0, haveA = gy > 0, haveAR = !lower && gy > 0 && gx + 1 < gw, haveAL = gy > 0 && gx > 0;
                            x265amd::MV pred[3], cand[2], b;
                            if (haveL) pred[0] = dec[xy - 1];
                            if (haveA) pred[1] = dec[xy - gw];
                            if (haveAR) pred[2] = dec[xy - gw + 1];
                            const bool haveB = haveAR || haveA || haveAL;
                            if (haveAR) b = dec[xy - gw + 1]; else if (haveA) b = dec[xy - gw]; else if (haveAL) b = dec[xy - gw - 1];
                            // (upper part of a 2NxN / 2NxnU / 2NxnD CU: the left scan finds the below-left neighbour first)
                            const x265amd::MV a = haveL ? ((twoRows && !lower) ? dec[xy + gw - 1] : dec[xy - 1]) : x265amd::MV();
                            int numCand = 0;
                            if (haveL) { cand[numCand++] = a; if (haveB && b != cand[0]) cand[numCand++] = b; }
                            else if (haveB) cand[numCand++] = b;
                            if (numCand < 2)
                            {
                                // the temporal candidate: below-right of the PU (inside the picture and the CTU row), else its centre
                                const int rbx = px + pu.w, rby = py + pu.h;
                                const x265amd_me_result *t = NULL;
                                if (rbx < W && rby < H && (rby & 63) && colOne[(rby >> 4) * colW + (rbx >> 4)].cost >= 0)
                                    t = &colOne[(rby >> 4) * colW + (rbx >> 4)];
                                else
                                {
                                    const int ctrx = px + pu.w / 8 * 4, ctry = py + pu.h / 8 * 4;
                                    if (colOne[(ctry >> 4) * colW + (ctrx >> 4)].cost >= 0)
                                        t = &colOne[(ctry >> 4) * colW + (ctrx >> 4)];
                                }
                                if (t)
                                    cand[numCand++] = x265amd::MV(t->mvx, t->mvy);
                            }
                            x265amd::MV mvp = cand[0];
                            if (cand[1] != cand[0])
                            {
                                x265amd::MV cl[2] = { x265amd::clipMv(cand[0], W, H, cuX, cuY), x265amd::clipMv(cand[1], W, H, cuX, cuY) };
                                int32_t t[2];
                                me.mcCost(X265AMD_SAD, pr, cl, 2, t);
                                if (t[1] < t[0]) mvp = cand[1];
                            }
                            x265amd::MV lo, hi, out;
                            x265amd::setSearchRange(mvp, merange, W, H, cuX, cuY, lo, hi);
                            const unsigned hitsBefore = field.hits();
                            int cost = me.motionEstimate(&pr, lo, hi, mvp, 3, pred, merange, out);
                            const bool intra = xy == 21;
                            dec[xy] = intra ? x265amd::MV(0, 0) : out;
                            total++;
                            printf("mchain %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %u\n", pu.w, pu.h, sh.part_mode, px, py,
                                   mvp.x, mvp.y, lo.x, lo.y, hi.x, hi.y, pred[0].x, pred[0].y, pred[1].x, pred[1].y, pred[2].x, pred[2].y,
                                   out.x, out.y, cost, intra ? 1 : 0, field.hits() - hitsBefore);
                        }
                printf("mchainshape %d %d %d %u %d\n", sh.pu_w, sh.pu_h, sh.part_mode, field.hits() - hits0, total - total0);
            }
            printf("mchainstats %u %u %d\n", field.hits(), field.misses(), total);
            me.attachField(NULL);
        }
        ctx.releasePlane(pe);
        ctx.releasePlane(pr);
    }
    catch (const std::exception &e)
    {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
