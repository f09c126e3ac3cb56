/* TEST INFRASTRUCTURE ONLY -- pixel-type-generic body of the CPU restatement.
 * Included by me_oracle.c once per pixel type with
 *     PIX   = uint8_t | uint16_t
 *     SUF(x) = x##_u8 | x##_u16
 * See me_oracle.h for the role of this code.  Citations are to /root/reference/source.
 */

/* ---------------------------------------------------------------- block costs */

/* common/pixel.cpp:78-95 */
static int SUF(blk_sad)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs, int w, int h)
{
    int sum = 0;
    for (int y = 0; y < h; y++, a += as, b += bs)
        for (int x = 0; x < w; x++)
            sum += abs((int)a[x] - (int)b[x]);
    return sum;
}

/* sum of |4x4 2-D Hadamard| of the difference block, NOT halved
 * (common/pixel.cpp:187-213 computes the same quantity in SWAR form) */
static int SUF(blk_had4)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs)
{
    int d[4][4], t[4][4], sum = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            d[y][x] = (int)a[y * as + x] - (int)b[y * bs + x];
    for (int y = 0; y < 4; y++)
    {
        int s01 = d[y][0] + d[y][1], d01 = d[y][0] - d[y][1];
        int s23 = d[y][2] + d[y][3], d23 = d[y][2] - d[y][3];
        t[y][0] = s01 + s23; t[y][1] = d01 + d23; t[y][2] = s01 - s23; t[y][3] = d01 - d23;
    }
    for (int x = 0; x < 4; x++)
    {
        int s01 = t[0][x] + t[1][x], d01 = t[0][x] - t[1][x];
        int s23 = t[2][x] + t[3][x], d23 = t[2][x] - t[3][x];
        sum += abs(s01 + s23) + abs(d01 + d23) + abs(s01 - s23) + abs(d01 - d23);
    }
    return sum;
}

/* sum of |8x8 2-D Hadamard| of the difference block (common/pixel.cpp:276-311) */
static int SUF(blk_had8)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs)
{
    int m[8][8], sum = 0;
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            m[y][x] = (int)a[y * as + x] - (int)b[y * bs + x];
    for (int y = 0; y < 8; y++)           /* rows */
        for (int s = 1; s < 8; s <<= 1)
            for (int i = 0; i < 8; i++)
                if (!(i & s))
                {
                    int p = m[y][i], q = m[y][i + s];
                    m[y][i] = p + q; m[y][i + s] = p - q;
                }
    for (int x = 0; x < 8; x++)           /* columns */
        for (int s = 1; s < 8; s <<= 1)
            for (int i = 0; i < 8; i++)
                if (!(i & s))
                {
                    int p = m[i][x], q = m[i + s][x];
                    m[i][x] = p + q; m[i + s][x] = p - q;
                }
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            sum += abs(m[y][x]);
    return sum;
}

/* satd table: common/pixel.cpp:652-722 (4x4 tiles for w in {4,12}, 8x4 tiles otherwise;
 * both halve a tile-group sum that is always even, so the result is additive over 4x4 tiles) */
static int SUF(blk_satd)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs, int w, int h)
{
    int sum = 0;
    for (int y = 0; y < h; y += 4)
        for (int x = 0; x < w; x += 4)
            sum += SUF(blk_had4)(a + y * as + x, as, b + y * bs + x, bs) >> 1;
    return sum;
}

/* sa8d8<w,h>: common/pixel.cpp:313-316,330-345 */
static int SUF(blk_sa8d8)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs, int w, int h)
{
    int sum = 0;
    for (int y = 0; y < h; y += 8)
        for (int x = 0; x < w; x += 8)
            sum += (SUF(blk_had8)(a + y * as + x, as, b + y * bs + x, bs) + 2) >> 2;
    return sum;
}

/* sa8d16<w,h>: common/pixel.cpp:318-328,347-362 (one rounding per 16x16) */
static int SUF(blk_sa8d16)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs, int w, int h)
{
    int sum = 0;
    for (int y = 0; y < h; y += 16)
        for (int x = 0; x < w; x += 16)
        {
            int s = 0;
            for (int k = 0; k < 4; k++)
            {
                int ox = x + (k & 1) * 8, oy = y + (k >> 1) * 8;
                s += SUF(blk_had8)(a + oy * as + ox, as, b + oy * bs + ox, bs);
            }
            sum += (s + 2) >> 2;
        }
    return sum;
}

/* sa8d_inter table: common/pixel.cpp:756-826 */
static int SUF(blk_sa8d_inter)(const PIX *a, intptr_t as, const PIX *b, intptr_t bs, int w, int h)
{
    if (w == 4 || w == 12 || h == 4 || h == 12)
        return SUF(blk_satd)(a, as, b, bs, w, h);
    if (!(w & 15) && !(h & 15))
        return SUF(blk_sa8d16)(a, as, b, bs, w, h);
    return SUF(blk_sa8d8)(a, as, b, bs, w, h);
}

static int SUF(blk_cost)(int kind, const PIX *a, intptr_t as, const PIX *b, intptr_t bs, int w, int h)
{
    switch (kind)
    {
    case MEO_SAD:        return SUF(blk_sad)(a, as, b, bs, w, h);
    case MEO_SATD:       return SUF(blk_satd)(a, as, b, bs, w, h);
    case MEO_SA8D_INTER: return SUF(blk_sa8d_inter)(a, as, b, bs, w, h);
    }
    return -1;
}

int SUF(meo_cmp)(int kind, int idx, const PIX *fenc, intptr_t fs, const PIX *fref, intptr_t rs)
{
    if (kind == MEO_SA8D_SQUARE)  /* common/pixel.cpp:750-754 */
    {
        int n = 4 << idx;
        if (idx == 0)
            return SUF(blk_satd)(fenc, fs, fref, rs, 4, 4);
        if (idx == 1)
            return SUF(blk_sa8d8)(fenc, fs, fref, rs, 8, 8);
        return SUF(blk_sa8d16)(fenc, fs, fref, rs, n, n);
    }
    return SUF(blk_cost)(kind, fenc, fs, fref, rs, g_dims[idx >> 3], g_dims[idx & 7]);
}

/* sad_x3 / sad_x4: fenc stride is fixed at 64 (common/pixel.cpp:97-142) */
void SUF(meo_sad_xn)(int n, int part, const PIX *fenc, const PIX *const *fref, intptr_t frefstride, int *res)
{
    for (int k = 0; k < n; k++)
        res[k] = SUF(blk_sad)(fenc, 64, fref[k], frefstride, g_dims[part >> 3], g_dims[part & 7]);
}

/* ---------------------------------------------------------------- interpolation */

/* HEVC 8-tap luma filters with the reference's exact casts (common/ipfilter.cpp):
 *  kind 0  filterHorizontal_p_p :91-139   val = (short)(sum + (1 << (hr-1))) >> hr,  clip
 *  kind 1  filterVertical_p_p   :354-425  val = (short)((sum + 32) >> 6),            clip
 *  kind 2  filterHorizontal_p_s :265-309  val = (short)(sum + (-8192 << (6-hr))) >> (6-hr)
 *  kind 3  filterVertical_s_p   :38-89    val = (short)((sum + (1 << (5+hr)) + (8192 << 6)) >> (6+hr)), clip
 *  with hr = 14 - depth.  src points at the output-aligned sample; the taps reach 3
 *  before and 4 after it along the filtered axis. */
static void SUF(filt)(int kind, const void *srcv, intptr_t ss, void *dstv, intptr_t ds,
                      int w, int h, int frac, int depth)
{
    const int16_t *c = g_luma_filter[frac];
    const int hr = 14 - depth;
    const int maxval = (1 << depth) - 1;

    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int sum = 0;
            if (kind == 3)
            {
                const int16_t *s = (const int16_t*)srcv + y * ss + x - 3 * ss;
                for (int k = 0; k < 8; k++)
                    sum += s[k * ss] * c[k];
            }
            else
            {
                const PIX *s = (const PIX*)srcv + y * ss + x;
                intptr_t step = (kind == 1) ? ss : 1;
                s -= 3 * step;
                for (int k = 0; k < 8; k++)
                    sum += (int)s[k * step] * c[k];
            }
            int16_t val;
            switch (kind)
            {
            case 0:
                val = (int16_t)((int16_t)(sum + (1 << (hr - 1))) >> hr);
                break;
            case 1:
                val = (int16_t)((sum + 32) >> 6);
                break;
            case 2:
                val = (int16_t)((int16_t)(sum + (int)(-8192 * (1 << (6 - hr)))) >> (6 - hr));
                ((int16_t*)dstv)[y * ds + x] = val;
                continue;
            default:
                val = (int16_t)((sum + (1 << (5 + hr)) + (8192 << 6)) >> (6 + hr));
                break;
            }
            if (val < 0) val = 0;
            if (val > maxval) val = (int16_t)maxval;
            ((PIX*)dstv)[y * ds + x] = (PIX)val;
        }
}

void SUF(meo_ipfilter)(int kind, const void *src, intptr_t ss, void *dst, intptr_t ds, int w, int h, int frac, int depth)
{
    SUF(filt)(kind, src, ss, dst, ds, w, h, frac, depth);
}

/* ---------------------------------------------------------------- search state */

typedef struct
{
    PIX      fenc[64 * 64];                 /* cached PU, stride 64 (encoder/motion.cpp:89,107) */
    PIX      subpel[65 * 65];               /* motion.cpp:90 */
    int16_t  immed[65 * (65 + 7)];          /* motion.cpp:91 */
    const PIX *fref;                        /* fpelPlane + blockOffset (motion.cpp:288) */
    const PIX *lowres[4];
    intptr_t stride;
    intptr_t blockOffset;
    const uint16_t *cost;                   /* centre of the per-QP table */
    int      mvpx, mvpy;                    /* setMVP (bitcost.h:43) */
    int      w, h, part, depth, isLowres;
    int      ikind;                         /* cost kind of the integer stage (MEO_SAD in the reference) */
    int64_t  evals;
} SUF(state);

static inline int SUF(mvcost)(const SUF(state) *st, int x, int y)
{
    /* bitcost.h:46 -- uint16 sum of two table entries indexed by the q-pel MVD */
    return (uint16_t)(st->cost[(int16_t)(x - st->mvpx)] + st->cost[(int16_t)(y - st->mvpy)]);
}

static inline int SUF(cmp)(SUF(state) *st, int kind, const PIX *ref, intptr_t rs)
{
    st->evals++;
    return SUF(blk_cost)(kind, st->fenc, 64, ref, rs, st->w, st->h);
}

/* SAD at a full-pel MV */
static inline int SUF(sad_at)(SUF(state) *st, int mx, int my)
{
    return SUF(cmp)(st, st->ikind, st->fref + mx + my * st->stride, st->stride);
}

/* COST_MV's cost term: SAD + mvcost(MV(mx,my) << 2)  (motion.cpp:173-179) */
static inline int SUF(cost_at)(SUF(state) *st, int mx, int my)
{
    return SUF(sad_at)(st, mx, my) + SUF(mvcost)(st, (int16_t)(mx << 2), (int16_t)(my << 2));
}

/* subpelCompare (motion.cpp:1110-1166) */
static int SUF(subpel_cmp)(SUF(state) *st, int qx, int qy, int kind)
{
    if (st->isLowres)
    {
        if ((qx | qy) & 1)
        {
            int hpelA = (qy & 2) | ((qx & 2) >> 1);
            const PIX *fa = st->lowres[hpelA] + st->blockOffset + (qx >> 2) + (qy >> 2) * st->stride;
            int bx = (int16_t)(qx + (qx & 1) * 2), by = (int16_t)(qy + (qy & 1) * 2);
            int hpelB = (by & 2) | ((bx & 2) >> 1);
            const PIX *fb = st->lowres[hpelB] + st->blockOffset + (bx >> 2) + (by >> 2) * st->stride;
            for (int y = 0; y < st->h; y++)
                for (int x = 0; x < st->w; x++)
                    st->subpel[y * 64 + x] = (PIX)((fa[y * st->stride + x] + fb[y * st->stride + x] + 1) >> 1);
            return SUF(cmp)(st, kind, st->subpel, 64);
        }
        int hpel = (qy & 2) | ((qx & 2) >> 1);
        const PIX *f = st->lowres[hpel] + st->blockOffset + (qx >> 2) + (qy >> 2) * st->stride;
        return SUF(cmp)(st, kind, f, st->stride);
    }

    const PIX *f = st->fref + (qx >> 2) + (qy >> 2) * st->stride;
    int xf = qx & 3, yf = qy & 3;
    if (!(xf | yf))
        return SUF(cmp)(st, kind, f, st->stride);
    if (!yf)
        SUF(filt)(0, f, st->stride, st->subpel, 64, st->w, st->h, xf, st->depth);
    else if (!xf)
        SUF(filt)(1, f, st->stride, st->subpel, 64, st->w, st->h, yf, st->depth);
    else
    {
        SUF(filt)(2, f - 3 * st->stride, st->stride, st->immed, st->w, st->w, st->h + 7, xf, st->depth);
        SUF(filt)(3, st->immed + 3 * st->w, st->w, st->subpel, 64, st->w, st->h, yf, st->depth);
    }
    return SUF(cmp)(st, kind, st->subpel, 64);
}

/* ---------------------------------------------------------------- integer searches */

typedef struct { int x, y; } SUF(pt);

static inline int SUF(in_range)(int x, int y, const meo_job *j)
{
    /* MV::checkRange (common/mv.h:96-99) */
    return x >= j->mvmin_x && x <= j->mvmax_x && y >= j->mvmin_y && y <= j->mvmax_y;
}

/* Candidate test used by every pattern: strict `<` in program order (common/common.h:71-92). */
#define TRY_MV(mx, my) \
    do { int c_ = SUF(cost_at)(st, (mx), (my)); \
         if (c_ < bcost) { bcost = c_; bmx = (int16_t)(mx); bmy = (int16_t)(my); } } while (0)

/* SAD measured at (sx,sy) but labelled / priced as (mx,my) */
#define TRY_MV_AT(sx, sy, mx, my) \
    do { int c_ = SUF(sad_at)(st, (sx), (sy)) + SUF(mvcost)(st, (int16_t)((int16_t)(mx) << 2), (int16_t)((int16_t)(my) << 2)); \
         if (c_ < bcost) { bcost = c_; bmx = (int16_t)(mx); bmy = (int16_t)(my); } } while (0)

/* HM-style star pattern around a fixed origin (motion.cpp:868-1108) */
static void SUF(star_pattern)(SUF(state) *st, const meo_job *j, int *pbmx, int *pbmy, int *pbcost,
                              int *pPointNr, int *pDistance, int earlyExit, int merange)
{
    int bmx = *pbmx, bmy = *pbmy, bcost = *pbcost;
    const int ox = bmx, oy = bmy;
    int saved = bcost, rounds = 0;

#define PT(mx, my, point, dist) \
    do { int c_ = SUF(cost_at)(st, (mx), (my)); \
         if (c_ < bcost) { bcost = c_; bmx = (int16_t)(mx); bmy = (int16_t)(my); *pPointNr = (point); *pDistance = (dist); } } while (0)
#define DONE() do { *pbmx = bmx; *pbmy = bmy; *pbcost = bcost; } while (0)

    {   /* distance 1: top, left, right, bottom (motion.cpp:886-929) */
        const int16_t top = (int16_t)(oy - 1), bottom = (int16_t)(oy + 1);
        const int16_t left = (int16_t)(ox - 1), right = (int16_t)(ox + 1);
        /* the all-inside x4 branch and the per-point branch test the same points in the same order */
        if (top >= j->mvmin_y)    PT(ox, top, 2, 1);
        if (left >= j->mvmin_x)   PT(left, oy, 4, 1);
        if (right <= j->mvmax_x)  PT(right, oy, 5, 1);
        if (bottom <= j->mvmax_y) PT(ox, bottom, 7, 1);
        if (bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExit)
        {
            DONE();
            return;
        }
    }

    for (int16_t dist = 2; dist <= 8; dist <<= 1)     /* motion.cpp:931-1010 */
    {
        const int16_t top = (int16_t)(oy - dist), bottom = (int16_t)(oy + dist);
        const int16_t left = (int16_t)(ox - dist), right = (int16_t)(ox + dist);
        const int16_t top2 = (int16_t)(oy - (dist >> 1)), bottom2 = (int16_t)(oy + (dist >> 1));
        const int16_t left2 = (int16_t)(ox - (dist >> 1)), right2 = (int16_t)(ox + (dist >> 1));
        saved = bcost;

        if (top >= j->mvmin_y && left >= j->mvmin_x && right <= j->mvmax_x && bottom <= j->mvmax_y)
        {
            PT(ox, top, 2, dist);
            PT(left2, top2, 1, dist >> 1);
            PT(right2, top2, 3, dist >> 1);
            PT(left, oy, 4, dist);
            PT(right, oy, 5, dist);
            PT(left2, bottom2, 6, dist >> 1);
            PT(right2, bottom2, 8, dist >> 1);
            PT(ox, bottom, 7, dist);
        }
        else
        {
            if (top >= j->mvmin_y) PT(ox, top, 2, dist);
            if (top2 >= j->mvmin_y)
            {
                if (left2 >= j->mvmin_x)  PT(left2, top2, 1, dist >> 1);
                if (right2 <= j->mvmax_x) PT(right2, top2, 3, dist >> 1);
            }
            if (left >= j->mvmin_x)  PT(left, oy, 4, dist);
            if (right <= j->mvmax_x) PT(right, oy, 5, dist);
            if (bottom2 <= j->mvmax_y)
            {
                if (left2 >= j->mvmin_x)  PT(left2, bottom2, 6, dist >> 1);
                if (right2 <= j->mvmax_x) PT(right2, bottom2, 8, dist >> 1);
            }
            if (bottom <= j->mvmax_y) PT(ox, bottom, 7, dist);
        }
        if (bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExit)
        {
            DONE();
            return;
        }
    }

    for (int16_t dist = 16; dist <= (int16_t)merange; dist <<= 1)   /* motion.cpp:1012-1107 */
    {
        const int16_t top = (int16_t)(oy - dist), bottom = (int16_t)(oy + dist);
        const int16_t left = (int16_t)(ox - dist), right = (int16_t)(ox + dist);
        saved = bcost;
        if (top >= j->mvmin_y && left >= j->mvmin_x && right <= j->mvmax_x && bottom <= j->mvmax_y)
        {
            PT(ox, top, 0, dist);
            PT(left, oy, 0, dist);
            PT(right, oy, 0, dist);
            PT(ox, bottom, 0, dist);
            for (int16_t idx = 1; idx < 4; idx++)
            {
                int16_t yt = (int16_t)(top + (dist >> 2) * idx), yb = (int16_t)(bottom - (dist >> 2) * idx);
                int16_t xl = (int16_t)(ox - (dist >> 2) * idx), xr = (int16_t)(ox + (dist >> 2) * idx);
                PT(xl, yt, 0, dist);
                PT(xr, yt, 0, dist);
                PT(xl, yb, 0, dist);
                PT(xr, yb, 0, dist);
            }
        }
        else
        {
            if (top >= j->mvmin_y)    PT(ox, top, 0, dist);
            if (left >= j->mvmin_x)   PT(left, oy, 0, dist);
            if (right <= j->mvmax_x)  PT(right, oy, 0, dist);
            if (bottom <= j->mvmax_y) PT(ox, bottom, 0, dist);
            for (int16_t idx = 1; idx < 4; idx++)
            {
                int16_t yt = (int16_t)(top + (dist >> 2) * idx), yb = (int16_t)(bottom - (dist >> 2) * idx);
                int16_t xl = (int16_t)(ox - (dist >> 2) * idx), xr = (int16_t)(ox + (dist >> 2) * idx);
                if (yt >= j->mvmin_y)
                {
                    if (xl >= j->mvmin_x) PT(xl, yt, 0, dist);
                    if (xr <= j->mvmax_x) PT(xr, yt, 0, dist);
                }
                if (yb <= j->mvmax_y)
                {
                    if (xl >= j->mvmin_x) PT(xl, yb, 0, dist);
                    if (xr <= j->mvmax_x) PT(xr, yb, 0, dist);
                }
            }
        }
        if (bcost < saved)
            rounds = 0;
        else if (++rounds >= earlyExit)
        {
            DONE();
            return;
        }
    }
    DONE();
#undef PT
#undef DONE
}

static const int8_t SUF(k_star_offsets)[16][2] =   /* two-point refinement table, motion.cpp:133-143 */
{
    { -1, 0 }, { 0, -1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { -1, -1 },
    { 1, -1 }, { 1, 1 }, { -1, 0 }, { 0, 1 }, { -1, 1 }, { 1, 1 }, { 1, 0 }, { 0, 1 },
};

/* the whole of MotionEstimate::motionEstimate (motion.cpp:277-866) */
/* The search of TEncSearch::xMotionEstimation (bidirectional refinement,
 * Lib/TLibEncoder/TEncSearch.cpp:2993-3000): xPatternSearch (:3026-3085), an exhaustive raster
 * scan of [mvmin, mvmax] with SAD + mvcost and strict `<`, then xPatternRefinement (:224-285) at
 * half-pel and at quarter-pel: 9 candidates each, SATD of the interpolated block + mvcost, in the
 * order of s_mvRefineHpel / s_mvRefineQPel (:59-83).  The cached PU is whatever the encoder plane
 * holds at the PU position (the caller's 2*org - otherPred block).  Returns the quarter-pel
 * refinement's cost (what xMotionEstimation keeps in outCost before its bit accounting). */
static int SUF(bipred_search)(SUF(state) *st, const meo_job *j, int16_t *outx, int16_t *outy)
{
    static const int8_t refine_hpel[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
    static const int8_t refine_qpel[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } };
    st->ikind = MEO_SAD;
    st->mvpx = j->qmvp_x;
    st->mvpy = j->qmvp_y;
    unsigned bcost = ~0u;
    int16_t bx = 0, by = 0;
    for (int y = j->mvmin_y; y <= j->mvmax_y; y++)
        for (int x = j->mvmin_x; x <= j->mvmax_x; x++)
        {
            unsigned c = (unsigned)SUF(cost_at)(st, x, y);
            if (c < bcost)
            {
                bcost = c;
                bx = (int16_t)x;
                by = (int16_t)y;
            }
        }
    for (int pass = 0; pass < 2; pass++)
    {
        const int8_t (*tab)[2] = pass ? refine_qpel : refine_hpel;
        const int scale = pass ? 1 : 2;
        bx = (int16_t)(bx << 1);            /* full-pel -> half-pel -> quarter-pel units */
        by = (int16_t)(by << 1);
        int best = 0;
        bcost = ~0u;
        for (int i = 0; i < 9; i++)
        {
            int16_t qx = (int16_t)((int16_t)(bx + tab[i][0]) * scale), qy = (int16_t)((int16_t)(by + tab[i][1]) * scale);
            unsigned c = (unsigned)SUF(mvcost)(st, qx, qy) + (unsigned)SUF(subpel_cmp)(st, qx, qy, MEO_SATD);
            if (c < bcost)
            {
                bcost = c;
                best = i;
            }
        }
        bx = (int16_t)(bx + tab[best][0]);
        by = (int16_t)(by + tab[best][1]);
    }
    *outx = bx;
    *outy = by;
    return (int)bcost;
}

static int SUF(search)(SUF(state) *st, const meo_job *j, int16_t *outx, int16_t *outy)
{
    if (j->method == MEO_BIPRED)
        return SUF(bipred_search)(st, j, outx, outy);
    static const int8_t hex2[8][2] = { { -1, -2 }, { -2, 0 }, { -1, 2 }, { 1, 2 }, { 2, 0 }, { 1, -2 }, { -1, -2 }, { -2, 0 } };
    static const int8_t square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
    static const int8_t hex4[16][2] = { { 0, -4 }, { 0, 4 }, { -2, -3 }, { 2, -3 }, { -4, -2 }, { 4, -2 }, { -4, -1 }, { 4, -1 },
                                        { -4, 0 }, { 4, 0 }, { -4, 1 }, { 4, 1 }, { -4, 2 }, { 4, 2 }, { -2, 3 }, { 2, 3 } };
    /* workload[] (motion.cpp:51-60): hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd */
    static const int8_t workload[8][5] = { { 1, 4, 0, 4, 0 }, { 1, 4, 1, 4, 0 }, { 1, 4, 1, 4, 1 }, { 2, 4, 1, 4, 1 },
                                           { 2, 4, 2, 4, 1 }, { 1, 8, 1, 8, 1 }, { 2, 8, 1, 8, 1 }, { 2, 8, 2, 8, 1 } };

    int merange = j->merange;
    st->ikind = j->icost;
    st->mvpx = j->qmvp_x;
    st->mvpy = j->qmvp_y;

    const int16_t qminx = (int16_t)(j->mvmin_x << 2), qminy = (int16_t)(j->mvmin_y << 2);
    const int16_t qmaxx = (int16_t)(j->mvmax_x << 2), qmaxy = (int16_t)(j->mvmax_y << 2);
#define CLIPQ(v, lo, hi) ((int16_t)(((v) > (hi) ? (hi) : (v)) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v))))

    /* predictor phase (motion.cpp:302-342) */
    int16_t pmx = CLIPQ(j->qmvp_x, qminx, qmaxx), pmy = CLIPQ(j->qmvp_y, qminy, qmaxy);
    int16_t bprex = pmx, bprey = pmy;
    int bprecost = SUF(subpel_cmp)(st, pmx, pmy, st->ikind);    /* no mvcost term (motion.cpp:305) */

    int16_t bmx = (int16_t)((int16_t)(pmx + 2) >> 2), bmy = (int16_t)((int16_t)(pmy + 2) >> 2);
    int bcost = bprecost;
    if ((pmx | pmy) & 3)
        bcost = SUF(cost_at)(st, bmx, bmy);
    if (pmx || pmy)
    {
        int c = SUF(sad_at)(st, 0, 0) + SUF(mvcost)(st, 0, 0);
        if (c < bcost)
        {
            bcost = c;
            bmx = bmy = 0;
        }
    }
    for (int i = 0; i < j->numc; i++)
    {
        int16_t mx = CLIPQ(j->mvc[i][0], qminx, qmaxx), my = CLIPQ(j->mvc[i][1], qminy, qmaxy);
        if ((mx || my) && !(mx == pmx && my == pmy) && !(mx == bprex && my == bprey))
        {
            int c = SUF(subpel_cmp)(st, mx, my, st->ikind) + SUF(mvcost)(st, mx, my);
            if (c < bprecost)
            {
                bprecost = c;
                bprex = mx;
                bprey = my;
            }
        }
    }
    pmx = (int16_t)((int16_t)(pmx + 2) >> 2);
    pmy = (int16_t)((int16_t)(pmy + 2) >> 2);
    int16_t omx = bmx, omy = bmy;
    int do_hex = 0;

    switch (j->method)
    {
    case MEO_DIA:   /* motion.cpp:346-367 */
    {
        int i = merange;
        do
        {
            const int cx = bmx, cy = bmy;
            int dir = 0;
            for (int k = 1; k <= 4; k++)
            {
                int c = SUF(cost_at)(st, (int16_t)(cx + square1[k][0]), (int16_t)(cy + square1[k][1]));
                if (c < bcost)
                {
                    bcost = c;
                    dir = k;
                }
            }
            if (!dir)
                break;
            bmx = (int16_t)(cx + square1[dir][0]);
            bmy = (int16_t)(cy + square1[dir][1]);
        }
        while (--i && SUF(in_range)(bmx, bmy, j));
        break;
    }

    case MEO_HEX:
        do_hex = 1;
        break;

    case MEO_UMH:   /* motion.cpp:443-628 */
    {
        /* quirk: every x4 evaluation in this block measures its SADs relative to the search
         * start `base` (pix_base is computed once, motion.cpp:451) but labels and prices the
         * candidates relative to the *current* omv. */
        const int basex = bmx, basey = bmy;
        const int scale = (st->w * st->h) >> 4;     /* size_scale[] (motion.cpp:64-80) */
        int cross_start = 1;
#define SAD_THRESH(v) (bcost < (((v) >> 4) * scale))
#define X4ONE(dx, dy) \
    do { int c_ = SUF(sad_at)(st, basex + (dx), basey + (dy)) + \
                  SUF(mvcost)(st, (int16_t)((int16_t)(omx + (dx)) << 2), (int16_t)((int16_t)(omy + (dy)) << 2)); \
         if (c_ < bcost) { bcost = c_; bmx = (int16_t)(omx + (dx)); bmy = (int16_t)(omy + (dy)); } } while (0)
#define X4(a0, a1, b0, b1, c0, c1, d0, d1) do { X4ONE(a0, a1); X4ONE(b0, b1); X4ONE(c0, c1); X4ONE(d0, d1); } while (0)
#define CROSS(start, x_max, y_max) \
    do { \
        int16_t i_ = (int16_t)(start); \
        if ((x_max) <= MEO_MIN(j->mvmax_x - omx, omx - j->mvmin_x)) \
            for (; i_ < (x_max) - 2; i_ += 4) \
                X4(i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2, 0); \
        for (; i_ < (x_max); i_ += 2) \
        { \
            if (omx + i_ <= j->mvmax_x) TRY_MV(omx + i_, omy); \
            if (omx - i_ >= j->mvmin_x) TRY_MV(omx - i_, omy); \
        } \
        i_ = (int16_t)(start); \
        if ((y_max) <= MEO_MIN(j->mvmax_y - omy, omy - j->mvmin_y)) \
            for (; i_ < (y_max) - 2; i_ += 4) \
                X4(0, i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2); \
        for (; i_ < (y_max); i_ += 2) \
        { \
            /* quirk: COST_MV(omv.x, omv.y + i) expands to fref + omv.x + omv.y + i * stride \
             * (motion.cpp:176,271-273): the SAD is taken at (omv.x + omv.y, +-i) while the \
             * candidate is labelled and priced as (omv.x, omv.y +- i). */ \
            if (omy + i_ <= j->mvmax_y) TRY_MV_AT(omx + omy, i_, omx, omy + i_); \
            if (omy - i_ >= j->mvmin_y) TRY_MV_AT(omx + omy, -i_, omx, omy - i_); \
        } \
    } while (0)

        int ucost1 = bcost, ucost2;
        omx = pmx; omy = pmy;
        X4(0, -1, 0, 1, -1, 0, 1, 0);
        if (pmx || pmy)
        {
            omx = 0; omy = 0;
            X4(0, -1, 0, 1, -1, 0, 1, 0);
        }
        ucost2 = bcost;
        if ((bmx || bmy) && !(bmx == pmx && bmy == pmy))
        {
            omx = bmx; omy = bmy;
            X4(0, -1, 0, 1, -1, 0, 1, 0);
        }
        if (bcost == ucost2)
            cross_start = 3;

        /* early termination (motion.cpp:463-481) */
        omx = bmx; omy = bmy;
        int terminated = 0;
        if (bcost == ucost2 && SAD_THRESH(2000))
        {
            X4(0, -2, -1, -1, 1, -1, -2, 0);
            X4(2, 0, -1, 1, 1, 1, 0, 2);
            if (bcost == ucost1 && SAD_THRESH(500))
                terminated = 1;
            else if (bcost == ucost2)
            {
                int16_t range = (int16_t)((int16_t)(merange >> 1) | 1);
                CROSS(3, range, range);
                X4(-1, -2, 1, -2, -2, -1, 2, -1);
                X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2)
                    terminated = 1;
                else
                    cross_start = range + 2;
            }
        }
        if (terminated)
            break;

        /* adaptive range from predictor spread (motion.cpp:486-538) */
        if (j->numc)
        {
            static const uint8_t range_mul[4][4] = { { 3, 3, 4, 4 }, { 3, 4, 4, 4 }, { 4, 4, 4, 5 }, { 4, 4, 5, 6 } };
            int mvd, denom = 1;
            const int is64 = (st->w == 64 && st->h == 64);
            if (j->numc == 1)
            {
                if (is64)
                    mvd = 25;
                else
                    mvd = abs(j->qmvp_x - j->mvc[0][0]) + abs(j->qmvp_y - j->mvc[0][1]);
            }
            else
            {
                denom = j->numc - 1;
                mvd = 0;
                if (!is64)
                {
                    mvd = abs(j->qmvp_x - j->mvc[0][0]) + abs(j->qmvp_y - j->mvc[0][1]);
                    denom++;
                }
                for (int i = 0; i < j->numc - 1; i++)   /* x265_predictor_difference, motion.cpp:146-157 */
                    mvd += abs(j->mvc[i][0] - j->mvc[i + 1][0]) + abs(j->mvc[i][1] - j->mvc[i + 1][1]);
            }
            int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            merange = (merange * range_mul[mvd_ctx][sad_ctx]) >> 2;
        }

        CROSS(cross_start, merange, merange >> 1);
        X4(-2, -2, -2, 2, 2, -2, 2, 2);

        /* hexagon grid (motion.cpp:545-624): 16 points scaled by i around a fixed omv */
        omx = bmx; omy = bmy;
        int16_t i = 1;
        do
        {
            if (4 * i > MEO_MIN(MEO_MIN(j->mvmax_x - omx, omx - j->mvmin_x), MEO_MIN(j->mvmax_y - omy, omy - j->mvmin_y)))
            {
                for (int k = 0; k < 16; k++)
                {
                    int16_t mx = (int16_t)(omx + (int16_t)(hex4[k][0] * i)), my = (int16_t)(omy + (int16_t)(hex4[k][1] * i));
                    if (SUF(in_range)(mx, my, j))
                        TRY_MV(mx, my);
                }
            }
            else
            {
                /* unclipped ring: mvcost is read straight from the table rows at omv*4 and is
                 * NOT truncated to 16 bits here (motion.cpp:547-548,575) */
                int best = -1;
                for (int k = 0; k < 16; k++)
                {
                    int mx = omx + hex4[k][0] * i, my = omy + hex4[k][1] * i;
                    int c = SUF(sad_at)(st, mx, my)
                        + st->cost[(int16_t)(omx * 4 + hex4[k][0] * 4 * i - st->mvpx)]
                        + st->cost[(int16_t)(omy * 4 + hex4[k][1] * 4 * i - st->mvpy)];
                    if (c < bcost)
                    {
                        bcost = c;
                        best = k;
                    }
                }
                if (best >= 0)
                {
                    bmx = (int16_t)(omx + i * hex4[best][0]);
                    bmy = (int16_t)(omy + i * hex4[best][1]);
                }
            }
        }
        while (++i <= merange >> 2);
        if (SUF(in_range)(bmx, bmy, j))
            do_hex = 1;      /* goto me_hex2 with the adapted merange */
        break;
#undef SAD_THRESH
#undef X4ONE
#undef X4
#undef CROSS
    }

    case MEO_STAR:  /* motion.cpp:630-737 */
    {
        int bPointNr = 0, bDistance = 0, ibmx = bmx, ibmy = bmy;
        SUF(star_pattern)(st, j, &ibmx, &ibmy, &bcost, &bPointNr, &bDistance, 3, merange);
        bmx = (int16_t)ibmx; bmy = (int16_t)ibmy;
        int stop = 0;
        if (bDistance == 1)
        {
            if (bPointNr)
            {
                int saved = bcost;
                const int16_t x1 = (int16_t)(bmx + SUF(k_star_offsets)[(bPointNr - 1) * 2][0]);
                const int16_t y1 = (int16_t)(bmy + SUF(k_star_offsets)[(bPointNr - 1) * 2][1]);
                const int16_t x2 = (int16_t)(bmx + SUF(k_star_offsets)[(bPointNr - 1) * 2 + 1][0]);
                const int16_t y2 = (int16_t)(bmy + SUF(k_star_offsets)[(bPointNr - 1) * 2 + 1][1]);
                if (SUF(in_range)(x1, y1, j)) TRY_MV(x1, y1);
                if (SUF(in_range)(x2, y2, j)) TRY_MV(x2, y2);
                if (bcost == saved)
                    stop = 1;
            }
            else
                stop = 1;
        }
        if (stop)
            break;

        if (bDistance > 5)   /* raster refinement, step 5 (motion.cpp:667-701) */
        {
            for (int16_t ty = j->mvmin_y; ty <= j->mvmax_y; ty += 5)
                for (int16_t tx = j->mvmin_x; tx <= j->mvmax_x; tx += 5)
                {
                    if (tx + 15 <= j->mvmax_x)
                    {
                        for (int k = 0; k < 4; k++)
                        {
                            /* quirk: the 4th candidate is priced with mvcost(tmv << 3) (motion.cpp:694) */
                            int sh = (k == 3) ? 3 : 2;
                            int c = SUF(sad_at)(st, tx, ty) + SUF(mvcost)(st, (int16_t)(tx << sh), (int16_t)(ty << sh));
                            if (c < bcost)
                            {
                                bcost = c;
                                bmx = tx;
                                bmy = ty;
                            }
                            if (k < 3)
                                tx += 5;
                        }
                    }
                    else
                        TRY_MV(tx, ty);
                }
        }

        while (bDistance > 0)
        {
            bDistance = 0;
            bPointNr = 0;
            ibmx = bmx; ibmy = bmy;
            SUF(star_pattern)(st, j, &ibmx, &ibmy, &bcost, &bPointNr, &bDistance, 32, merange);
            bmx = (int16_t)ibmx; bmy = (int16_t)ibmy;
            if (bDistance == 1)
            {
                if (!bPointNr)
                    break;
                const int16_t x1 = (int16_t)(bmx + SUF(k_star_offsets)[(bPointNr - 1) * 2][0]);
                const int16_t y1 = (int16_t)(bmy + SUF(k_star_offsets)[(bPointNr - 1) * 2][1]);
                const int16_t x2 = (int16_t)(bmx + SUF(k_star_offsets)[(bPointNr - 1) * 2 + 1][0]);
                const int16_t y2 = (int16_t)(bmy + SUF(k_star_offsets)[(bPointNr - 1) * 2 + 1][1]);
                if (SUF(in_range)(x1, y1, j)) TRY_MV(x1, y1);
                if (SUF(in_range)(x2, y2, j)) TRY_MV(x2, y2);
                break;
            }
        }
        break;
    }

    case MEO_FULL:  /* motion.cpp:738-771: raster order; x4 grouping does not change order */
    default:
        for (int16_t ty = j->mvmin_y; ty <= j->mvmax_y; ty++)
            for (int16_t tx = j->mvmin_x; tx <= j->mvmax_x; tx++)
                TRY_MV(tx, ty);
        break;
    }

    if (do_hex)     /* me_hex2 (motion.cpp:369-441) */
    {
        int dir = -1;
        {
            const int cx = bmx, cy = bmy;
            /* (-2,0) (-1,2) (1,2) | (2,0) (1,-2) (-1,-2) == hex2[1..6] */
            for (int k = 0; k < 6; k++)
            {
                int c = SUF(cost_at)(st, (int16_t)(cx + hex2[k + 1][0]), (int16_t)(cy + hex2[k + 1][1]));
                if (c < bcost)
                {
                    bcost = c;
                    dir = k;
                }
            }
        }
        if (dir >= 0)
        {
            bmx = (int16_t)(bmx + hex2[dir + 1][0]);
            bmy = (int16_t)(bmy + hex2[dir + 1][1]);
            for (int i = (merange >> 1) - 1; i > 0 && SUF(in_range)(bmx, bmy, j); i--)
            {
                int step = -1;
                for (int k = 0; k < 3; k++)
                {
                    int c = SUF(cost_at)(st, (int16_t)(bmx + hex2[dir + k][0]), (int16_t)(bmy + hex2[dir + k][1]));
                    if (c < bcost)
                    {
                        bcost = c;
                        step = k;
                    }
                }
                if (step < 0)
                    break;
                dir = (dir + step - 1 + 6) % 6;      /* dir += tag-2; dir = mod6m1[dir+1] */
                bmx = (int16_t)(bmx + hex2[dir + 1][0]);
                bmy = (int16_t)(bmy + hex2[dir + 1][1]);
            }
        }
        /* square refine: 8 neighbours of the same bmv (motion.cpp:427-439) */
        int sq = 0;
        for (int k = 1; k <= 8; k++)
        {
            int c = SUF(cost_at)(st, (int16_t)(bmx + square1[k][0]), (int16_t)(bmy + square1[k][1]));
            if (c < bcost)
            {
                bcost = c;
                sq = k;
            }
        }
        bmx = (int16_t)(bmx + square1[sq][0]);
        bmy = (int16_t)(bmy + square1[sq][1]);
    }

    /* motion.cpp:778-784 */
    if (bprecost < bcost)
    {
        bmx = bprex;
        bmy = bprey;
        bcost = bprecost;
    }
    else
    {
        bmx = (int16_t)(bmx << 2);
        bmy = (int16_t)(bmy << 2);
    }

    /* sub-pel refinement (motion.cpp:786-861).  The reference's full-res half-pel loop
     * interpolates one (w+1)xh or wx(h+1) block per direction pair; both members of a pair
     * share the fractional phase, so each candidate equals an independent subpelCompare. */
    const int8_t *wl = workload[j->subme];
    const int hkind = wl[4] ? MEO_SATD : st->ikind;
    if (wl[4])
        bcost = SUF(subpel_cmp)(st, bmx, bmy, MEO_SATD) + SUF(mvcost)(st, bmx, bmy);
    for (int it = 0; it < wl[0]; it++)
    {
        int bdir = 0;
        for (int i = 1; i <= wl[1]; i++)
        {
            int16_t qx = (int16_t)(bmx + square1[i][0] * 2), qy = (int16_t)(bmy + square1[i][1] * 2);
            int c = SUF(subpel_cmp)(st, qx, qy, hkind) + SUF(mvcost)(st, qx, qy);
            if (c < bcost)
            {
                bcost = c;
                bdir = i;
            }
        }
        bmx = (int16_t)(bmx + square1[bdir][0] * 2);
        bmy = (int16_t)(bmy + square1[bdir][1] * 2);
    }
    if (!wl[4])
        bcost = SUF(subpel_cmp)(st, bmx, bmy, MEO_SATD) + SUF(mvcost)(st, bmx, bmy);
    for (int it = 0; it < wl[2]; it++)
    {
        int bdir = 0;
        for (int i = 1; i <= wl[3]; i++)
        {
            int16_t qx = (int16_t)(bmx + square1[i][0]), qy = (int16_t)(bmy + square1[i][1]);
            int c = SUF(subpel_cmp)(st, qx, qy, MEO_SATD) + SUF(mvcost)(st, qx, qy);
            if (c < bcost)
            {
                bcost = c;
                bdir = i;
            }
        }
        bmx = (int16_t)(bmx + square1[bdir][0]);
        bmy = (int16_t)(bmy + square1[bdir][1]);
    }
    *outx = bmx;
    *outy = bmy;
    return bcost;
#undef CLIPQ
}
#undef TRY_MV
#undef TRY_MV_AT

/* setSourcePU (motion.cpp:104-119) + state binding */
static void SUF(bind)(SUF(state) *st, const meo_plane *enc, const meo_plane *ref, int pu_x, int pu_y, int w, int h, int qp)
{
    const PIX *src = (const PIX*)enc->pix + (intptr_t)pu_y * enc->stride + pu_x;
    for (int y = 0; y < h; y++)
        memcpy(st->fenc + y * 64, src + y * enc->stride, (size_t)w * sizeof(PIX));
    st->w = w;
    st->h = h;
    st->part = meo_partition_from_sizes(w, h);
    if (ref)
    {
        st->stride = ref->stride;
        st->blockOffset = (intptr_t)pu_y * ref->stride + pu_x;
        st->fref = (const PIX*)ref->pix + st->blockOffset;
        st->depth = ref->bit_depth;
        st->isLowres = ref->lowres[0] != NULL;
        for (int i = 0; i < 4; i++)
            st->lowres[i] = (const PIX*)ref->lowres[i];
    }
    st->cost = cost_table_for_qp(qp);
    st->evals = 0;
}

/* ---------------------------------------------------------------- lookahead (half-resolution) path */

/* frame_init_lowres_core (common/pixel.cpp:614-637): the four half-resolution planes
 * F (full-pel), H, V, C (half-pel shifted) from 2x2 / shifted 2x2 neighbourhoods */
static void SUF(lowres_core)(const PIX *src0, PIX *dst0, PIX *dsth, PIX *dstv, PIX *dstc,
                             intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
#define AVG4(a, b, c, d) (PIX)(((((a) + (b) + 1) >> 1) + (((c) + (d) + 1) >> 1) + 1) >> 1)
    for (int y = 0; y < height; y++)
    {
        const PIX *r0 = src0 + 2 * y * src_stride, *r1 = r0 + src_stride, *r2 = r1 + src_stride;
        for (int x = 0; x < width; x++)
        {
            dst0[y * dst_stride + x] = AVG4(r0[2 * x], r1[2 * x], r0[2 * x + 1], r1[2 * x + 1]);
            dsth[y * dst_stride + x] = AVG4(r0[2 * x + 1], r1[2 * x + 1], r0[2 * x + 2], r1[2 * x + 2]);
            dstv[y * dst_stride + x] = AVG4(r1[2 * x], r2[2 * x], r1[2 * x + 1], r2[2 * x + 1]);
            dstc[y * dst_stride + x] = AVG4(r1[2 * x + 1], r2[2 * x + 1], r1[2 * x + 2], r2[2 * x + 2]);
        }
    }
#undef AVG4
}

/* TComPicYuv::xExtendPicCompBorder (Lib/TLibCommon/TComPicYuv.cpp:259-286) */
static void SUF(extend_border)(PIX *p, intptr_t stride, int width, int height, int mx, int my)
{
    for (int y = 0; y < height; y++)
    {
        PIX *row = p + y * stride;
        for (int x = 0; x < mx; x++)
        {
            row[-mx + x] = row[0];
            row[width + x] = row[width - 1];
        }
    }
    const size_t bytes = sizeof(PIX) * (size_t)(width + 2 * mx);
    for (int y = 0; y < my; y++)
    {
        memcpy(p + (height + y) * stride - mx, p + (height - 1) * stride - mx, bytes);
        memcpy(p - (y + 1) * stride - mx, p - mx, bytes);
    }
}

/* Lowres::init, picture part (common/lowres.cpp:175-185): downscale, then extend the borders of
 * all four planes.  `src` points at pixel (0,0) of a plane whose margins hold replicated borders
 * (the reference extends the source by 7 columns / 2*lines - height rows itself, lowres.cpp:139-173,
 * and its core then reads one column and one row further still). */
static void SUF(lowres_init)(const PIX *src, intptr_t src_stride, PIX *const dst[4], intptr_t dst_stride,
                             int width, int lines, int mx, int my)
{
    SUF(lowres_core)(src, dst[0], dst[1], dst[2], dst[3], src_stride, dst_stride, width, lines);
    for (int k = 0; k < 4; k++)
        SUF(extend_border)(dst[k], dst_stride, width, lines, mx, my);
}

/* median() of encoder/slicetype.cpp:48-57, including its 16-bit intermediate (which only matters for
 * components more than 32767 apart, beyond any picture size) */
static inline int SUF(median3)(int a, int b, int c)
{
    int t = (int16_t)(a - b < 0 ? a - b : 0);
    a = (int16_t)(a - t);                               /* max(a, b) */
    b = (int16_t)(b + t);                               /* min(a, b) */
    b = (int16_t)(b - (b - c < 0 ? b - c : 0));         /* max(min(a, b), c) */
    b = (int16_t)(b + (a - b < 0 ? a - b : 0));         /* min of the two */
    return b;
}

/* The motion-search part of Lookahead::estimateFrameCost / estimateCUCost for one reference list
 * (encoder/slicetype.cpp:306-316, 346-369, 381-407): 8x8 CUs of the half-resolution picture in
 * reverse raster order, each predicted from its already searched right / lower neighbours.
 * out[cux + cuy * cuw] = (lowresMvs, lowresMvCosts) of that CU. */
static int64_t SUF(lookahead_me)(const meo_plane *enc, const meo_plane *ref, int cuw, int cuh, int merange,
                                 int method, int subme, int qp, meo_result *out)
{
    SUF(state) *st = (SUF(state)*)malloc(sizeof(SUF(state)));
    int64_t evals = 0;
    for (int cuy = cuh - 1; cuy >= 0; cuy--)
    {
        for (int cux = cuw - 1; cux >= 0; cux--)
        {
            const int xy = cux + cuy * cuw;
            meo_job j;
            memset(&j, 0, sizeof(j));
            j.pu_x = 8 * cux;
            j.pu_y = 8 * cuy;
            j.w = j.h = 8;
            /* slicetype.cpp:366-369 */
            j.mvmin_x = (int16_t)(uint16_t)(-cux * 8 - 8);
            j.mvmin_y = (int16_t)(uint16_t)(-cuy * 8 - 8);
            j.mvmax_x = (int16_t)(uint16_t)((cuw - cux - 1) * 8 + 8);
            j.mvmax_y = (int16_t)(uint16_t)((cuh - cuy - 1) * 8 + 8);
            j.merange = (int16_t)merange;
            j.method = (uint8_t)method;
            j.subme = (uint8_t)subme;
            j.qp = (uint8_t)qp;
            /* slicetype.cpp:385-405: candidates right, below, below-left, below-right; mvc[0] and
             * mvc[2] start as zero vectors */
            int numc = 0;
#define ADD_MVC(k) do { j.mvc[numc][0] = out[k].mvx; j.mvc[numc][1] = out[k].mvy; numc++; } while (0)
            if (cux < cuw - 1)
                ADD_MVC(xy + 1);
            if (cuy < cuh - 1)
            {
                ADD_MVC(xy + cuw);
                if (cux > 0)
                    ADD_MVC(xy + cuw - 1);
                if (cux < cuw - 1)
                    ADD_MVC(xy + cuw + 1);
            }
#undef ADD_MVC
            j.numc = (uint8_t)numc;
            if (numc <= 1)
            {
                j.qmvp_x = j.mvc[0][0];
                j.qmvp_y = j.mvc[0][1];
            }
            else
            {
                j.qmvp_x = (int16_t)SUF(median3)(j.mvc[0][0], j.mvc[1][0], j.mvc[2][0]);
                j.qmvp_y = (int16_t)SUF(median3)(j.mvc[0][1], j.mvc[1][1], j.mvc[2][1]);
            }
            SUF(bind)(st, enc, ref, j.pu_x, j.pu_y, 8, 8, qp);
            out[xy].cost = SUF(search)(st, &j, &out[xy].mvx, &out[xy].mvy);
            evals += st->evals;
        }
    }
    free(st);
    return evals;
}

/* ---------------------------------------------------------------- whole-frame field, predInterSearch's chain */

static inline int SUF(clipi)(int v, int lo, int hi)
{
    v = v > hi ? hi : v;
    return v < lo ? lo : v;
}

/* The searches of TEncSearch::predInterSearch (Lib/TLibEncoder/TEncSearch.cpp:2341-2455) for a uniform grid of
 * pu_w x pu_h PUs, CUs in raster order: AMVP candidates A = left MV, B = first of above-right / above / above-left MV
 * (dropped if equal to A), zero-padded to two; mvp = the one with the smaller template SAD at the clipMv'ed
 * candidate (xEstimateMvPredAMVP :2767-2819, xGetTemplateCost :2944-2958, TComDataCU::clipMv
 * TComDataCU.cpp:2724-2736; equal index bits, ties keep the first); range = xSetSearchRange (:3007-3024);
 * motionEstimate with the three neighbour MVs m_mvPredictors (:2366-2368) as candidates.
 * col (optional): the co-located picture's motion, one entry per 16x16 block in raster order ((pic_w + 15) / 16 per row), already
 * scaled to this picture / reference distance (xGetColMVP, :3008-3077); cost < 0 = the block has none (intra, other list empty).
 * part_mode 0: every PU is the 2Nx2N PU of its own CU.  1 (2NxN): CUs of pu_w x 2 pu_h, PU rows 2r / 2r + 1 their upper /
 * lower parts; 2 (Nx2N): CUs of 2 pu_w x pu_h, PU columns 2c / 2c + 1 their left / right parts; the parts of a CU are
 * searched one after the other (the partIdx loop, :2341).  clipMv / xSetSearchRange take the CU's position.  Where the
 * neighbour positions of TComDataCU::fillMvpCand (TComDataCU.cpp:2553-2633) fall: the upper part of a 2NxN CU has the
 * lower part of the CU to its left below-left of it, coded already, and the left scan asks below-left first (:2574-2588),
 * so A is that PU's MV; the lower part has the upper part of its own CU above it and nothing coded above-right.
 * part_mode 4..7 (the asymmetric partitions, numbered as PartSize, Lib/TLibCommon/TypeDef.h:100-111): pu_w x pu_h is the CU;
 * 4 (2NxnU) / 5 (2NxnD) cut it into an upper and a lower part a quarter / three quarters of the way down, 6 (nLx2N) / 7
 * (nRx2N) into a left and a right part; PU sizes and corner positions follow deriveLeftRightTopIdxGeneral /
 * deriveLeftBottomIdxGeneral (TComDataCU.cpp:1837-1944), and the neighbours are those of mode 1 / mode 2 (the cut moves, the
 * CUs the corner positions fall into do not).  PU rows 2r / 2r + 1 (columns 2c / 2c + 1) are the two parts, as there. */
static int64_t SUF(field_me)(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int part_mode,
                             const meo_result *col, int merange, int method, int subme, int qp, meo_result *out, int16_t *out_mvp)
{
    SUF(state) *st = (SUF(state)*)malloc(sizeof(SUF(state)));
    int64_t evals = 0;
    const int max_cu = 64;
    /* modes 4..7 (AMP): pu_w x pu_h is the CU, cut at a quarter (4: 2NxnU, 6: nLx2N) or three quarters (5: 2NxnD, 7: nRx2N) */
    const int horiz = part_mode == 1 || part_mode == 4 || part_mode == 5;    /* upper / lower parts */
    const int vert = part_mode == 2 || part_mode == 6 || part_mode == 7;     /* left / right parts */
    const int amp = part_mode >= 4;
    int gw = pic_w / pu_w, gh = pic_h / pu_h;
    if (part_mode == 1) gh &= ~1;
    if (part_mode == 2) gw &= ~1;
    if (amp && horiz) gh *= 2;
    if (amp && vert) gw *= 2;
    const int cut_h = part_mode == 4 ? pu_h >> 2 : pu_h - (pu_h >> 2), cut_w = part_mode == 6 ? pu_w >> 2 : pu_w - (pu_w >> 2);
    const int parts = horiz ? 2 : 1;
    for (int row = 0; row < gh / parts; row++)
    {
        for (int gx = 0; gx < gw; gx++)
          for (int part = 0; part < parts; part++)
          {
            const int gy = row * parts + part, lower = part == 1;
            const int xy = gx + gy * gw;
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
            const int have_l = gx > 0, have_a = gy > 0, have_ar = !lower && gy > 0 && gx + 1 < gw, have_al = gy > 0 && gx > 0;
            /* candidate A: the left neighbour; for the upper part of a 2NxN CU the below-left one */
            const int xa = horiz && !lower ? xy + gw - 1 : xy - 1;
            int c[2][2] = { { 0, 0 }, { 0, 0 } };
            int bx = 0, by = 0, have_b = 0;
            if (have_ar)      { bx = out[xy - gw + 1].mvx; by = out[xy - gw + 1].mvy; have_b = 1; }
            else if (have_a)  { bx = out[xy - gw].mvx;     by = out[xy - gw].mvy;     have_b = 1; }
            else if (have_al) { bx = out[xy - gw - 1].mvx; by = out[xy - gw - 1].mvy; have_b = 1; }
            int n_cand = 0;
            if (have_l)
            {
                c[0][0] = out[xa].mvx; c[0][1] = out[xa].mvy;
                n_cand = 1;
                if (have_b && !(bx == c[0][0] && by == c[0][1]))
                {
                    c[1][0] = bx; c[1][1] = by;
                    n_cand = 2;
                }
            }
            else if (have_b)
            {
                c[0][0] = bx; c[0][1] = by;
                n_cand = 1;
            }
            if (col && n_cand < 2)
            {
                /* the temporal candidate (fillMvpCand, TComDataCU.cpp:2635-2697): the co-located motion below-right of the PU
                 * unless that lies outside the picture or in the next CTU row, else (also when that block has none) at the
                 * PU's centre; motion is kept per 16x16 block (xGetColMVP masks the address, :3010).  It goes in unpruned;
                 * the list is then cut to two (:2699-2702), so it counts only while there is room. */
                const int cw = (pic_w + 15) >> 4;
                const int rbx = px + bw, rby = py + bh;
                const meo_result *t = NULL;
                if (rbx < pic_w && rby < pic_h && (rby & (max_cu - 1)) != 0 && col[(rby >> 4) * cw + (rbx >> 4)].cost >= 0)
                    t = &col[(rby >> 4) * cw + (rbx >> 4)];
                else
                {
                    const int ctx = px + ((bw >> 3) << 2), cty = py + ((bh >> 3) << 2);       /* xDeriveCenterIdx, :3104-3116 */
                    if (col[(cty >> 4) * cw + (ctx >> 4)].cost >= 0)
                        t = &col[(cty >> 4) * cw + (ctx >> 4)];
                }
                if (t)
                {
                    c[n_cand][0] = t->mvx; c[n_cand][1] = t->mvy;
                    n_cand++;
                }
            }
            const int xmax = (pic_w + 8 - cux - 1) << 2, xmin = (-max_cu - 8 - cux + 1) << 2;
            const int ymax = (pic_h + 8 - cuy - 1) << 2, ymin = (-max_cu - 8 - cuy + 1) << 2;
            SUF(bind)(st, enc, ref, px, py, bw, bh, qp);
            int mvpx = c[0][0], mvpy = c[0][1];
            if (c[1][0] != c[0][0] || c[1][1] != c[0][1])
            {
                const int t0 = SUF(subpel_cmp)(st, SUF(clipi)(c[0][0], xmin, xmax), SUF(clipi)(c[0][1], ymin, ymax), MEO_SAD);
                const int t1 = SUF(subpel_cmp)(st, SUF(clipi)(c[1][0], xmin, xmax), SUF(clipi)(c[1][1], ymin, ymax), MEO_SAD);
                if (t1 < t0)
                {
                    mvpx = c[1][0]; mvpy = c[1][1];
                }
            }
            evals += st->evals;
            meo_job j;
            memset(&j, 0, sizeof(j));
            j.pu_x = px; j.pu_y = py; j.w = (int16_t)bw; j.h = (int16_t)bh;
            const int cpx = SUF(clipi)(mvpx, xmin, xmax), cpy = SUF(clipi)(mvpy, ymin, ymax);
            j.mvmin_x = (int16_t)((int16_t)SUF(clipi)((int16_t)(cpx - (merange << 2)), xmin, xmax) >> 2);
            j.mvmin_y = (int16_t)((int16_t)SUF(clipi)((int16_t)(cpy - (merange << 2)), ymin, ymax) >> 2);
            j.mvmax_x = (int16_t)((int16_t)SUF(clipi)((int16_t)(cpx + (merange << 2)), xmin, xmax) >> 2);
            j.mvmax_y = (int16_t)((int16_t)SUF(clipi)((int16_t)(cpy + (merange << 2)), ymin, ymax) >> 2);
            j.qmvp_x = (int16_t)mvpx; j.qmvp_y = (int16_t)mvpy;
            j.merange = (int16_t)merange; j.method = (uint8_t)method; j.subme = (uint8_t)subme; j.qp = (uint8_t)qp;
            j.numc = 3;
            if (have_l)  { j.mvc[0][0] = out[xy - 1].mvx;      j.mvc[0][1] = out[xy - 1].mvy; }
            if (have_a)  { j.mvc[1][0] = out[xy - gw].mvx;     j.mvc[1][1] = out[xy - gw].mvy; }
            if (have_ar) { j.mvc[2][0] = out[xy - gw + 1].mvx; j.mvc[2][1] = out[xy - gw + 1].mvy; }
            SUF(bind)(st, enc, ref, px, py, bw, bh, qp);
            out[xy].cost = SUF(search)(st, &j, &out[xy].mvx, &out[xy].mvy);
            evals += st->evals;
            if (out_mvp)
            {
                out_mvp[2 * xy] = (int16_t)mvpx;
                out_mvp[2 * xy + 1] = (int16_t)mvpy;
            }
          }
    }
    free(st);
    return evals;
}

