/* TEST INFRASTRUCTURE ONLY -- see me_oracle.h.  Citations are to /root/reference/source. */
#include "me_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* The MV-cost table is built from float/double expressions whose rounding must match
 * the reference build (g++ -O2, no FMA): forbid contraction here. */
#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC optimize ("fp-contract=off")
#endif

#define MEO_MIN(a, b) ((a) < (b) ? (a) : (b))

/* partition dimensions; Partitions enum = (w_idx << 3) + h_idx (common/primitives.h:77-88) */
static const int g_dims[8] = { 4, 8, 12, 16, 24, 32, 48, 64 };

/* HEVC luma interpolation taps (Lib/TLibCommon/TComRom.cpp:176-182) */
static const int16_t g_luma_filter[4][8] =
{
    {  0, 0,   0, 64,  0,   0, 0,  0 },
    { -1, 4, -10, 58, 17,  -5, 1,  0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    {  0, 1,  -5, 17, 58, -10, 4, -1 }
};

/* lambda^2 for non-I slices, indexed by QP (Lib/TLibCommon/TComRom.cpp:527-536) */
static const double g_lambda2_non_I[52] =
{
    0.05231, 0.060686, 0.07646, 0.096333333, 0.151715667, 0.15292, 0.192666667, 0.242745, 0.382299,
    0.385333333, 0.485489333, 0.611678333, 0.963333333, 0.970979, 1.223356667, 1.541333333, 2.427447667,
    2.446714, 3.082666667, 3.883916667, 6.116785, 6.165333333, 7.767833333, 9.786856667, 15.41333333,
    16.57137733, 22.183542, 29.5936, 39.357022, 52.19656867, 69.05173333, 91.14257667, 150.0651347,
    157.8325333, 207.1422197, 271.4221573, 443.904, 447.4271947, 563.722941, 710.2464, 1118.567987,
    1127.445883, 1420.4928, 1789.70878, 2818.614706, 2840.9856, 3579.41756, 4509.78353, 7102.464,
    7158.83512, 9019.56706, 11363.9424
};

/* common/primitives.cpp:40-54 */
int meo_partition_from_sizes(int w, int h)
{
    int wi = -1, hi = -1;
    for (int i = 0; i < 8; i++)
    {
        if (g_dims[i] == w) wi = i;
        if (g_dims[i] == h) hi = i;
    }
    if (wi < 0 || hi < 0)
        return -1;
    return (wi << 3) + hi;
}

/* ---------------------------------------------------------------- MV cost model */

static float    *g_bitsizes;            /* [0 .. 32768] */
static uint16_t *g_costs[52];           /* 65536 entries each; centre at +32768 */

/* BitCost::CalculateLogs (encoder/bitcost.cpp:73-85) */
static void build_bitsizes(void)
{
    float *b = (float*)malloc(sizeof(float) * 32769);
    b[0] = 0.718f;
    float log2_2 = 2.0f / logf(2.0f);
    for (int i = 1; i <= 32768; i++)
        b[i] = logf((float)(i + 1)) * log2_2 + 1.718f;
    g_bitsizes = b;
}

/* BitCost::setQP (encoder/bitcost.cpp:35-61) */
static const uint16_t *cost_table_for_qp(int qp)
{
    const uint16_t *t;
#pragma omp critical(meo_cost_table)
    {
        if (!g_bitsizes)
            build_bitsizes();
        if (!g_costs[qp])
        {
            uint16_t *c = (uint16_t*)calloc(65536, sizeof(uint16_t));
            double lambda = sqrt(g_lambda2_non_I[qp]);
            for (int i = 0; i < 32768; i++)
            {
                double v = g_bitsizes[i] * lambda + 0.5f;
                uint16_t e = (uint16_t)(v < 65535 ? v : 65535);
                c[32768 + i] = e;
                c[32768 - i] = e;
            }
            g_costs[qp] = c;
        }
        t = g_costs[qp] + 32768;
    }
    return t;
}

void meo_mvcost_table(int qp, uint16_t *out)
{
    memcpy(out, cost_table_for_qp(qp) - 32768, 65536 * sizeof(uint16_t));
}

/* bitcost.h:43-46 */
int meo_mvcost(int qp, int mvpx, int mvpy, int mvx, int mvy)
{
    const uint16_t *c = cost_table_for_qp(qp);
    return (uint16_t)(c[(int16_t)(mvx - mvpx)] + c[(int16_t)(mvy - mvpy)]);
}

/* bitcost.h:49-53 */
int meo_bitcost(int mvpx, int mvpy, int mvx, int mvy)
{
    cost_table_for_qp(32);
    return (uint16_t)(g_bitsizes[(abs(mvx - mvpx) << 1) + !!(mvx < mvpx)] +
                      g_bitsizes[(abs(mvy - mvpy) << 1) + !!(mvy < mvpy)] + 0.5f);
}

/* ---------------------------------------------------------------- per-pixel-type bodies */

#define PIX uint8_t
#define SUF(x) x##_u8
#include "me_oracle_impl.h"
#undef PIX
#undef SUF

#define PIX uint16_t
#define SUF(x) x##_u16
#include "me_oracle_impl.h"
#undef PIX
#undef SUF

/* common/pixel.cpp:614-637 */
void meo_lowres_core_u8(const uint8_t *src0, uint8_t *dst0, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc,
                        intptr_t src_stride, intptr_t dst_stride, int width, int height)
{
    lowres_core_u8(src0, dst0, dsth, dstv, dstc, src_stride, dst_stride, width, height);
}

void meo_lowres_init(const void *src, intptr_t src_stride, int bytes_per_px, void *const dst[4], intptr_t dst_stride,
                     int width, int lines, int mx, int my)
{
    if (bytes_per_px == 1)
        lowres_init_u8((const uint8_t*)src, src_stride, (uint8_t *const*)dst, dst_stride, width, lines, mx, my);
    else
        lowres_init_u16((const uint16_t*)src, src_stride, (uint16_t *const*)dst, dst_stride, width, lines, mx, my);
}

int64_t meo_lookahead_me(const meo_plane *enc, const meo_plane *ref, int cuw, int cuh, int merange,
                         int method, int subme, int qp, meo_result *out)
{
    cost_table_for_qp(qp);
    return enc->bytes_per_px == 1 ? lookahead_me_u8(enc, ref, cuw, cuh, merange, method, subme, qp, out)
                                  : lookahead_me_u16(enc, ref, cuw, cuh, merange, method, subme, qp, out);
}

int64_t meo_field_me_tmvp(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int part_mode,
                          const meo_result *col, int merange, int method, int subme, int qp, meo_result *out, int16_t *out_mvp)
{
    cost_table_for_qp(qp);
    return enc->bytes_per_px == 1 ? field_me_u8(enc, ref, pic_w, pic_h, pu_w, pu_h, part_mode, col, merange, method, subme, qp, out, out_mvp)
                                  : field_me_u16(enc, ref, pic_w, pic_h, pu_w, pu_h, part_mode, col, merange, method, subme, qp, out, out_mvp);
}

int64_t meo_field_me_parts(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int part_mode,
                           int merange, int method, int subme, int qp, meo_result *out, int16_t *out_mvp)
{
    return meo_field_me_tmvp(enc, ref, pic_w, pic_h, pu_w, pu_h, part_mode, NULL, merange, method, subme, qp, out, out_mvp);
}

/* TComDataCU::xGetDistScaleFactor (TComDataCU.cpp:3079-3096) and scaleMv (:45-51) as xGetColMVP applies them (:3066-3074) */
void meo_scale_col_mv(int cur_poc, int cur_ref_poc, int col_poc, int col_ref_poc, int16_t *mvx, int16_t *mvy)
{
    const int d = col_poc - col_ref_poc, b = cur_poc - cur_ref_poc;
    if (d == b)
        return;
    const int tdb = b < -128 ? -128 : b > 127 ? 127 : b, tdd = d < -128 ? -128 : d > 127 ? 127 : d;
    const int x = (0x4000 + abs(tdd / 2)) / tdd;
    int scale = (tdb * x + 32) >> 6;
    scale = scale < -4096 ? -4096 : scale > 4095 ? 4095 : scale;
    int vx = (scale * *mvx + 127 + (scale * *mvx < 0)) >> 8, vy = (scale * *mvy + 127 + (scale * *mvy < 0)) >> 8;
    *mvx = (int16_t)(vx < -32768 ? -32768 : vx > 32767 ? 32767 : vx);
    *mvy = (int16_t)(vy < -32768 ? -32768 : vy > 32767 ? 32767 : vy);
}

int64_t meo_field_me(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int merange,
                     int method, int subme, int qp, meo_result *out, int16_t *out_mvp)
{
    return meo_field_me_parts(enc, ref, pic_w, pic_h, pu_w, pu_h, 0, merange, method, subme, qp, out, out_mvp);
}

int64_t meo_search_batch(const meo_plane *enc, const meo_plane *refs, const meo_job *jobs, int n,
                         meo_result *out, int threads)
{
    int64_t evals = 0;
    if (threads < 1)
        threads = 1;
    cost_table_for_qp(0);   /* make sure g_bitsizes exists before the parallel region */
#pragma omp parallel num_threads(threads) reduction(+ : evals)
    {
        void *mem = NULL;
        if (enc->bytes_per_px == 1)
        {
            state_u8 *st = (state_u8*)(mem = malloc(sizeof(state_u8)));
#pragma omp for schedule(dynamic, 16)
            for (int i = 0; i < n; i++)
            {
                bind_u8(st, enc, &refs[jobs[i].ref_idx], jobs[i].pu_x, jobs[i].pu_y, jobs[i].w, jobs[i].h, jobs[i].qp);
                out[i].cost = search_u8(st, &jobs[i], &out[i].mvx, &out[i].mvy);
                evals += st->evals;
            }
        }
        else
        {
            state_u16 *st = (state_u16*)(mem = malloc(sizeof(state_u16)));
#pragma omp for schedule(dynamic, 16)
            for (int i = 0; i < n; i++)
            {
                bind_u16(st, enc, &refs[jobs[i].ref_idx], jobs[i].pu_x, jobs[i].pu_y, jobs[i].w, jobs[i].h, jobs[i].qp);
                out[i].cost = search_u16(st, &jobs[i], &out[i].mvx, &out[i].mvy);
                evals += st->evals;
            }
        }
        free(mem);
    }
    return evals;
}

int meo_buf_cost(const meo_plane *enc, int pu_x, int pu_y, int w, int h, int kind, const void *fref, intptr_t stride)
{
    int r;
    if (enc->bytes_per_px == 1)
    {
        state_u8 *st = (state_u8*)malloc(sizeof(state_u8));
        bind_u8(st, enc, NULL, pu_x, pu_y, w, h, 0);
        r = blk_cost_u8(kind, st->fenc, 64, (const uint8_t*)fref, stride, w, h);
        free(st);
    }
    else
    {
        state_u16 *st = (state_u16*)malloc(sizeof(state_u16));
        bind_u16(st, enc, NULL, pu_x, pu_y, w, h, 0);
        r = blk_cost_u16(kind, st->fenc, 64, (const uint16_t*)fref, stride, w, h);
        free(st);
    }
    return r;
}

/* cost of the cached PU against the motion-compensated block at q-pel (qx, qy): xPredInterLumaBlk + bufSAD /
 * bufSA8D (Lib/TLibEncoder/TEncSearch.cpp:2944-2958, 2182-2190) = subpelCompare's interpolation */
int meo_mc_cost(const meo_plane *enc, const meo_plane *ref, int pu_x, int pu_y, int w, int h, int qx, int qy, int kind)
{
    int r;
    if (enc->bytes_per_px == 1)
    {
        state_u8 *st = (state_u8*)malloc(sizeof(state_u8));
        bind_u8(st, enc, ref, pu_x, pu_y, w, h, 0);
        r = subpel_cmp_u8(st, qx, qy, kind);
        free(st);
    }
    else
    {
        state_u16 *st = (state_u16*)malloc(sizeof(state_u16));
        bind_u16(st, enc, ref, pu_x, pu_y, w, h, 0);
        r = subpel_cmp_u16(st, qx, qy, kind);
        free(st);
    }
    return r;
}

