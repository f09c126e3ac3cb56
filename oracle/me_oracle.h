/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the x265 motion-estimation hot path.
 *
 * Plain-C restatement of the reference algorithm (block-cost primitives, MV bit-cost
 * table, HEVC luma interpolation, the five integer search methods and the sub-pel
 * refinement of MotionEstimate::motionEstimate).  It is the checker for the HIP path:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library never links, loads or calls anything in this directory.
 *
 * Parity pin: validated bit-for-bit against the reference's own compiled C path
 * (oracle/_ref, built from /root/reference by oracle/Makefile) by
 * tests/test_oracle_vs_reference.py, and against the golden vectors under
 * tests/golden/ that were generated from that compiled reference by
 * oracle/gen_golden.py.
 *
 * Two call-site loops live in translation units that cannot be linked without the whole
 * encoder (encoder/slicetype.cpp, Lib/TLibEncoder/TEncSearch.cpp): the lookahead's neighbour-
 * predicted CU scan and the bidirectional refinement's xPatternSearch / xPatternRefinement.  For
 * those the golden vectors come from the reference's own MotionEstimate / primitives / filters /
 * MV costs driven by a restated loop in oracle/ref_shim.cpp: the arithmetic is pinned, the two
 * short control loops are restated on both sides.
 *
 * Every function cites the reference file:line it restates.
 */
#ifndef ME_ORACLE_H
#define ME_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MEO_MAX_MVC 4

/* cost kinds */
enum { MEO_SAD = 0, MEO_SATD = 1, MEO_SA8D_INTER = 2, MEO_SA8D_SQUARE = 3 };
/* search methods, numbering of x265.h (X265_DIA_SEARCH ... X265_FULL_SEARCH) */
enum { MEO_DIA = 0, MEO_HEX = 1, MEO_UMH = 2, MEO_STAR = 3, MEO_FULL = 4,
       /* not a MotionEstimate method: the search of TEncSearch::xMotionEstimation (bidirectional
        * refinement: exhaustive SAD scan + two 9-point SATD refinements, TEncSearch.cpp:2993-3000) */
       MEO_BIPRED = 5 };

/* One motionEstimate() call, fully described (reference call sequence:
 * setQP; setSearchMethod; setSubpelRefine; setSourcePlane; setSourcePU; motionEstimate). */
typedef struct meo_job
{
    int32_t pu_x, pu_y;          /* PU top-left, pixels; blockOffset = pu_y*stride + pu_x */
    int16_t w, h;                /* PU size, each in {4,8,12,16,24,32,48,64} */
    int16_t mvmin_x, mvmin_y;    /* full-pel, inclusive */
    int16_t mvmax_x, mvmax_y;
    int16_t qmvp_x, qmvp_y;      /* q-pel predictor */
    int16_t merange;
    uint8_t method;              /* 0 DIA, 1 HEX, 2 UMH, 3 STAR, 4 FULL, 5 BIPRED */
    uint8_t subme;               /* 0..7 */
    uint8_t qp;                  /* 0..51 */
    uint8_t numc;                /* number of candidates, <= MEO_MAX_MVC */
    uint8_t ref_idx;             /* which plane of the batch's reference list is searched */
    uint8_t icost;               /* cost kind of the integer stage: 0 SAD (the reference's behaviour);
                                    1 SATD / 2 SA8D_INTER are extensions with no reference counterpart */
    int16_t mvc[MEO_MAX_MVC][2]; /* q-pel candidates */
} meo_job;

typedef struct meo_result
{
    int16_t mvx, mvy;            /* q-pel */
    int32_t cost;
} meo_result;

/* A border-extended luma plane; `pix` points at pixel (0,0). */
typedef struct meo_plane
{
    const void *pix;             /* uint8_t* or uint16_t* */
    intptr_t    stride;          /* in pixels */
    int         bytes_per_px;    /* 1 or 2 */
    int         bit_depth;       /* X265_DEPTH seen by the interpolation filters */
    const void *lowres[4];       /* optional F,H,V,C half-pel planes; non-NULL [0] => isLowres */
} meo_plane;

int  meo_partition_from_sizes(int w, int h);

/* block costs; `idx` is the Partitions enum (kinds 0-2) or SquareBlocks (kind 3) */
int  meo_cmp_u8 (int kind, int idx, const uint8_t  *fenc, intptr_t fencstride, const uint8_t  *fref, intptr_t frefstride);
int  meo_cmp_u16(int kind, int idx, const uint16_t *fenc, intptr_t fencstride, const uint16_t *fref, intptr_t frefstride);
void meo_sad_xn_u8 (int n, int part, const uint8_t  *fenc, const uint8_t  *const *fref, intptr_t frefstride, int *res);
void meo_sad_xn_u16(int n, int part, const uint16_t *fenc, const uint16_t *const *fref, intptr_t frefstride, int *res);

/* out[i + 32768] = lambda-scaled cost of MVD i, i in [-32767, 32767]; out[0] = 0 */
void meo_mvcost_table(int qp, uint16_t *out);
int  meo_mvcost(int qp, int mvpx, int mvpy, int mvx, int mvy);
int  meo_bitcost(int mvpx, int mvpy, int mvx, int mvy);

/* kind 0: H pixel->pixel, 1: V pixel->pixel, 2: H pixel->short, 3: V short->pixel */
void meo_ipfilter_u8 (int kind, const void *src, intptr_t srcStride, void *dst, intptr_t dstStride, int w, int h, int frac, int depth);
void meo_ipfilter_u16(int kind, const void *src, intptr_t srcStride, void *dst, intptr_t dstStride, int w, int h, int frac, int depth);

void meo_lowres_core_u8(const uint8_t *src0, uint8_t *dst0, uint8_t *dsth, uint8_t *dstv, uint8_t *dstc,
                        intptr_t src_stride, intptr_t dst_stride, int width, int height);

/* Lowres::init, picture part (common/lowres.cpp:175-185): the four half-resolution planes of a
 * border-replicated source plane (`src` at pixel (0,0)), borders extended by (mx,my);
 * dst[k] point at pixel (0,0) of the F, H, V, C planes */
void meo_lowres_init(const void *src, intptr_t src_stride, int bytes_per_px, void *const dst[4], intptr_t dst_stride,
                     int width, int lines, int mx, int my);

/* motion-search part of Lookahead::estimateFrameCost for one list (encoder/slicetype.cpp:306-316,
 * 346-407): cuw x cuh 8x8 CUs in reverse raster order with the neighbour-MV predictor chain;
 * enc / ref carry lowres[] planes.  out[cux + cuy*cuw]; returns the number of block-cost evaluations */
int64_t meo_lookahead_me(const meo_plane *enc, const meo_plane *ref, int cuw, int cuh, int merange,
                         int method, int subme, int qp, meo_result *out);

/* the searches of TEncSearch::predInterSearch (Lib/TLibEncoder/TEncSearch.cpp:2341-2455) for a uniform grid of
 * pu_w x pu_h PUs in raster order, predictor and candidates from the already searched neighbours (AMVP list
 * left / first of above-right, above, above-left; the one with the smaller template SAD; xSetSearchRange);
 * out[gx + gy * (pic_w / pu_w)], out_mvp (optional) = the predictor used, (x, y) pairs; returns the evaluations */
int64_t meo_field_me(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int merange,
                     int method, int subme, int qp, meo_result *out, int16_t *out_mvp);
/* the same for the parts of 2NxN (part_mode 1: CUs of pu_w x 2 pu_h) or Nx2N CUs (2: CUs of 2 pu_w x pu_h), 0 = as above:
 * CUs in raster order, their two parts one after the other; whole CUs only (an odd last PU row / column is left out).
 * part_mode 4 / 5 / 6 / 7 = the asymmetric partitions 2NxnU / 2NxnD / nLx2N / nRx2N of CUs of pu_w x pu_h (numbered as PartSize,
 * Lib/TLibCommon/TypeDef.h:100-111): PU rows 2r / 2r + 1 are the upper / lower parts (4, 5: 2 (pic_h / pu_h) PU rows), PU columns
 * 2c / 2c + 1 the left / right parts (6, 7: 2 (pic_w / pu_w) PUs per row) */
int64_t meo_field_me_parts(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int part_mode,
                           int merange, int method, int subme, int qp, meo_result *out, int16_t *out_mvp);
/* the same with the temporal AMVP candidate (TComDataCU::fillMvpCand, Lib/TLibCommon/TComDataCU.cpp:2635-2697): col = the co-located
 * picture's motion per 16x16 block, ((pic_w + 15) / 16) x ((pic_h + 15) / 16) entries in raster order, scaled already to this
 * picture's reference distance (meo_scale_col_mv), cost < 0 where the block has none; it is taken below-right of the PU, else at
 * the PU's centre, and joins the list while that holds fewer than two spatial candidates.  col == NULL: as above. */
int64_t meo_field_me_tmvp(const meo_plane *enc, const meo_plane *ref, int pic_w, int pic_h, int pu_w, int pu_h, int part_mode,
                          const meo_result *col, int merange, int method, int subme, int qp, meo_result *out, int16_t *out_mvp);
/* the scaling of a co-located MV by the POC distances (xGetDistScaleFactor :3079-3096, scaleMv :45-51), in place */
void meo_scale_col_mv(int cur_poc, int cur_ref_poc, int col_poc, int col_ref_poc, int16_t *mvx, int16_t *mvy);

/* run n jobs (one enc plane, jobs[i].ref_idx selects refs[]) with `threads` OpenMP threads
 * (one search state per thread); returns total number of block-cost evaluations. */
int64_t meo_search_batch(const meo_plane *enc, const meo_plane *refs, const meo_job *jobs, int n,
                         meo_result *out, int threads);

/* cost of cached PU against an arbitrary block (bufSAD/bufSATD/bufSA8D, motion.h:89-93) */
int  meo_buf_cost(const meo_plane *enc, int pu_x, int pu_y, int w, int h, int kind, const void *fref, intptr_t stride);

/* cost of the PU against the motion-compensated (interpolated) block at q-pel (qx, qy) of ref */
int  meo_mc_cost(const meo_plane *enc, const meo_plane *ref, int pu_x, int pu_y, int w, int h, int qx, int qy, int kind);

#ifdef __cplusplus
}
#endif
#endif
