/* x265_amd -- MI355X (gfx950) motion-estimation engine: C ABI.
 *
 * This is the drop-in boundary for the motion-estimation hot path of x265.  Every entry
 * point names the reference interface it stands in for (paths relative to the reference's
 * source/ directory).  Plain pointers and sizes only; no C++ or torch types.
 *
 * All compute entry points run hand-written HIP kernels on the GPU.  There is no CPU
 * fallback: every call returns X265AMD_ERR_NO_DEVICE / a HIP error code when the device or
 * the kernels are unavailable.
 *
 * Threading: a context owns one HIP stream.  Calls on one context are serialised by an
 * internal mutex; use one context per host thread for concurrency (the reference uses one
 * MotionEstimate per CTU row, encoder/cturow.h:56).
 */
#ifndef X265_AMD_H
#define X265_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define X265AMD_MAX_MVC     4     /* candidates per job (lookahead passes at most 4: encoder/slicetype.cpp:382) */
#define X265AMD_MAX_REFS    16    /* reference planes addressable by one batch */
#define X265AMD_FENC_STRIDE 64    /* common/primitives.h:34 */

/* error codes (negative).  The reference has no error convention for this path
 * (int/void returns and assert(), common/primitives.cpp:49-50); batch calls here return 0 or: */
enum
{
    X265AMD_OK             = 0,
    X265AMD_ERR_NO_DEVICE  = -1,   /* no HIP device / kernels not loadable */
    X265AMD_ERR_INVALID    = -2,   /* bad argument (size not a partition, plane id, qp, ...) */
    X265AMD_ERR_NOMEM      = -3,
    X265AMD_ERR_HIP        = -4,   /* a HIP runtime call failed; see x265amd_last_error() */
    X265AMD_ERR_RANGE      = -5    /* a job's search window reaches outside the registered plane margins */
};

/* cost kinds: EncoderPrimitives::sad / satd / sa8d_inter (common/primitives.h:242-249) */
enum { X265AMD_SAD = 0, X265AMD_SATD = 1, X265AMD_SA8D = 2 };

/* search methods: X265_DIA_SEARCH .. X265_FULL_SEARCH (x265.h, --me 0..4) */
enum { X265AMD_DIA = 0, X265AMD_HEX = 1, X265AMD_UMH = 2, X265AMD_STAR = 3, X265AMD_FULL = 4,
       /* Not a MotionEstimate method: the search of TEncSearch::xMotionEstimation, the bidirectional
        * refinement (Lib/TLibEncoder/TEncSearch.cpp:2960-3005).  The job's encoder plane holds the
        * caller's modified block (2*org - otherPred, TComYuv::removeHighFreq) at the PU position;
        * [mvmin, mvmax] is xSetSearchRange's result around the uni-directional MV (:2983), qmvp the
        * predictor of m_me.setMVP (:2991).  Runs xPatternSearch (exhaustive SAD + mvcost, :3026-3085)
        * and xPatternRefinement at half-pel and quarter-pel (9-point SATD + mvcost, :224-285);
        * result = (outmv in q-pel, the quarter-pel refinement's cost, :3000).  merange, subme, numc,
        * mvc are ignored; the range must not be empty. */
       X265AMD_BIPRED = 5 };

typedef struct x265amd_ctx x265amd_ctx;

/* ---- context ------------------------------------------------------------------------- */

/* Create a context on HIP device `device` (one stream, MV-cost tables built lazily per QP as
 * BitCost::setQP does, encoder/bitcost.cpp:35-61). */
int  x265amd_create(int device, x265amd_ctx **out);
void x265amd_destroy(x265amd_ctx *ctx);
/* text of the last error on this context (never NULL) */
const char *x265amd_last_error(const x265amd_ctx *ctx);
/* wait for all work queued on the context: every stream set, asynchronous result downloads and plane refills */
int  x265amd_sync(x265amd_ctx *ctx);
/* the hipStream_t the context launches on (for callers that time or order against it) */
void *x265amd_stream(x265amd_ctx *ctx);
/* developer aid: one line of text saying which of the context's streams and fork / join events are still
 * busy; takes no lock, so a watchdog thread may call it while another thread waits in x265amd_sync */
int  x265amd_debug_state(x265amd_ctx *ctx, char *buf, size_t n);
/* GPU-side ordering against other streams (a collective's, a copy's) without a host wait: work queued
 * on the context after x265amd_wait_event starts only when `event` (a hipEvent_t recorded on the other
 * stream) has completed; x265amd_record_event records `event` behind everything queued on the context
 * so far.  The reference's counterpart is the row-count hand-over between the frame that reconstructs a
 * reference and the frames that search it (m_reconRowCount, encoder/framefilter.cpp:257,
 * encoder/frameencoder.cpp:906-918). */
int  x265amd_wait_event(x265amd_ctx *ctx, void *event);
int  x265amd_record_event(x265amd_ctx *ctx, void *event);
/* the same against the main stream of stream set `set` (x265amd_joblist_run_overlapped): orders one frame in flight
 * against its own reference hand-over without holding up the frames on the other sets */
int  x265amd_wait_event_on(x265amd_ctx *ctx, void *event, int set);
int  x265amd_record_event_on(x265amd_ctx *ctx, void *event, int set);

/* ---- planes --------------------------------------------------------------------------
 * A plane is a border-extended luma picture as TComPicYuv lays it out
 * (Lib/TLibCommon/TComPicYuv.cpp:68-120): `base` is the first byte of the padded
 * allocation, pixel (0,0) sits at base + (margin_y*stride + margin_x) pixels, stride in
 * pixels.  This is what ReferencePlanes::fpelPlane / lumaStride (common/reference.h:36-53)
 * and MotionEstimate::setSourcePlane (encoder/motion.h:78-82) carry. */

/* copy a host plane to the device (H2D on the context's stream); borrowed only during the call */
int x265amd_plane_upload(x265amd_ctx *ctx, const void *host_base, int width, int height,
                         int margin_x, int margin_y, intptr_t stride, int bytes_per_px, int bit_depth,
                         int *plane_id);
/* wrap a plane that already lives in device memory (e.g. filled by an RCCL broadcast);
 * the memory is borrowed until x265amd_plane_release */
int x265amd_plane_wrap_device(x265amd_ctx *ctx, void *dev_base, int width, int height,
                              int margin_x, int margin_y, intptr_t stride, int bytes_per_px, int bit_depth,
                              int *plane_id);
/* Reference planes that become valid row by row.  The encoder publishes a reconstructed picture
 * one CTU row at a time (FrameFilter: deblock/SAO the row, extend its borders, m_reconRowCount++,
 * encoder/framefilter.cpp:201-262) and other frames may search the rows already published
 * (refLagRows, encoder/frameencoder.cpp:893-918).  x265amd_plane_create makes an empty device plane of
 * the TComPicYuv geometry; x265amd_plane_update_rows copies picture rows [y0, y0 + n_rows) -- the
 * width interior pixels of each, from a host picture whose pixel (0,0) is `host_origin` -- and extends
 * their borders on the device (left / right margins; the top margin with row 0, the bottom margin with
 * the last row), so the margins need not be valid (or exist) on the host.  Searches queued afterwards
 * see the rows; keeping a job's window inside the published rows is the caller's business, as it is in
 * the reference (TEncSearch::xSetSearchRange's m_refLagPixels clamp, TEncSearch.cpp:3021-3023). */
int x265amd_plane_create(x265amd_ctx *ctx, int width, int height, int margin_x, int margin_y, intptr_t stride,
                         int bytes_per_px, int bit_depth, int *plane_id);
int x265amd_plane_update_rows(x265amd_ctx *ctx, int plane_id, const void *host_origin, intptr_t host_stride,
                              int y0, int n_rows);
/* Refill a registered plane (uploaded, created or wrapped) with a new picture of the same geometry,
 * asynchronously: `host_base` addresses the first byte of a padded plane in page-locked host memory, which
 * must stay valid until x265amd_sync.  The copy runs on an upload stream of the library: behind every
 * search queued BEFORE the call (they may still read the old picture), beside everything queued after it
 * that does not name this plane; a search that names the plane starts behind the copy.  With two sets of
 * planes an encoder uploads picture k + 1 while picture k is searched - the reference's frame-parallel
 * hand-over of reconstructed pictures (encoder/frameencoder.cpp:893-918) in picture units.  Pageable
 * memory works too (the runtime stages it and the call may block). */
int x265amd_plane_refill_async(x265amd_ctx *ctx, int plane_id, const void *host_base);
/* geometry a plane was registered with (TComPicYuv::getWidth / getHeight / getLumaMarginX / getStride,
 * Lib/TLibCommon/TComPicYuv.h); any pointer may be NULL */
int x265amd_plane_geometry(x265amd_ctx *ctx, int plane_id, int *width, int *height, int *margin_x, int *margin_y,
                           intptr_t *stride, int *bytes_per_px, int *bit_depth);
/* device address of the padded allocation of a plane (upload target / broadcast buffer) */
void *x265amd_plane_device_ptr(x265amd_ctx *ctx, int plane_id);
int x265amd_plane_release(x265amd_ctx *ctx, int plane_id);

/* copy plane `which` of a registered plane (0; 0..3 = F, H, V, C for a half-resolution set) back
 * to the host: `bytes` from the start of its padded allocation (tests, debugging) */
int x265amd_plane_download(x265amd_ctx *ctx, int plane_id, int which, void *host, size_t bytes);

/* ---- lookahead: half-resolution planes --------------------------------------------------
 * Lowres::create + the picture part of Lowres::init (common/lowres.cpp:30-58, 132-185): the four
 * half-resolution planes of a registered full-resolution plane -- F (2x2 means) and the half-pel
 * shifted H, V, C (frame_init_lowres_core, common/pixel.cpp:614-637) -- computed and
 * border-extended on the device, in one allocation, with the source's margins and
 * stride = (width/2 + 2*margin_x) rounded up to 32; width and lines are rounded up to whole 8x8
 * CUs.  The source's margins must hold replicated borders and be >= 16 pixels.  The returned id
 * is a plane id: as a job's reference it stands for ReferencePlanes with isLowres set
 * (common/reference.h:36-53), so sub-pel candidates take the half-pel-plane averaging branch of
 * subpelCompare (encoder/motion.cpp:1112-1137); as an encoder plane it is its F plane
 * (Lowres::lowresPlane[0], encoder/slicetype.cpp:306).  Release with x265amd_plane_release. */
int x265amd_lowres_create(x265amd_ctx *ctx, int src_plane, int *lowres_plane);
/* Lowres::init for a new picture into an existing set of the same geometry (no allocation);
 * asynchronous on the context's stream */
int x265amd_lowres_update(x265amd_ctx *ctx, int lowres_plane, int src_plane);
/* Lowres::width / lines / lumaStride and Lookahead::widthInCU / heightInCU
 * (common/lowres.cpp:35-46, encoder/slicetype.cpp:74-75); any pointer may be NULL */
int x265amd_lowres_geometry(x265amd_ctx *ctx, int lowres_plane, int *width, int *lines, intptr_t *stride,
                            int *cu_w, int *cu_h);

/* ---- 8-tap luma interpolation ------------------------------------------------------------
 * One call of EncoderPrimitives::ipfilter_pp[FILTER_H_P_P_8] (kind 0), ipfilter_pp[FILTER_V_P_P_8]
 * (kind 1), ipfilter_ps[FILTER_H_P_S_8] (kind 2: pixel -> short) or ipfilter_sp[FILTER_V_S_P_8]
 * (kind 3: short -> pixel) (common/primitives.h:186-188, 263-265; bodies common/ipfilter.cpp:91-139,
 * 354-425, 265-309, 38-89) with the taps g_lumaFilter[frac] (Lib/TLibCommon/TComRom.cpp:176-182) on HOST
 * pointers: src / dst address element (0,0) of the block, strides in elements, and the source must be
 * readable 3 elements before and 4 behind the block in the filtered direction, as in the reference.
 * `bit_depth` plays the role of X265_DEPTH (headRoom = 14 - bit_depth).  The block is staged to the GPU
 * and filtered by the device functions the motion search interpolates with (subpelCompare's H, V and
 * H-then-V chains), so this entry point is how those are checked against the reference's filters one to
 * one; it is not a performance path. */
int x265amd_ipfilter(x265amd_ctx *ctx, int kind, const void *src, intptr_t src_stride, void *dst, intptr_t dst_stride,
                     int w, int h, int frac, int bytes_per_px, int bit_depth);

/* ---- batched block costs -------------------------------------------------------------
 * n independent evaluations of EncoderPrimitives::sad / satd / sa8d_inter
 * [PartitionFromSizes(w,h)] (common/primitives.h:242-249, common/pixel.cpp:645-826):
 * out[i] = cost(enc block at (enc_x,enc_y), ref block at (ref_x,ref_y)). */
typedef struct x265amd_block_pair
{
    int32_t enc_plane, enc_x, enc_y;
    int32_t ref_plane, ref_x, ref_y;
} x265amd_block_pair;

int x265amd_cost_batch(x265amd_ctx *ctx, int kind, int w, int h, int n,
                       const x265amd_block_pair *pairs, int32_t *out);

/* n costs of motion-compensated blocks: out[i] = sad / satd / sa8d_inter of the w x h PU at (pu_x, pu_y) of
 * enc_plane against the prediction block of the same position displaced by the q-pel MV in ref_plane,
 * interpolated like TComPrediction::xPredInterLumaBlk (Lib/TLibCommon/TComPrediction.cpp:464-496).  With
 * X265AMD_SAD this is the distortion term of TEncSearch::xGetTemplateCost (the AMVP candidate check,
 * Lib/TLibEncoder/TEncSearch.cpp:2944-2958: xPredInterLumaBlk + m_me.bufSAD), with X265AMD_SA8D that of
 * the uni-directional merge candidates and xGetInterPredictionError (:2182-2190: motionCompensation +
 * m_me.bufSA8D).  The caller clips the MVs (TComDataCU::clipMv) as those call sites do. */
typedef struct x265amd_mc_pair
{
    int32_t enc_plane, pu_x, pu_y;
    int32_t ref_plane, qmv_x, qmv_y;
} x265amd_mc_pair;

int x265amd_mc_cost_batch(x265amd_ctx *ctx, int kind, int w, int h, int n,
                          const x265amd_mc_pair *pairs, int32_t *out);

/* The source blocks of bidirectional-refinement searches, made on the device: for each of the n w x h PUs,
 *   dst_plane[PU] = ClipY(2 * enc_plane[PU] - pred),  pred = the motion-compensated prediction of the PU from
 * ref_plane at the q-pel MV (the OTHER list's best uni-directional MV),
 * which is what TEncSearch::xMotionEstimation prepares on the host before a bi-predictive search
 * (fencYuv->copyPartToPartYuv + TComYuv::removeHighFreq against m_predYuv[1 - list],
 * Lib/TLibEncoder/TEncSearch.cpp:2970-2980, Lib/TLibCommon/TComYuv.cpp:673-702; the prediction as
 * TComPrediction::xPredInterLumaBlk makes it, Lib/TLibCommon/TComPrediction.cpp:464-496).  dst_plane is a
 * registered full-resolution plane of the pictures' geometry (e.g. from x265amd_plane_create); X265AMD_BIPRED
 * jobs queued afterwards with it as their encoder plane search the blocks without a host round trip.
 * Asynchronous on the context's stream; `pairs` is borrowed only during the call. */
int x265amd_bipred_source_batch(x265amd_ctx *ctx, int dst_plane, int w, int h, int n, const x265amd_mc_pair *pairs);

/* ---- batched motion search ------------------------------------------------------------
 * One job == one MotionEstimate::motionEstimate call (encoder/motion.h:95,
 * encoder/motion.cpp:277-866) preceded by setQP / setSearchMethod / setSubpelRefine /
 * setSourcePU (encoder/motion.h:72-84).  (pu_x,pu_y) replaces the reference's linear
 * `offset` argument: offset = pu_y*stride + pu_x. */
typedef struct x265amd_me_job
{
    int32_t pu_x, pu_y;
    int16_t w, h;                       /* each in {4,8,12,16,24,32,48,64} */
    int16_t mvmin_x, mvmin_y;           /* full-pel, inclusive (const MV& mvmin) */
    int16_t mvmax_x, mvmax_y;
    int16_t qmvp_x, qmvp_y;             /* q-pel predictor (const MV& qmvp) */
    int16_t merange;
    uint8_t method;                     /* X265AMD_DIA .. X265AMD_FULL, or X265AMD_BIPRED */
    uint8_t subme;                      /* 0..7, workload[] of encoder/motion.cpp:51-60 */
    uint8_t qp;                         /* 0..51 */
    uint8_t numc;                       /* numCandidates, <= X265AMD_MAX_MVC */
    uint8_t ref_idx;                    /* index into the batch's ref_planes[] */
    uint8_t icost;                      /* integer-stage cost kind: X265AMD_SAD is the reference's
                                           behaviour; SATD / SA8D are extensions */
    int16_t mvc[X265AMD_MAX_MVC][2];    /* q-pel candidates (const MV* mvc) */
} x265amd_me_job;

typedef struct x265amd_me_result
{
    int16_t mvx, mvy;                   /* outQMv, q-pel */
    int32_t cost;                       /* return value of motionEstimate() */
} x265amd_me_result;

/* jobs / out in host memory; blocks until the results are in `out`.  X265AMD_ERR_RANGE if a job's
 * bounds (+3 pixels of pattern overshoot, +3/+4 of filter apron) leave the reference plane's margins.
 * Job lists are checked the same way when they run (once per plane geometry).  The device-array forms
 * cannot check on the host (their jobs are not visible to it): there the kernels check every job's
 * FIELDS (sizes, method, subme, qp, numc, ref_idx, icost, merange; the same predicate as the host
 * entry points) and give a job that fails the result (MV 0, cost X265AMD_ERR_INVALID) without
 * searching it; the WINDOW of a job is not checked against the plane margins, pixel addresses are
 * clamped into the plane allocation instead, so such a job measures other pixels than the reference
 * would, it does not fault. */
int x265amd_me_search_batch(x265amd_ctx *ctx, int enc_plane, const int *ref_planes, int n_refs,
                            const x265amd_me_job *jobs, int n, x265amd_me_result *out);

/* jobs / out already in device memory; asynchronous on the context's stream.
 * The caller guarantees the buffers stay valid until x265amd_sync().
 * The host never sees these jobs, so the work a job list plans on the host is done by small kernels here: the
 * FULL-search SAD jobs on the squares of each 64x64 cell's quadtree (8-bit planes) are bundled on the device and share
 * one pass over the cell's window, as in a planned list (a 1080p FULL +-64 frame takes about 1.15 x the planned list's
 * time); every other exhaustive job is scanned on its own. */
int x265amd_me_search_batch_device(x265amd_ctx *ctx, int enc_plane, const int *ref_planes, int n_refs,
                                   const x265amd_me_job *d_jobs, int n, x265amd_me_result *d_out);

/* The same search for callers that must not allocate device memory while they call it (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 * Identical results to x265amd_me_search_batch_device; the per-job scratch of the exhaustive kernel
 * (partial winners, work items) lives in a fixed device array inside the library instead of buffers
 * grown on demand (9.5 MiB per GPU; it holds the device-side planner's tables for pictures of up to 1024 64x64 cells
 * times reference planes - 1080p with two references - and larger pictures are searched without bundles), so:
 * n <= 65536 jobs per call (X265AMD_ERR_INVALID beyond; split the batch); the call
 * performs no hipMalloc / hipFree; all contexts on one GPU share that GPU's array: the library orders
 * every such call behind the previous one on that GPU with a stream event (created at x265amd_create),
 * so calls from several contexts or threads are serialised on the GPU rather than racing. */
int x265amd_me_search_batch_device_noalloc(x265amd_ctx *ctx, int enc_plane, const int *ref_planes, int n_refs,
                                           const x265amd_me_job *d_jobs, int n, x265amd_me_result *d_out);

/* Planned job lists.  An encoder searches the same PU geometry frame after frame; a job list
 * keeps the jobs in device memory and plans the exhaustive kernel once (PUs of one size that
 * share a 32x32 cell share one staged search window; large PUs are cut into candidate-row slabs).
 * Results are identical to x265amd_me_search_batch on the same jobs.  The reference analogue is
 * the per-slice set-up before the per-PU calls (Lib/TLibEncoder/TEncSearch.cpp:2332-2364). */
typedef struct x265amd_joblist x265amd_joblist;
int x265amd_joblist_create(x265amd_ctx *ctx, const x265amd_me_job *jobs, int n, int n_refs, int bytes_per_px,
                           x265amd_joblist **out);
/* run the list against planes; out == NULL: asynchronous, results stay in device memory.
 * X265AMD_ERR_INVALID if n_refs is smaller than the n_refs the list was created with,
 * X265AMD_ERR_RANGE as for x265amd_me_search_batch */
int x265amd_joblist_run(x265amd_ctx *ctx, x265amd_joblist *list, int enc_plane, const int *ref_planes, int n_refs,
                        x265amd_me_result *out);
/* Several frames in flight.  x265amd_joblist_run(out = NULL) on stream set `set` of the context.  The context owns four
 * streams, each on a hardware queue of its own: set 0 = the main stream and its side stream, which every other call
 * uses; set 1 = the other two; sets 2 and 3 = those pairs with main and side stream exchanged; sets 4 .. 7 = the main
 * stream of set 0 .. 3 alone (the launches of one batch then follow each other instead of running side by side).
 * Runs queued on different sets execute side by side, so the workgroups of one frame's search fill the compute units
 * that the end of another frame's kernels leaves idle (MI355X, 1080p: exhaustive frames 0.50 ms one at a time, 0.43
 * on sets 0 / 1 in turn, 0.405 on sets 4 .. 7 in turn; HEX + subme 5 frames 0.153 -> 0.098 on sets 0 / 1).  Each frame
 * in flight needs its own list (a list's next run, on any set, starts behind its last one).  x265amd_sync waits for all
 * sets; x265amd_joblist_results / _results_async / _eval_counts wait for the list's last run wherever it ran; a pointer
 * from x265amd_joblist_results_device is valid after x265amd_sync.  The kernel times of x265amd_set_timing are those of
 * runs one at a time only.  Planes named by the call must be complete: a refill queued with x265amd_plane_refill_async is
 * waited for, any other write to a plane (x265amd_plane_update_rows, a caller's own kernels or collective on a wrapped
 * plane) is ordered by the caller (x265amd_wait_event_on, x265amd_sync) - work queued on one set is NOT awaited by a run
 * on another.  The reference runs the motion searches of several frames side by side on its frame threads
 * (encoder/encoder.cpp:199-203 `frameNumThreads`, encoder/frameencoder.cpp:285-300 FrameEncoder::threadMain ->
 * compressFrame: one thread per frame in flight, each reaching TEncSearch::predInterSearch). */
int x265amd_joblist_run_overlapped(x265amd_ctx *ctx, x265amd_joblist *list, int enc_plane, const int *ref_planes, int n_refs,
                                   int set);
const x265amd_me_result *x265amd_joblist_results_device(const x265amd_joblist *list);
/* wait for the list's last run and copy its results (list length entries) to `out` in host memory */
int x265amd_joblist_results(x265amd_ctx *ctx, x265amd_joblist *list, x265amd_me_result *out);
/* queue the copy of the list's results to `out` behind its last run and return at once (page-locked,
 * device-mapped memory is written by a small kernel on the context's stream; other memory by a DMA copy
 * on a stream of its own, which only the next run of THIS list waits for).  The caller reads `out` after
 * x265amd_sync.  `out` should be
 * page-locked host memory (hipHostMalloc / a pinned tensor); with pageable memory the runtime stages the
 * copy and the call may block.  This is how a frame's MV field leaves the GPU while the next frame is
 * searched (the reference hands a PU's result back by return value, encoder/motion.cpp:863-865). */
int x265amd_joblist_results_async(x265amd_ctx *ctx, x265amd_joblist *list, x265amd_me_result *out);
/* Reorder the list's launches so that the jobs that took the most block-cost evaluations in its last
 * run start first (results do not depend on the order; a launch ends with its longest searches, and
 * consecutive frames move alike).  Blocks until the last run has finished.  No counterpart in the
 * reference, whose CTU rows are load-balanced by its thread pool (common/threadpool.cpp). */
int x265amd_joblist_rebalance(x265amd_ctx *ctx, x265amd_joblist *list);
/* out[i] = block-cost evaluations job i took in the list's last run (waits for it) */
int x265amd_joblist_eval_counts(x265amd_ctx *ctx, x265amd_joblist *list, uint32_t *out);
/* the same reordering from lengths the caller predicts: length[i] for job i, any unit (8 groups, longest first) */
int x265amd_joblist_order_by_length(x265amd_ctx *ctx, x265amd_joblist *list, const uint32_t *length);
/* The same reordering for a NEW picture, from the lookahead's half-resolution motion field of that picture
 * (x265amd_lookahead_me's output for the (picture, reference) pair; cu_w x cu_h entries, host memory): a
 * pattern search is long when the block moved far from its predictor, and the field says how far
 * (2 x the half-resolution MV against the job's qmvp).  Results do not depend on the order.  In the
 * reference the lookahead runs ahead of the encode for its own reasons (encoder/slicetype.cpp); its field
 * is free information for the scheduler here.  Blocks until the list is idle. */
int x265amd_joblist_order_by_field(x265amd_ctx *ctx, x265amd_joblist *list, const x265amd_me_result *field, int cu_w, int cu_h);
int x265amd_joblist_destroy(x265amd_ctx *ctx, x265amd_joblist *list);

/* ---- lookahead: whole-frame motion search ------------------------------------------------
 * The motion-search part of Lookahead::estimateFrameCost / estimateCUCost for one reference list
 * (encoder/slicetype.cpp:306-316, 346-369, 381-407) for n_pairs (encoder frame, reference frame)
 * pairs of half-resolution sets of one geometry: every 8x8 CU, in an order equivalent to the
 * reference's reverse raster scan, searched with mvp = median of / candidates = the MVs of its
 * right, lower, lower-left and lower-right neighbours and bounds that keep the block within 8
 * pixels of the picture.  The reference's lookahead uses method HEX, subme 1, merange 16,
 * qp X265_LOOKAHEAD_QP (encoder/slicetype.cpp:70-73).
 * out[pair * cu_w*cu_h + cux + cuy*cu_w] = (lowresMvs, lowresMvCosts) of that CU.
 * out == NULL: asynchronous, results stay in device memory (x265amd_lookahead_results_device). */
int x265amd_lookahead_me(x265amd_ctx *ctx, int n_pairs, const int *enc_lowres, const int *ref_lowres,
                         int merange, int method, int subme, int qp, x265amd_me_result *out);
/* The device array holding the last call's fields.  The pointer stays valid until the next
 * x265amd_lookahead_me call on the context (which may re-allocate it) or x265amd_destroy: consume
 * (or copy) an asynchronous call's results before the next call of the same kind. */
const x265amd_me_result *x265amd_lookahead_results_device(const x265amd_ctx *ctx);
/* Waits for the lookahead and field searches queued so far and reports whether every asynchronous call
 * since the last x265amd_lookahead_status completed: 0, or X265AMD_ERR_HIP if a row of a lookahead
 * search gave up waiting for the row below it, or a row of a field search for the row above it (that
 * call's field is then not valid; x265amd_last_error says which kind).  Failures of the two kinds are
 * kept in separate words that only this call clears, so a later call does not hide an earlier failure.
 * The synchronous forms (out != NULL) perform this check themselves. */
int x265amd_lookahead_status(x265amd_ctx *ctx);

/* ---- whole-frame motion field with the encoder's predictor chain -------------------------------
 * The searches TEncSearch::predInterSearch (Lib/TLibEncoder/TEncSearch.cpp:2341-2455) would run for a
 * picture cut into a uniform grid of pu_w x pu_h PUs (2Nx2N PUs of CUs of that size, those fully inside
 * the picture), each against `ref_plane`, in raster order, every search centred on a predictor derived ON
 * THE DEVICE from the final MVs of the PU's already searched neighbours:
 *   - AMVP candidates: A = left MV; B = the first that exists of the above-right, above, above-left MV;
 *     B dropped if equal to A; padded with zero vectors to two (xEstimateMvPredAMVP's list, without the
 *     temporal candidate);
 *   - mvp = the candidate with the smaller xGetTemplateCost (:2944-2958): SAD of the PU against the
 *     motion-compensated (interpolated) block at the candidate after TComDataCU::clipMv; ties keep A;
 *   - [mvmin, mvmax] = xSetSearchRange(mvp, merange) (:3007-3024, no frame-parallel row clamp);
 *   - motionEstimate(ref, mvmin, mvmax, mvp, 3, {left, above, above-right MV or zero}, merange, ...)
 *     (:2429 / :2445; m_mvPredictors of :2366-2368).
 * out[gx + gy * (width / pu_w)] = (outQMv, return value) of that PU; out_mvp (optional) receives the
 * predictor each search used as (x, y) int16 pairs - the key under which a call site finds the result.
 * Reference planes need margins of at least 78 pixels (X265AMD_ERR_RANGE otherwise).
 * out == NULL: asynchronous, results stay in device memory (x265amd_field_results_device; valid until the
 * next x265amd_field_me on the context; x265amd_lookahead_status reports a row that gave up waiting). */
int x265amd_field_me(x265amd_ctx *ctx, int enc_plane, int ref_plane, int pu_w, int pu_h, int merange, int method,
                     int subme, int qp, x265amd_me_result *out, int16_t *out_mvp);
/* The same for n_fields (encoder plane, reference plane) pairs of one picture size and pixel type in one launch -
 * e.g. the fields of a frame against each of its references (the loop over lists and references around
 * TEncSearch.cpp:2425-2451).  A field's time is set by its chain of dependent PUs, not by throughput, so the
 * fields of one call run side by side at about the time of one.  out / out_mvp hold field f at
 * [f * (width / pu_w) * (height / pu_h) ...). */
int x265amd_field_me_batch(x265amd_ctx *ctx, int n_fields, const int *enc_planes, const int *ref_planes, int pu_w, int pu_h,
                           int merange, int method, int subme, int qp, x265amd_me_result *out, int16_t *out_mvp);
/* The same with a PU size and a partition mode PER FIELD, e.g. one field per quadtree depth of a frame, or the 2NxN and
 * Nx2N fields beside the 2Nx2N one (the PartSize loop around predInterSearch in TEncCu::xCompressCU, Lib/TLibEncoder/TEncCu.cpp).
 * All fields of a call are in flight side by side - fields whose PUs fall into one kernel class (at most 256 / 1024 / 4096 pixels)
 * share a launch whatever their size and mode, and the up to three launches run on different streams - so the call takes about as
 * long as its longest chain (MI355X, 1080p: the twenty fields of the four square sizes, the 2NxN / Nx2N parts of 16 and 32 and all
 * asymmetric parts of 16, 32 and 64 in one call 8.8 ms; the slowest of them alone, 16x8 parts, 8.1 ms).
 *   part_mode[f] (NULL: all 0) = 0: 2Nx2N - every PU is its own CU;
 *                                1: 2NxN - CUs of pu_w x 2 pu_h, PU rows 2r / 2r + 1 are their upper / lower parts;
 *                                2: Nx2N - CUs of 2 pu_w x pu_h, PU columns 2c / 2c + 1 their left / right parts;
 *                                4 / 5: 2NxnU / 2NxnD - CUs of pu_w x pu_h (square, 16, 32 or 64), cut a quarter / three quarters
 *                                   of the way down: PU rows 2r / 2r + 1 are the upper / lower parts (16x4 + 16x12 ... 64x48 + 64x16);
 *                                6 / 7: nLx2N / nRx2N - the same across: PU columns 2c / 2c + 1 are the left / right parts
 *                                   (the numbers are PartSize's, Lib/TLibCommon/TypeDef.h:100-111; 3, NxN, is not an inter mode of
 *                                   CUs above the smallest and is rejected).
 * CUs are taken in raster order, the two parts of a CU one after the other (the partIdx loop, TEncSearch.cpp:2341).  Against
 * mode 0: clipMv and xSetSearchRange work from the CU's position; the upper part of a 2NxN CU takes candidate A from its
 * below-left neighbour (the lower part of the CU to its left, which the left scan of TComDataCU::fillMvpCand asks first,
 * Lib/TLibCommon/TComDataCU.cpp:2574-2588); its lower part has the upper part above it and nothing coded above-right.
 * The asymmetric parts have the neighbours of the 2NxN / Nx2N parts (the cut moves, the CUs that fillMvpCand's corner positions
 * fall into do not); their sizes and corners are deriveLeftRightTopIdxGeneral / deriveLeftBottomIdxGeneral's (TComDataCU.cpp:1837-1944).
 * Only whole CUs are searched: a 2NxN grid has 2 * (height / (2 pu_h)) PU rows, an Nx2N grid 2 * (width / (2 pu_w)) columns, a
 * 2NxnU / 2NxnD grid 2 * (height / pu_h) rows of width / pu_w PUs, an nLx2N / nRx2N grid height / pu_h rows of 2 * (width / pu_w).
 * out / out_mvp hold the fields one after the other, field f as gw_f x gh_f entries in raster order of its PU grid. */
int x265amd_field_me_multi(x265amd_ctx *ctx, int n_fields, const int *enc_planes, const int *ref_planes, const int *pu_w,
                           const int *pu_h, const int *part_mode, int merange, int method, int subme, int qp,
                           x265amd_me_result *out, int16_t *out_mvp);
/* The same with the temporal AMVP candidate of TComDataCU::fillMvpCand (Lib/TLibCommon/TComDataCU.cpp:2635-2697).
 *   col (NULL: none, x265amd_field_me_multi) = the co-located picture's motion for every field, one after the other: per field
 *   ((width + 15) / 16) x ((height + 15) / 16) entries in raster order, one per 16x16 block (the granularity at which the encoder
 *   keeps a coded picture's motion: xGetColMVP masks the partition address, :3010), already scaled to the field's picture /
 *   reference distance (xGetDistScaleFactor :3079-3096 and scaleMv :45-51, as xGetColMVP applies them :3047-3075); cost < 0 marks a
 *   block without motion (intra, or no MV in the list xGetColMVP would read).  Host memory, read before the call returns.
 * Per PU the entry below-right of it is taken unless that position lies outside the picture or in the next CTU row (64), else -
 * and when that block has no motion - the one at the PU's centre (xDeriveCenterIdx, :3104-3116).  The candidate follows the spatial
 * ones unpruned and the list is cut to two (:2699-2702): it takes part where fewer than two spatial candidates exist (no left or no
 * above neighbour, or both carrying the same MV - the common case in smooth motion). */
int x265amd_field_me_multi_tmvp(x265amd_ctx *ctx, int n_fields, const int *enc_planes, const int *ref_planes, const int *pu_w,
                                const int *pu_h, const int *part_mode, const x265amd_me_result *col, int merange, int method,
                                int subme, int qp, x265amd_me_result *out, int16_t *out_mvp);
const x265amd_me_result *x265amd_field_results_device(const x265amd_ctx *ctx);

/* number of block-cost evaluations executed by the search kernels since the counter was last
 * reset (one sad/satd/sa8d of a w x h block == 1; sad_x4 == 4); blocks until the stream is idle */
int x265amd_eval_count(x265amd_ctx *ctx, int reset, uint64_t *evals);

/* Kernel timing: when enabled, every search batch brackets its kernels with HIP events on the
 * context's stream, the way the reference brackets its two motionEstimate calls with
 * CYCLE_COUNTER_START/STOP(ME) (Lib/TLibEncoder/TEncSearch.cpp:2424-2451, common/common.h:110-125).
 * x265amd_kernel_times returns, for the most recent batch, ms[0] = exhaustive (FULL) kernel,
 * ms[1] = general search kernel (0 if a kernel did not run); it waits for the batch. */
int x265amd_set_timing(x265amd_ctx *ctx, int enable);
int x265amd_kernel_times(x265amd_ctx *ctx, float ms[2]);

/* MV-cost model (encoder/bitcost.h:43-53, encoder/bitcost.cpp:35-85) */
/* out[i + 32768] = BitCost::s_costs[qp][i], i in [-32767, 32767]; out[0] = 0 */
int x265amd_mvcost_table(x265amd_ctx *ctx, int qp, uint16_t *out65536);
/* BitCost::bitcost(mv) for predictor mvp (all q-pel) */
int x265amd_bitcost(x265amd_ctx *ctx, int mvp_x, int mvp_y, int mv_x, int mv_y);

/* ---- per-call primitives ---------------------------------------------------------------
 * Functions with the exact signatures of the reference's primitive table
 * (pixelcmp_t / pixelcmp_x3_t / pixelcmp_x4_t, common/primitives.h:180-184) operating on HOST
 * pointers.  `pixel` is uint8_t in the 8-bit table and uint16_t in the 16-bit table.  Each
 * call stages its blocks to the GPU, runs the cost kernel and returns the result, so that
 * the reference's own TestBench recipe (test/pixelharness.cpp) can be replayed; the batched
 * calls above are the performance path.  They use a process-wide context created by
 * x265amd_setup_primitives(). */
typedef int  (*x265amd_pixelcmp_t)(void *fenc, intptr_t fencstride, void *fref, intptr_t frefstride);
typedef void (*x265amd_pixelcmp_x3_t)(void *fenc, void *fref0, void *fref1, void *fref2, intptr_t frefstride, int *res);
typedef void (*x265amd_pixelcmp_x4_t)(void *fenc, void *fref0, void *fref1, void *fref2, void *fref3, intptr_t frefstride, int *res);

/* The slice of EncoderPrimitives (common/primitives.h:240-250) this library provides;
 * indices are the Partitions / SquareBlocks enums (common/primitives.h:77-98). */
typedef struct x265amd_primitives
{
    x265amd_pixelcmp_t    sad[64];
    x265amd_pixelcmp_x3_t sad_x3[64];
    x265amd_pixelcmp_x4_t sad_x4[64];
    x265amd_pixelcmp_t    satd[64];
    x265amd_pixelcmp_t    sa8d_inter[64];
    x265amd_pixelcmp_t    sa8d[5];
} x265amd_primitives;

/* Fill `p` the way Setup_Vector_Primitives / Setup_Assembly_Primitives overwrite slots of
 * x265::primitives (common/primitives.cpp:129-139).  bytes_per_px selects the pixel type
 * (1: HIGH_BIT_DEPTH=0, 2: HIGH_BIT_DEPTH=1). */
int x265amd_setup_primitives(x265amd_primitives *p, int device, int bytes_per_px);

/* Text identifying the build: "x265_amd <hash>", <hash> = SHA-1 (first 12 hex digits) over the library's source
 * files at build time (x265_amd/build.py), so that a measurement can name the sources of the binary it ran.
 * The reference's counterpart is x265_version_str (common/version.cpp). */
const char *x265amd_build_info(void);

/* PartitionFromSizes (common/primitives.cpp:40-54); -1 if not a partition */
int x265amd_partition_from_sizes(int w, int h);

#ifdef __cplusplus
}
#endif
#endif /* X265_AMD_H */
