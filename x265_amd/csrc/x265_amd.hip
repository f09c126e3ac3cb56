// x265_amd C ABI (include/x265_amd.h): context, planes, MV cost tables and kernel launches.
// Everything computes on the GPU; there is no host fallback.
#include "me_kernels.h"
#include "me_full.h"
#include "me_hier.h"
#include "me_plan.h"
#include "me_had.h"
#include "me_lowres.h"
#include "me_field.h"
#include "me_lane.h"

// (the general search, field and lookahead kernels are instantiated in x265_amd_part.hip)
#define XA_INST extern template
#include "x265_amd_inst.h"
#undef XA_INST

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

using namespace xa;

namespace {

// lambda^2 for non-I slices per QP (reference: Lib/TLibCommon/TComRom.cpp:527-536)
const double k_lambda2_non_I[52] =
{
    0.05231, 0.060686, 0.07646, 0.096333333, 0.151715667, 0.15292, 0.192666667, 0.242745, 0.382299,
    0.385333333, 0.485489333, 0.611678333, 0.963333333, 0.970979, 1.223356667, 1.541333333, 2.427447667,
    2.446714, 3.082666667, 3.883916667, 6.116785, 6.165333333, 7.767833333, 9.786856667, 15.41333333,
    16.57137733, 22.183542, 29.5936, 39.357022, 52.19656867, 69.05173333, 91.14257667, 150.0651347,
    157.8325333, 207.1422197, 271.4221573, 443.904, 447.4271947, 563.722941, 710.2464, 1118.567987,
    1127.445883, 1420.4928, 1789.70878, 2818.614706, 2840.9856, 3579.41756, 4509.78353, 7102.464,
    7158.83512, 9019.56706, 11363.9424
};

const int k_dims[8] = { 4, 8, 12, 16, 24, 32, 48, 64 };

struct DevBuf
{
    void  *p = nullptr;
    size_t cap = 0;
};

} // namespace

constexpr int EVAL_WORDS = (EVAL_SLOTS + 3) * 16;   // device words behind d_evals (the last 32 only hold profiling sums)

// results to device-mapped host memory, one 64-bit word per thread
__global__ void __launch_bounds__(256) k_copy_words(const unsigned long long *src, unsigned long long *dst, int n)
{
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i < n)
        dst[i] = src[i];
}

// A padded plane from device-mapped host memory.  Few workgroups on purpose: the reads cross the bus with microseconds
// of latency, and a launch that floods the memory pipeline with them holds up the search kernels' own reads (a
// bus-wide launch stretched a concurrent k_pack_fenc from 6 to 60 us and a pipelined 1080p frame from 0.52 to 0.65 ms; with
// 4 workgroups it is 0.55 ms, with 24: 0.58); each thread keeps four 16-byte reads in flight, which is enough for the
// copy to finish well inside the search of the picture before.
constexpr int COPY_WGS = 4;
__global__ void __launch_bounds__(256) k_copy_bytes(const unsigned char *src, unsigned char *dst, size_t bytes)
{
    const size_t n16 = bytes / 16, stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride)
    {
        const uint4 a = ((const uint4*)src)[i], b = ((const uint4*)src)[i + stride], c = ((const uint4*)src)[i + 2 * stride],
                    d = ((const uint4*)src)[i + 3 * stride];
        ((uint4*)dst)[i] = a; ((uint4*)dst)[i + stride] = b; ((uint4*)dst)[i + 2 * stride] = c; ((uint4*)dst)[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride)
        ((uint4*)dst)[i] = ((const uint4*)src)[i];
    if (blockIdx.x == 0 && threadIdx.x < bytes - 16 * n16)
        dst[16 * n16 + threadIdx.x] = src[16 * n16 + threadIdx.x];
}

struct x265amd_ctx
{
    int          device = 0;
    hipStream_t  stream = nullptr;
    std::mutex   mu;
    std::string  err = "";

    PlaneDesc    h_planes[MAX_PLANES];
    bool         used[MAX_PLANES];
    void        *owned[MAX_PLANES];       // device allocation owned by the context (upload) or NULL (wrapped)
    void        *base[MAX_PLANES];        // device address of the padded allocation
    PlaneDesc   *d_planes = nullptr;
    PlaneDesc   *h_planes_pin = nullptr;  // pinned copy the table is uploaded from (a plain DMA, no staging)
    bool         planes_dirty = true;

    std::vector<float>    bitsizes;       // BitCost::s_bitsizes
    std::vector<uint16_t> h_cost[52];
    uint16_t    *d_cost = nullptr;        // 52 x 65536
    unsigned long long *d_evals = nullptr;   // EVAL_SLOTS counters 16 words apart (+ 16 words of developer counters)

    DevBuf jobs, results, fullbest, pairs, costs, stage, items, item_jobs, look_pairs, look_results, look_progress, plan_mem;
    DevBuf field_results, field_mvp, field_col;
    void  *h_stage = nullptr;             // pinned staging for the per-call primitives
    size_t h_stage_cap = 0;
    // Largest PU of a FULL job with an exhaustive-kernel scan that the row-per-job kernel takes (X265AMD_LANE_FULL, up to
    // ROW_MAX_AREA).  Default: none.  Measured on MI355X the general kernel is as fast or faster on these leftovers: 1080p
    // frames of one shape, subme 0 / 5: 16x16 0.022 / 0.077 ms against 0.026 / 0.065, 32x32 0.033 / 0.129 against 0.026 /
    // 0.072; the tail of the 4K 10-bit frame with all 593 PUs per CTU 2.8 ms against 2.3.
    int    lane_full_area = LANE_MAX_AREA;
    bool   hier_ok = false;               // the MV-cost tables are small enough for the hierarchical kernel's 32-bit keys
    bool   timing = false;
    int    flags = 0;
    bool   timed_full = false, timed_any = false;
    hipEvent_t ev[3] = { nullptr, nullptr, nullptr };
    // Independent kernels of one batch (job classes, exhaustive-kernel size classes) run side by
    // side on auxiliary streams, so that one launch's tail overlaps the others: fork() makes the
    // auxiliary streams wait for the main one, join() the reverse.
    static constexpr int N_AUX = 3;
    int n_lanes = 2;                       // streams in use: 1 .. N_AUX + 1 (X265AMD_LANES); two give the overlap, more only cost hardware queues
    hipStream_t aux[N_AUX] = { nullptr, nullptr, nullptr };
    hipEvent_t ev_fork = nullptr, ev_join[N_AUX] = { nullptr, nullptr, nullptr };
    // planes refilled asynchronously (x265amd_plane_refill_async): the upload stream, a marker event of the main stream,
    // and per plane the event behind its last refill (a search that names the plane waits for it)
    hipStream_t up_stream = nullptr;
    bool        up_stream_owned = false;
    hipEvent_t  ev_mark = nullptr;
    hipEvent_t  plane_filled[MAX_PLANES] = {};
    // per plane and stream set: behind the last run on that set that named the plane (sets 0 and 4 run on the main stream, which a
    // refill orders itself behind anyway); read_pending bit s: a refill of the plane has to wait for plane_read[.][s]
    hipEvent_t  plane_read[MAX_PLANES][8] = {};
    unsigned char read_pending[MAX_PLANES] = {};
    unsigned char fill_pending[MAX_PLANES] = {};   // bit s: stream set s has not yet waited for the plane's last refill
    // results leave on a stream of their own (x265amd_joblist_results_async), so that a download does not hold up the
    // next batch on the main stream; created on first use
    hipStream_t out_stream = nullptr;
    x265amd_joblist *batch_list = nullptr;   // the host batch call's job list: its device buffers are kept between calls
    hipStream_t lane(int k) const { return (k % n_lanes) == 0 ? stream : aux[(k % n_lanes) - 1]; }
    // Stream sets: set 0 = { stream, aux[0] }, set 1 = { aux[1], aux[2] }, sets 2 and 3 = those pairs with main and side
    // stream exchanged.  The four streams are created back to back, so each has a hardware queue of its own, and batches
    // queued on different sets run side by side (x265amd_joblist_run_overlapped).
    int cur_set = 0;
};

// For the lifetime of the object the context's main and first side stream are those of stream set `set`
struct StreamSet
{
    x265amd_ctx *c;
    int lanes;
    hipStream_t s0, a0, a1, a2;
    hipEvent_t j0, j1, j2;
    StreamSet(x265amd_ctx *ctx, int set) : c(ctx), lanes(ctx->n_lanes), s0(ctx->stream), a0(ctx->aux[0]), a1(ctx->aux[1]), a2(ctx->aux[2]),
                                           j0(ctx->ev_join[0]), j1(ctx->ev_join[1]), j2(ctx->ev_join[2])
    {
        if (!set)
            return;
        // (main, side, and the join event that belongs to the side stream)
        hipStream_t m[4] = { s0, a1, a0, a2 }, sd[4] = { a0, a2, s0, a1 };
        hipEvent_t jv[4] = { j0, j2, j1, j1 };       // (sets 2 and 3 record on what is another set's main stream: the spare event)
        c->stream = m[set & 3];
        c->aux[0] = sd[set & 3];
        c->ev_join[0] = jv[set & 3];
        c->n_lanes = set >= 4 ? 1 : std::min(c->n_lanes, 2);       // sets 4 .. 7: the main stream of set 0 .. 3 alone
        c->cur_set = set;
    }
    ~StreamSet()
    {
        c->stream = s0; c->aux[0] = a0; c->aux[1] = a1; c->aux[2] = a2;
        c->ev_join[0] = j0; c->ev_join[1] = j1; c->ev_join[2] = j2;
        c->n_lanes = lanes;
        c->cur_set = 0;
    }
};

// A planned job list: the jobs live in device memory together with the work-item plan of the
// exhaustive kernel (which jobs share a staged window, how large jobs are cut into row slabs).
struct x265amd_joblist
{
    int     n = 0, bpp = 1;
    int     n_refs = 0;                   // reference planes the jobs were validated against (ref_idx < n_refs)
    std::vector<x265amd_me_job> h_jobs;   // host copy, for the range check against the planes of a run
    std::vector<int64_t> checked_geom;    // geometry of the planes the ranges were last checked against
    DevBuf  jobs, results, fb, items, item_jobs, idx, hist;
    DevBuf  hier_bundles, hier_items, hier_fenc;      // whole-cell bundles of the hierarchical exhaustive kernel
    int     n_hier_bundles = 0, n_hier_items = 0, hier_lds_bytes = 0;
    DevBuf  had_bundles, had_items, had_fenc, had_quads;   // the same for the Hadamard exhaustive kernel (had_quads: HierQuads per bundle, if any)
    int     n_had_bundles = 0, n_had_items = 0, had_lds_bytes = 0, had_max_depth = 10, had_threads = 128;
    bool    had_rects = false;            // some bundle of the candidate-per-lane kernel has rectangular PUs
    bool    had_quad_pus = false;         // some bundle has PUs with 4- or 12-pixel sides
    DevBuf  fb_jobs;                      // with Hadamard bundles: one flag per job, "an exhaustive kernel covers its integer stage"
    bool    has_fb = false;               // some job's integer stage runs in an exhaustive kernel
    std::vector<int> order;               // host copy of idx
    bool    ran = false;
    hipEvent_t ev_ran = nullptr, ev_out = nullptr;   // behind the last run / behind the last asynchronous download of the results
    bool    out_pending = false;          // a download may still read `results`: the next run waits for ev_out
    hipEvent_t ev_done = nullptr;         // behind the list's last run, on the main stream of the set it ran on
    int     last_set = -1;                // stream set of the last run (-1: none yet)
    bool    keep_hist = true;             // record per-job evaluation counts (for x265amd_joblist_rebalance)
    int     cls_count[JOB_CLASSES] = { 0, 0, 0, 0, 0 };   // idx = [class 0 ..., class 1 ..., class 2 ..., quad-per-job class ..., row-per-job class ...]
    struct Launch { int first, count, lds_bytes, w4; };
    std::vector<Launch> launches;          // one k_full_items launch per LDS size class
};

namespace {

int fail(x265amd_ctx *c, int code, const char *what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof(buf), "%s", what);
    if (c)
        c->err = buf;
    return code;
}

#define HIPCHK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return fail((c), X265AMD_ERR_HIP, #call, e_); } while (0)

int ensure(x265amd_ctx *c, DevBuf &b, size_t bytes)
{
    if (b.cap >= bytes)
        return 0;
    if (b.p)
    {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t cap = bytes + bytes / 4 + 4096;
    HIPCHK(c, hipMalloc(&b.p, cap));
    b.cap = cap;
    return 0;
}

// BitCost::CalculateLogs + BitCost::setQP (reference: encoder/bitcost.cpp:35-85).  The float /
// double expression shapes are kept (this file is compiled with -ffp-contract=off).
void build_cost_tables(x265amd_ctx *c)
{
    c->bitsizes.resize(32769);
    c->bitsizes[0] = 0.718f;
    float log2_2 = 2.0f / logf(2.0f);
    for (int i = 1; i <= 32768; i++)
        c->bitsizes[i] = logf((float)(i + 1)) * log2_2 + 1.718f;
    for (int qp = 0; qp < 52; qp++)
    {
        std::vector<uint16_t> &t = c->h_cost[qp];
        t.assign(65536, 0);
        double lambda = sqrt(k_lambda2_non_I[qp]);
        for (int i = 0; i < 32768; i++)
        {
            double v = c->bitsizes[i] * lambda + 0.5f;
            uint16_t e = (uint16_t)(v < 65535 ? v : 65535);
            t[32768 + i] = e;
            t[32768 - i] = e;
        }
    }
}

int sync_planes(x265amd_ctx *c)
{
    if (!c->planes_dirty)
        return 0;
    memcpy(c->h_planes_pin, c->h_planes, sizeof(c->h_planes));
    HIPCHK(c, hipMemcpyAsync(c->d_planes, c->h_planes_pin, sizeof(c->h_planes), hipMemcpyHostToDevice, c->stream));
    // the pinned copy may change right after this call returns
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->planes_dirty = false;
    return 0;
}

int add_plane(x265amd_ctx *c, void *dev_base, void *owned, int width, int height, int margin_x, int margin_y,
              intptr_t stride, int bpp, int depth, int *plane_id)
{
    int id = -1;
    for (int i = 0; i < MAX_PLANES; i++)
        if (!c->used[i])
        {
            id = i;
            break;
        }
    if (id < 0)
        return fail(c, X265AMD_ERR_NOMEM, "too many planes registered");
    PlaneDesc &d = c->h_planes[id];
    int64_t org = (int64_t)margin_y * stride + margin_x;
    d.origin = (const char*)dev_base + org * bpp;
    d.lo = -org;
    d.hi = (int64_t)(height + 2 * margin_y) * stride - org;
    d.stride = (int32_t)stride;
    d.width = width;
    d.height = height;
    d.bpp = bpp;
    d.depth = depth;
    d.margin_x = margin_x;
    d.margin_y = margin_y;
    d.lr_step = 0;
    c->used[id] = true;
    c->owned[id] = owned;
    c->base[id] = dev_base;
    c->planes_dirty = true;
    *plane_id = id;
    return 0;
}

int check_plane_args(x265amd_ctx *c, int width, int height, int margin_x, int margin_y, intptr_t stride, int bpp, int depth)
{
    if (width <= 0 || height <= 0 || margin_x < 0 || margin_y < 0 || stride < width + 2 * margin_x ||
        (bpp != 1 && bpp != 2) || depth < 8 || depth > 8 * bpp || depth > 14)
        return fail(c, X265AMD_ERR_INVALID, "bad plane geometry");
    // the kernels address pixels with 32-bit element offsets relative to a block inside the plane
    if ((int64_t)(height + 2 * margin_y) * (int64_t)stride >= (int64_t)1 << 30)
        return fail(c, X265AMD_ERR_INVALID, "plane too large (2^30 elements or more)");
    return 0;
}

bool valid_dim(int v)
{
    for (int i = 0; i < 8; i++)
        if (k_dims[i] == v)
            return true;
    return false;
}

BatchParams make_params(x265amd_ctx *c, int enc_plane, const int *ref_planes, int n_refs)
{
    BatchParams P;
    memset(&P, 0, sizeof(P));
    P.planes = c->d_planes;
    for (int q = 0; q < 52; q++)
        P.cost[q] = c->d_cost + (size_t)q * 65536 + 32768;
    P.enc_plane = enc_plane;
    P.n_refs = n_refs;
    P.flags = c->flags;
    for (int i = 0; i < n_refs; i++)
        P.ref_planes[i] = ref_planes[i];
    return P;
}

int check_batch_planes(x265amd_ctx *c, int enc_plane, const int *ref_planes, int n_refs)
{
    if (enc_plane < 0 || enc_plane >= MAX_PLANES || !c->used[enc_plane])
        return fail(c, X265AMD_ERR_INVALID, "enc_plane is not a registered plane");
    if (n_refs < 1 || n_refs > X265AMD_MAX_REFS || !ref_planes)
        return fail(c, X265AMD_ERR_INVALID, "bad reference plane list");
    for (int i = 0; i < n_refs; i++)
    {
        int r = ref_planes[i];
        if (r < 0 || r >= MAX_PLANES || !c->used[r])
            return fail(c, X265AMD_ERR_INVALID, "ref plane is not a registered plane");
        if (c->h_planes[r].bpp != c->h_planes[enc_plane].bpp)
            return fail(c, X265AMD_ERR_INVALID, "enc and ref planes differ in pixel size");
    }
    return 0;
}

// searches that name a plane with a refill in flight start behind it
int wait_plane_fills(x265amd_ctx *c, int enc_plane, const int *ref_planes, int n_refs)
{
    for (int k = -1; k < n_refs; k++)
    {
        const int id = k < 0 ? enc_plane : ref_planes[k];
        if (c->fill_pending[id] & (1 << c->cur_set))
        {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->plane_filled[id], 0));
            c->fill_pending[id] &= ~(1 << c->cur_set);
        }
    }
    return 0;
}

constexpr int FULL_LDS_MAX = 64 * 1024;

int record_start(x265amd_ctx *c)
{
    if (c->timing)
        HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->timed_full = false;
    return 0;
}

int fork_streams(x265amd_ctx *c, int lanes)
{
    HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
    for (int i = 0; i + 1 < c->n_lanes && i + 1 < lanes; i++)
        HIPCHK(c, hipStreamWaitEvent(c->aux[i], c->ev_fork, 0));
    return 0;
}
int join_streams(x265amd_ctx *c, int lanes)
{
    for (int i = 0; i + 1 < c->n_lanes && i + 1 < lanes; i++)
    {
        HIPCHK(c, hipEventRecord(c->ev_join[i], c->aux[i]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[i], 0));
    }
    return 0;
}

// idx[cls] / cnt[cls]: device index lists of the job classes (idx == NULL: every block looks at its own
// job and the kernels of the other classes skip it)
// (LR: some reference of the batch is a half-resolution set)
template <typename PIX, bool LR>
void launch_classes(x265amd_ctx *c, const BatchParams &P, const x265amd_me_job *d_jobs, const int *const *idx, const int *cnt,
                    const FullBest *fb, int fb_stride, x265amd_me_result *d_out, bool side_by_side)
{
    // The large PUs first: they are the long-running workgroups.  A large PU's job is a dependent chain whose worker
    // wavefronts mostly wait; beside launches of small PUs (throughput-bound: they want the registers and wavefront slots)
    // it runs on four wavefronts, alone on the GPU on eight (10 % shorter chain).
    int k = 0;
    const bool shared = side_by_side && cnt[0] + cnt[1] >= cnt[2];
    if (cnt[2] > 0 && shared)
        hipLaunchKernelGGL((k_search<JobLds<PIX, 4, 1024, LR, !LR>>), dim3(cnt[2]), dim3(256), 0, c->lane(side_by_side ? k++ : 0), P, d_jobs, idx[2], cnt[2], 2, fb, fb_stride, d_out, c->d_evals);
    else if (cnt[2] > 0)
        hipLaunchKernelGGL((k_search<JobLds<PIX, 8, 1024, LR, !LR>>), dim3(cnt[2]), dim3(512), 0, c->lane(side_by_side ? k++ : 0), P, d_jobs, idx[2], cnt[2], 2, fb, fb_stride, d_out, c->d_evals);
    if (cnt[1] > 0)
        hipLaunchKernelGGL((k_search<JobLds<PIX, 1, 256, LR, !LR>>), dim3(cnt[1]), dim3(64), 0, c->lane(side_by_side ? k++ : 0), P, d_jobs, idx[1], cnt[1], 1, fb, fb_stride, d_out, c->d_evals);
    if (cnt[0] > 0)
        hipLaunchKernelGGL((k_search<JobLds<PIX, 1, 64, LR, !LR>>), dim3(cnt[0]), dim3(64), 0, c->lane(side_by_side ? k++ : 0), P, d_jobs, idx[0], cnt[0], 0, fb, fb_stride, d_out, c->d_evals);
    // the smallest PUs (one job per quad of lanes) and the leftovers of exhaustive scans (one job per 16-lane row) against
    // full-resolution references; else the general kernel, every launch taking the jobs of its own class
    for (int gc = LANE_CLASS; gc <= ROW_CLASS; gc++)
    {
        const int m = cnt[gc];
        if (m <= 0)
            continue;
        hipStream_t st = c->lane(side_by_side ? k++ : 0);
        if (P.lane_on && gc == LANE_CLASS)
            hipLaunchKernelGGL((k_lane<PIX, 4>), dim3((m + 15) / 16), dim3(64), 0, st, P, d_jobs, idx[gc], m, fb, fb_stride, d_out, c->d_evals);
        else if (P.lane_on)
            hipLaunchKernelGGL((k_lane<PIX, 16>), dim3((m + 3) / 4), dim3(64), 0, st, P, d_jobs, idx[gc], m, fb, fb_stride, d_out, c->d_evals);
        else if (idx[gc])
        {
            // (a planned list run against half-resolution references)
            hipLaunchKernelGGL((k_search<JobLds<PIX, 4, 1024, LR, !LR>>), dim3(m), dim3(256), 0, st, P, d_jobs, idx[gc], m, 2, fb, fb_stride, d_out, c->d_evals);
            hipLaunchKernelGGL((k_search<JobLds<PIX, 1, 256, LR, !LR>>), dim3(m), dim3(64), 0, st, P, d_jobs, idx[gc], m, 1, fb, fb_stride, d_out, c->d_evals);
            hipLaunchKernelGGL((k_search<JobLds<PIX, 1, 64, LR, !LR>>), dim3(m), dim3(64), 0, st, P, d_jobs, idx[gc], m, 0, fb, fb_stride, d_out, c->d_evals);
        }
    }
}

int launch_general(x265amd_ctx *c, const BatchParams &P0, int bpp, const x265amd_me_job *d_jobs, int n,
                   const int *idx_base, const int *cls_count,
                   const FullBest *fb, int fb_stride, x265amd_me_result *d_out, bool timing_events = true)
{
    if (c->timing && timing_events)
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    const int *idx[JOB_CLASSES] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    int cnt[JOB_CLASSES] = { n, n, n, n, n };
    BatchParams P = P0;
    if (idx_base)
    {
        int off = 0;
        for (int k = 0; k < JOB_CLASSES; k++)
        {
            idx[k] = idx_base + off;
            cnt[k] = cls_count[k];
            off += cls_count[k];
        }
    }
    // with index lists every launch has work: run the classes side by side
    // (without them every launch visits all jobs and most workgroups leave at once: side by side as well, so that the
    // few long ones of the large-PU class do not hold up the rest)
    const int lanes = idx_base ? std::min(c->n_lanes, (cnt[0] > 0) + (cnt[1] > 0) + (cnt[2] > 0) + (cnt[3] > 0) + (cnt[4] > 0))
                               : (n >= 4096 ? c->n_lanes : 1);
    if (lanes > 1)
        if (int rc = fork_streams(c, lanes)) return rc;
    bool lr = false;
    for (int i = 0; i < P.n_refs; i++)
        lr = lr || c->h_planes[P.ref_planes[i]].lr_step != 0;
    P.lane_on = (!lr && !(c->flags & 128)) ? c->lane_full_area : 0;       // (tuning aid: X265AMD_FLAGS bit 7 keeps every PU in the general kernel)
    if (bpp == 1 && !lr)
        launch_classes<uint8_t, false>(c, P, d_jobs, idx, cnt, fb, fb_stride, d_out, lanes > 1);
    else if (bpp == 1)
        launch_classes<uint8_t, true>(c, P, d_jobs, idx, cnt, fb, fb_stride, d_out, lanes > 1);
    else if (!lr)
        launch_classes<uint16_t, false>(c, P, d_jobs, idx, cnt, fb, fb_stride, d_out, lanes > 1);
    else
        launch_classes<uint16_t, true>(c, P, d_jobs, idx, cnt, fb, fb_stride, d_out, lanes > 1);
    if (lanes > 1)
        if (int rc = join_streams(c, lanes)) return rc;
    HIPCHK(c, hipGetLastError());
    if (c->timing && timing_events)
    {
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
        c->timed_any = true;
    }
    return 0;
}

// Unplanned path (job array only in device memory): one exhaustive work item per job, built on the device.
// Scratch of n jobs: one FullBest, one FullItem and one item-job index each.  Every slot a kernel of the
// call reads is written earlier in the same call (k_make_items fills all items and indices; the exhaustive
// kernel writes the FullBest of every job it is eligible for, and only those are read), so the buffers
// need no clearing and may hold an earlier call's contents.
// plan_mem (optional): memory for the device-side planner of me_plan.h (plan_bytes(cells, n) bytes, 256-byte aligned) for
// the picture's cells_x x cells_y 64x64 cells per reference plane.  With it the square FULL-search SAD jobs of a cell on
// 8-bit planes share the hierarchical kernel's pass, as in a planned list; what the planner leaves out is scanned job by job.
size_t plan_bytes(int n_cells, int n)
{
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    return up(4 * (size_t)n_cells * HIER_PUS) + up(sizeof(HierBundle) * (size_t)n_cells) +
           up(4 * ((size_t)n_cells + 1)) + up((size_t)HIER_FENC_DW * 4 * n_cells) + up((size_t)n);
}

int launch_search_unplanned(x265amd_ctx *c, const BatchParams &P, int bpp, const x265amd_me_job *d_jobs, int n,
                            x265amd_me_result *d_out, FullBest *fb, FullItem *items, int *item_jobs,
                            unsigned char *plan_mem = nullptr, int cells_x = 0, int cells_y = 0)
{
    if (int rc = record_start(c)) return rc;
    const unsigned char *skip = nullptr;
    if (plan_mem && bpp == 1 && c->hier_ok && !(c->flags & 16))
    {
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        DevPlan D;
        D.cells_x = cells_x; D.cells_y = cells_y; D.n_cells = cells_x * cells_y * P.n_refs;
        unsigned char *p = plan_mem;
        D.slots = (uint32_t*)p;        p += up(4 * (size_t)D.n_cells * HIER_PUS);
        D.bundles = (HierBundle*)p;    p += up(sizeof(HierBundle) * (size_t)D.n_cells);
        D.offsets = (int32_t*)p;       p += up(4 * ((size_t)D.n_cells + 1));
        uint32_t *fenc = (uint32_t*)p; p += up((size_t)HIER_FENC_DW * 4 * D.n_cells);
        D.in_bundle = p;
        D.keys = (unsigned long long*)fb;
        static_assert(sizeof(FullBest) == sizeof(unsigned long long), "a job's key lives in its FullBest");
        const int jb = (n + 255) / 256;
        hipLaunchKernelGGL(k_plan_clear, dim3(std::min(1024, std::max(jb, (D.n_cells * HIER_PUS + 255) / 256))), dim3(256), 0, c->stream, D, n);
        hipLaunchKernelGGL(k_plan_slots, dim3(jb), dim3(256), 0, c->stream, P, D, d_jobs, n);
        hipLaunchKernelGGL(k_plan_bundles, dim3(D.n_cells), dim3(64), 0, c->stream, P, D, d_jobs, plan_lds_class(PLAN_LDS_CLASSES - 1));
        hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(256), 0, c->stream, D);
        hipLaunchKernelGGL(k_pack_fenc, dim3(D.n_cells), dim3(256), 0, c->stream, P, (const HierBundle*)D.bundles, D.n_cells, fenc);
        // (one launch per LDS size class over all work items; a workgroup whose bundle belongs to another class leaves at once)
        for (int k = 0; k < PLAN_LDS_CLASSES; k++)
            hipLaunchKernelGGL(k_full_hier<true>, dim3(D.n_cells * PLAN_SLABS), dim3(HIER_THREADS), plan_lds_class(k), c->stream, P, d_jobs,
                               (const HierBundle*)D.bundles, (const HierItem*)D.offsets, D.n_cells, (const uint32_t*)fenc, fb, -k);
        HIPCHK(c, hipGetLastError());
        skip = D.in_bundle;
    }
    hipLaunchKernelGGL(k_make_items, dim3((n + 255) / 256), dim3(256), 0, c->stream, P, d_jobs, n, item_jobs, items, skip,
                       (unsigned long long*)fb);
    // (two workgroups with the full LDS allocation fit a CU; four rounds of them even out items of different length)
    const int full_grid = std::min(n, 2048);
    if (bpp == 1)
        hipLaunchKernelGGL((k_full_items<0, uint8_t>), dim3(full_grid), dim3(FULL_THREADS), FULL_LDS_MAX, c->stream, P, d_jobs,
                           (const int*)item_jobs, (const FullItem*)items, n, fb, 1, FULL_LDS_MAX / 4);
    else
        hipLaunchKernelGGL((k_full_items<0, uint16_t>), dim3(full_grid), dim3(FULL_THREADS), FULL_LDS_MAX, c->stream, P, d_jobs,
                           (const int*)item_jobs, (const FullItem*)items, n, fb, 1, FULL_LDS_MAX / 4);
    HIPCHK(c, hipGetLastError());
    c->timed_full = true;
    return launch_general(c, P, bpp, d_jobs, n, nullptr, nullptr, fb, 1, d_out);
}

// ---- exhaustive-kernel plan ------------------------------------------------------------------
// LDS size classes: a class admits floor(160 KiB / bytes) workgroups per CU
const int k_lds_classes[] = { 160 * 1024 / 6, 160 * 1024 / 5, 160 * 1024 / 4, 160 * 1024 / 3, FULL_LDS_MAX };
constexpr int PLAN_CELL = 32;                  // small PUs whose origins share a 32x32 cell share a window
constexpr double PLAN_SLAB_WORK = 16.0e6;      // target pixel-SADs per work item of a large PU

int pitch_for(int nd, int xge)
{
    return nd + ((((xge >> 1) - nd) % 16) + 16) % 16;       // 2*pitch == xge (mod 32), pitch >= nd
}

struct PlannedItem { FullItem it; std::vector<int> members; int lds_bytes; };

// window geometry + LDS need of an item holding `members` (indices into jobs) for rows [y_lo,y_hi)
bool plan_item(const x265amd_me_job *jobs, int bpp, const std::vector<int> &members, int y_lo, int y_hi, int slot, PlannedItem &out)
{
    const x265amd_me_job &j0 = jobs[members[0]];
    const int W4 = j0.w >> 2, h = j0.h;
    int x0 = INT32_MAX, y0 = INT32_MAX, y1 = INT32_MIN;
    for (int m : members)
    {
        const x265amd_me_job &j = jobs[m];
        x0 = std::min(x0, j.pu_x + j.mvmin_x);
        y0 = std::min(y0, j.pu_y + j.mvmin_y + y_lo);
        int ny = j.mvmax_y - j.mvmin_y + 1;
        int yh = std::min(y_hi, ny);
        // window rows: candidate rows + h - 1, rounded up to the rows a lane task spans (2; 4 for tiny PUs)
        y1 = std::max(y1, j.pu_y + j.mvmin_y + yh + h + (bpp == 1 && h * W4 <= 16 ? 3 : 1));
    }
    int nd = 0;
    for (int m : members)
    {
        const x265amd_me_job &j = jobs[m];
        int nx = j.mvmax_x - j.mvmin_x + 1, xg = (nx + 3) >> 2, xge = xg + (xg & 1);
        nd = std::max(nd, (((j.pu_x + j.mvmin_x - x0) >> 2) + xge + W4) * bpp);      // dwords: 4 / bpp pixels each
    }
    const int nx0 = j0.mvmax_x - j0.mvmin_x + 1, xg0 = (nx0 + 3) >> 2, xge0 = xg0 + (xg0 & 1);
    FullItem &it = out.it;
    it.first = 0;
    it.count = (int)members.size();
    it.y_lo = y_lo;
    it.y_hi = y_hi;
    it.slot = slot;
    it.win_x0 = x0;
    it.win_y0 = y0;
    it.win_rows = y1 - y0;
    it.win_nd = nd;
    it.pitch = bpp == 1 ? pitch_for(nd, xge0) : nd + 2;
    const int scratch_dw = h * W4 * bpp + ((2 * FULL_MAX_CAND + 1) >> 1) + 1;
    const int n_scratch = it.count >= FULL_WAVES ? FULL_WAVES : 1;
    out.lds_bytes = 4 * (n_scratch * scratch_dw + it.win_rows * it.pitch);
    out.members = members;
    return out.lds_bytes <= FULL_LDS_MAX;
}

void plan_group(const x265amd_me_job *jobs, int bpp, std::vector<int> members, std::vector<PlannedItem> &out)
{
    // a shared-window item must cover all candidate rows of its jobs in one staging pass
    int nymax = 0;
    for (int m : members)
        nymax = std::max(nymax, jobs[m].mvmax_y - jobs[m].mvmin_y + 1);
    PlannedItem pi;
    if (members.size() > 1 && plan_item(jobs, bpp, members, 0, nymax, 0, pi) && pi.lds_bytes <= k_lds_classes[2])
    {
        out.push_back(pi);
        return;
    }
    if (members.size() > 1)
    {
        size_t half = members.size() / 2;
        plan_group(jobs, bpp, std::vector<int>(members.begin(), members.begin() + half), out);
        plan_group(jobs, bpp, std::vector<int>(members.begin() + half, members.end()), out);
        return;
    }
    // single job: cut into row slabs.  A slab of p candidate-row pairs is p * XGe lane tasks, executed
    // in ceil(p * XGe / 256) passes of the workgroup.  Slabs are sized to fill k whole passes
    // (p = floor(256 k / XGe)), the remainder goes to the last slab; k is the smallest that keeps
    // a reasonable amount of work per item and at most FB_STRIDE_PLAN slabs.  Short items of
    // whole passes keep both the pass slots and the last round of workgroups on the GPU full.
    const x265amd_me_job &j = jobs[members[0]];
    const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    const int xg = (nx + 3) >> 2, xge = xg + (xg & 1);
    const int pairs_total = (ny + 1) >> 1;
    int p_first = pairs_total;
    for (int k = 1; k <= 64; k++)
    {
        const int p = std::max(1, FULL_THREADS * k / xge);
        if (p >= pairs_total)
            break;                                   // one slab
        const double slab_work = 2.0 * p * nx * j.w * j.h;
        if (slab_work >= PLAN_SLAB_WORK / 4 && (pairs_total + p - 1) / p <= FB_STRIDE_PLAN)
        {
            p_first = p;
            break;
        }
    }
    auto try_plan = [&](int p) {
        std::vector<PlannedItem> tmp;
        int y = 0;
        for (int k = 0; y < ny; k++)
        {
            int y1 = std::min(ny, y + 2 * p);
            PlannedItem p2;
            if (!plan_item(jobs, bpp, members, y, y1, k, p2))
                return false;
            tmp.push_back(p2);
            y = y1;
        }
        out.insert(out.end(), tmp.begin(), tmp.end());
        return true;
    };
    // (if a slab's window does not fit the LDS, thinner slabs need fewer rows)
    for (int p = p_first; p >= 1 && (pairs_total + p - 1) / p <= FB_STRIDE_PLAN; p--)
        if (try_plan(p))
            return;
    // does not fit even as FB_STRIDE_PLAN slabs: one item, tiled inside the kernel (win_rows = 0)
    PlannedItem p3;
    plan_item(jobs, bpp, members, 0, ny, 0, p3);
    p3.it.win_rows = 0;
    p3.it.win_y0 = j.pu_y + j.mvmin_y;
    p3.lds_bytes = FULL_LDS_MAX;
    out.push_back(p3);
}

// Workgroup i of a launch runs on XCD i % 8, and every XCD has its own L2.  Reorder a list that is
// sorted by picture position so that each XCD works on contiguous bands of it: the list is cut
// into 8 * XCD_BANDS bands and XCD x takes bands x, x + 8, ... in turn (several bands per XCD even
// out the lighter work at the picture's top and bottom edges, where search ranges are clipped).
constexpr int N_XCD = 8, XCD_BANDS = 2;
bool g_no_xcd_bands = false;       // tuning aid: X265AMD_FLAGS bit 2 keeps picture order
bool g_hier_one_pred = false;      // tuning aid: X265AMD_FLAGS bit 8 keeps jobs with different predictors in separate bundles
template <class T>
void xcd_interleave(std::vector<T> &v)
{
    const size_t n = v.size(), per = (n + N_XCD * XCD_BANDS - 1) / (N_XCD * XCD_BANDS);
    if (n < 4 * N_XCD * XCD_BANDS || g_no_xcd_bands)
        return;
    std::vector<T> out;
    out.reserve(n);
    for (size_t g = 0; g < (size_t)XCD_BANDS; g++)
        for (size_t k = 0; k < per; k++)
            for (size_t x = 0; x < (size_t)N_XCD; x++)
            {
                const size_t src = (g * N_XCD + x) * per + k;
                if (src < n)
                    out.push_back(v[src]);
            }
    v.swap(out);
}

// ---- plan of the hierarchical exhaustive kernel (me_hier.h) -----------------------------------
// Bundles: FULL-search SAD jobs on 8-bit planes whose PUs are squares of the quadtree of one 64x64
// picture cell and that share reference, QP and predictor.  A bundle is used when measuring its 8x8
// cells once is clearly less work than scanning every job on its own; its candidate rows are cut into
// slabs of about one lane task (4 candidates x HIER_R rows) per thread.
//
// had = true plans the same bundles for the Hadamard kernel (me_had.h): FULL-search SA8D jobs on 8- or 16-bit
// planes up to 10 bits deep.  There a lane task is ONE candidate, a bundle may hold a single job (the
// general kernel's scan is far slower), and a slab has at most 2^HAD_RIDX_BITS candidates.
void plan_hier(const x265amd_me_job *jobs, int n, int bpp, bool had, std::vector<char> &in_bundle,
               std::vector<HierBundle> &bundles, std::vector<HierItem> &items, int &lds_bytes, int *had_threads_out = nullptr,
               std::vector<HierQuads> *quads_out = nullptr)
{
    lds_bytes = 0;
    if (!had && bpp != 1)
        return;
    // (candidate-per-lane kernel: windows of more than 48 x 48 candidates are shared by 4 wavefronts instead of 2, or
    // too few wavefronts fit a CU beside each other's windows; one workgroup size per launch)
    // Workgroup size of the launch: the wavefronts that share a staged window decide how many fit a CU (measured: 4K
    // +-16 SA8D 2.51 / 2.29 / 2.61 ms with 128 / 256 / 512 threads; 1080p 10-bit +-64 SAD 3.06 / 1.64 / 1.23 ms).
    int had_threads = 256;
    if (had)
    {
        for (int i = 0; i < n; i++)
            if (had_kernel_eligible(jobs[i], bpp) && !in_bundle[i] &&
                (jobs[i].mvmax_x - jobs[i].mvmin_x >= 48 || jobs[i].mvmax_y - jobs[i].mvmin_y >= 48))
                had_threads = HAD_THREADS_BIG;
        if (const char *ev = getenv("X265AMD_HAD_THREADS"))     // tuning aid: 128, 256 or 512
            had_threads = atoi(ev) >= 512 ? 512 : atoi(ev) >= 256 ? 256 : 128;
    }
    if (had_threads_out)
        *had_threads_out = had_threads;
    auto fl = [](int v) { return v >= 0 ? v / 64 : -((-v + 63) / 64); };
    auto zidx = [](int cx, int cy) {
        return (cx & 1) | ((cy & 1) << 1) | ((cx & 2) << 1) | ((cy & 2) << 2) | ((cx & 4) << 2) | ((cy & 4) << 3); };
    // a square of the quadtree of cell (gx, gy)?
    auto is_square = [](const x265amd_me_job &j, int gx, int gy) {
        return j.w == j.h && (j.w == 8 || j.w == 16 || j.w == 32 || j.w == 64) &&
               (j.pu_x - gx * 64) % j.w == 0 && (j.pu_y - gy * 64) % j.h == 0; };
    // Another partition of the cell (gx, gy) the candidate-per-lane kernel derives from its running sums (had_rect_value in
    // me_had.h)?  Returns the group it completes in (0..15: 16x16 block in Z-order, 16..19: 32x32 block, 20: whole cell) and
    // its kind, or -1.
    auto rect_kind = [&](const x265amd_me_job &j, int gx, int gy, int &kind) {
        const int ox = j.pu_x - gx * 64, oy = j.pu_y - gy * 64, w = j.w, h = j.h;
        if (((w | h | ox | oy) & 7) || ox + w > 64 || oy + h > 64)
            return -1;
        // halves of a 16x16 block
        if (w == 16 && h == 8 && !(ox & 15)) { kind = (oy & 15) ? 9 : 8;  return zidx(ox >> 3, (oy & ~15) >> 3) >> 2; }
        if (w == 8 && h == 16 && !(oy & 15)) { kind = (ox & 15) ? 11 : 10; return zidx((ox & ~15) >> 3, oy >> 3) >> 2; }
        // strips and halves of a 32x32 block / of the cell: `unit` = 8 or 16 pixels per strip
        for (int unit = 8; unit <= 16; unit += 8)
        {
            const int full = 4 * unit, rx = ox & (full - 1), ry = oy & (full - 1);
            const int grp = unit == 8 ? 16 + (zidx(ox >> 3, oy >> 3) >> 4) : 20;
            const int half_base = unit == 8 ? 8 : 12;       // (32x16 / 16x32 are pairs of 16x16 values; 64x32 / 32x64 pairs of strips)
            if (w == full && rx == 0)
            {
                if (h == unit && ry == 0)            { kind = 0; return grp; }
                if (h == 3 * unit && ry == unit)     { kind = 1; return grp; }
                if (h == 3 * unit && ry == 0)        { kind = 2; return grp; }
                if (h == unit && ry == 3 * unit)     { kind = 3; return grp; }
                if (h == 2 * unit && ry == 0)        { kind = half_base; return grp; }
                if (h == 2 * unit && ry == 2 * unit) { kind = half_base + 1; return grp; }
            }
            if (h == full && ry == 0)
            {
                if (w == unit && rx == 0)            { kind = 4; return grp; }
                if (w == 3 * unit && rx == unit)     { kind = 5; return grp; }
                if (w == 3 * unit && rx == 0)        { kind = 6; return grp; }
                if (w == unit && rx == 3 * unit)     { kind = 7; return grp; }
                if (w == 2 * unit && rx == 0)        { kind = half_base + 2; return grp; }
                if (w == 2 * unit && rx == 2 * unit) { kind = half_base + 3; return grp; }
            }
        }
        return -1; };
    // A partition of the cell (gx, gy) with a 4- or 12-pixel side (HierQuads in me_hier.h)?  Returns its slot: 4 z + k
    // for the halves of cell z, 256 + 8 q + k for the asymmetric partitions of 16x16 block q; or -1.
    auto quad_slot = [&](const x265amd_me_job &j, int gx, int gy) {
        const int ox = j.pu_x - gx * 64, oy = j.pu_y - gy * 64, w = j.w, h = j.h;
        if (ox < 0 || oy < 0 || ox + w > 64 || oy + h > 64)
            return -1;
        if (w == 8 && h == 4 && !(ox & 7) && !(oy & 3)) return 4 * zidx(ox >> 3, oy >> 3) + ((oy & 4) ? 1 : 0);
        if (w == 4 && h == 8 && !(oy & 7) && !(ox & 3)) return 4 * zidx(ox >> 3, oy >> 3) + ((ox & 4) ? 3 : 2);
        const int q = zidx(ox >> 3, oy >> 3) >> 2, rx = ox & 15, ry = oy & 15;
        if (w == 16 && rx == 0)
        {
            if (h == 4 && ry == 0)   return 256 + 8 * q + 0;
            if (h == 12 && ry == 4)  return 256 + 8 * q + 1;
            if (h == 12 && ry == 0)  return 256 + 8 * q + 2;
            if (h == 4 && ry == 12)  return 256 + 8 * q + 3;
        }
        if (h == 16 && ry == 0)
        {
            if (w == 4 && rx == 0)   return 256 + 8 * q + 4;
            if (w == 12 && rx == 4)  return 256 + 8 * q + 5;
            if (w == 12 && rx == 0)  return 256 + 8 * q + 6;
            if (w == 4 && rx == 12)  return 256 + 8 * q + 7;
        }
        return -1; };
    std::map<std::tuple<int, int, int, int, int, int>, std::vector<int>> groups;
    for (int i = 0; i < n; i++)
    {
        const x265amd_me_job &j = jobs[i];
        if (!(had ? had_kernel_eligible(j, bpp) : full_kernel_eligible(j, bpp)) || in_bundle[i])
            continue;
        const int gx = fl(j.pu_x), gy = fl(j.pu_y);
        int kind_unused;
        if (!is_square(j, gx, gy) && !(had && rect_kind(j, gx, gy, kind_unused) >= 0) && !(had && quads_out && quad_slot(j, gx, gy) >= 0))
            continue;
        // (jobs with different predictors share a bundle: they are clustered by window below)
        groups[std::make_tuple((int)j.ref_idx, (int)j.qp | (j.method << 8) | (j.icost << 16), gy, gx, g_hier_one_pred ? (int)j.qmvp_x : 0,
                               g_hier_one_pred ? (int)j.qmvp_y : 0)].push_back(i);
    }
    for (auto &g : groups)
    {
      // Clusters of the cell's jobs: jobs with one predictor stay together; groups with different predictors join while the
      // union of their windows stays within an eighth (at least 4 candidates) of the larger one per axis, so that the
      // shared pass does not scan much more than each job's own range.
      std::vector<std::vector<int>> clusters;
      if (g_hier_one_pred)
          clusters.push_back(g.second);
      else
      {
          struct Sub { std::vector<int> jobs; int x0, y0, x1, y1; long area; };
          std::map<std::pair<int, int>, Sub> subs;
          for (int i : g.second)
          {
              const x265amd_me_job &j = jobs[i];
              auto itb = subs.find({ j.qmvp_x, j.qmvp_y });
              if (itb == subs.end())
                  itb = subs.insert({ { j.qmvp_x, j.qmvp_y }, Sub{ {}, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, 0 } }).first;
              Sub &sb = itb->second;
              sb.jobs.push_back(i);
              sb.x0 = std::min(sb.x0, (int)j.mvmin_x); sb.y0 = std::min(sb.y0, (int)j.mvmin_y);
              sb.x1 = std::max(sb.x1, (int)j.mvmax_x); sb.y1 = std::max(sb.y1, (int)j.mvmax_y);
              sb.area += (long)j.w * j.h;
          }
          std::vector<Sub*> order;
          for (auto &kv : subs)
              order.push_back(&kv.second);
          std::stable_sort(order.begin(), order.end(), [](const Sub *a, const Sub *b) { return a->area > b->area; });
          std::vector<Sub> cl;
          for (Sub *sb : order)
          {
              int best = -1;
              long best_grow = 0;
              for (size_t k = 0; k < cl.size(); k++)
              {
                  const Sub &c2 = cl[k];
                  const int ux0 = std::min(c2.x0, sb->x0), uy0 = std::min(c2.y0, sb->y0), ux1 = std::max(c2.x1, sb->x1), uy1 = std::max(c2.y1, sb->y1);
                  const int nxa = std::max(c2.x1 - c2.x0, sb->x1 - sb->x0) + 1, nya = std::max(c2.y1 - c2.y0, sb->y1 - sb->y0) + 1;
                  if (ux1 - ux0 + 1 > nxa + std::max(4, nxa / 8) || uy1 - uy0 + 1 > nya + std::max(4, nya / 8))
                      continue;
                  const long grow = (long)(ux1 - ux0 + 1) * (uy1 - uy0 + 1) - (long)(c2.x1 - c2.x0 + 1) * (c2.y1 - c2.y0 + 1);
                  if (best < 0 || grow < best_grow)
                  {
                      best = (int)k;
                      best_grow = grow;
                  }
              }
              if (best < 0)
                  cl.push_back(*sb);
              else
              {
                  Sub &c2 = cl[best];
                  c2.jobs.insert(c2.jobs.end(), sb->jobs.begin(), sb->jobs.end());
                  c2.x0 = std::min(c2.x0, sb->x0); c2.y0 = std::min(c2.y0, sb->y0);
                  c2.x1 = std::max(c2.x1, sb->x1); c2.y1 = std::max(c2.y1, sb->y1);
              }
          }
          for (Sub &c2 : cl)
          {
              std::sort(c2.jobs.begin(), c2.jobs.end());
              clusters.push_back(c2.jobs);
          }
      }
      for (const std::vector<int> &members : clusters)
      {
        HierBundle B;
        memset(&B, 0, sizeof(B));
        for (int k = 0; k < HIER_PUS; k++)
            B.job[k] = -1;
        const int gy = std::get<2>(g.first), gx = std::get<3>(g.first);
        int x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
        double old_work = 0;
        std::vector<int> taken;
        std::vector<HierBundle::Rect> rects[21];
        HierQuads QJ;
        memset(&QJ, 0xff, sizeof(QJ));       // every job -1
        for (int i : members)
        {
            const x265amd_me_job &j = jobs[i];
            const int cx = (j.pu_x - gx * 64) >> 3, cy = (j.pu_y - gy * 64) >> 3, z = zidx(cx, cy);
            const int qs = had && quads_out ? quad_slot(j, gx, gy) : -1;
            if (qs >= 0)
            {
                int32_t &slot = qs < 256 ? QJ.cell[qs >> 2][qs & 3] : QJ.blk[(qs - 256) >> 3][(qs - 256) & 7];
                if (slot >= 0)
                    continue;           // the same PU twice: the second one is scanned on its own
                slot = i;
                B.quads = 1;
                if (qs < 256)
                    B.cells |= 1ull << z;
                else                    // the strips of a 16x16 block need its four cells
                    for (int k = 0; k < 4; k++)
                        B.cells |= 1ull << (4 * (z >> 2) + k);
            }
            else if (is_square(j, gx, gy))
            {
                const int k = j.w == 8 ? z : j.w == 16 ? 64 + (z >> 2) : j.w == 32 ? 80 + (z >> 4) : 84;
                if (B.job[k] >= 0)
                    continue;           // the same PU twice: the second one is scanned on its own
                B.job[k] = i;
            }
            else
            {
                size_t total = 0;
                for (auto &r : rects)
                    total += r.size();
                if (total >= (size_t)HIER_MAX_RECT)
                    continue;
                int kind = 0;
                const int grp = rect_kind(j, gx, gy, kind);
                HierBundle::Rect r = { i, (uint32_t)kind };
                // (32-bit keys hold costs below 2^21: bound the PU's cost by its 8x8 cells' worst case plus the largest MV cost)
                const double cell_max = j.icost == X265AMD_SA8D ? 131072.0 : 64.0 * (bpp == 1 ? 255.0 : 4095.0);
                if ((j.w >> 3) * (j.h >> 3) * cell_max + HIER_MVCOST_MAX >= (double)(1 << 21))
                    r.kind |= 0x80000000u;
                rects[grp].push_back(r);
                // the running sums need every cell of the 16x16 block / 32x32 block / cell the PU belongs to
                const int span = grp < 16 ? 2 : grp < 20 ? 4 : 8, bx = cx & ~(span - 1), by = cy & ~(span - 1);
                for (int yy = 0; yy < span; yy++)
                    for (int xx = 0; xx < span; xx++)
                        B.cells |= 1ull << zidx(bx + xx, by + yy);
            }
            taken.push_back(i);
            for (int yy = 0; yy < (j.h >> 3); yy++)
                for (int xx = 0; xx < (j.w >> 3); xx++)
                    B.cells |= 1ull << zidx(cx + xx, cy + yy);
            x0 = std::min(x0, (int)j.mvmin_x); y0 = std::min(y0, (int)j.mvmin_y);
            x1 = std::max(x1, (int)j.mvmax_x); y1 = std::max(y1, (int)j.mvmax_y);
            // (16-bit SAD: k_full_items reaches 19 / 36 / 52 / 43 % of the v_sad_u16 rate at 8 / 16 / 32 / 64 wide PUs, the
            // candidate-per-lane kernel about 36 % on the cells it measures; 8-bit: 39 / 50 / 57 / 57 % against about 30 %)
            const double eff = !had ? 1.0 : bpp == 2 ? (j.w <= 8 ? 0.19 / 0.36 : j.w <= 16 ? 0.35 / 0.36 : j.w <= 32 ? 0.52 / 0.36 : 0.43 / 0.36)
                                                     : (j.w <= 8 ? 0.39 / 0.30 : j.w <= 16 ? 0.50 / 0.30 : 0.57 / 0.30);
            old_work += (double)j.w * j.h * (j.mvmax_x - j.mvmin_x + 1) * (j.mvmax_y - j.mvmin_y + 1) / eff;
        }
        if (taken.size() < (had ? 1u : 2u))
            continue;
        const int nxw = x1 - x0 + 1, nyw = y1 - y0 + 1;
        if ((int64_t)nxw * nyw > (1 << HIER_RIDX_BITS))
            continue;
        const double hier_work = (double)__builtin_popcountll(B.cells) * 64.0 * nxw * nyw;
        const bool had_sad = had && jobs[members[0]].icost == X265AMD_SAD;
        if (!had ? hier_work * 1.25 > old_work : had_sad && hier_work > old_work)
            continue;
        B.uniform = 1;
        for (int i : taken)
            if (jobs[i].mvmin_x != x0 || jobs[i].mvmax_x != x1 || jobs[i].mvmin_y != y0 || jobs[i].mvmax_y != y1)
                B.uniform = 0;
        const x265amd_me_job &j0 = jobs[taken[0]];
        B.ctu_x = gx * 64; B.ctu_y = gy * 64;
        B.mvx0 = x0; B.mvy0 = y0; B.nxw = nxw; B.nyw = nyw;
        B.qmvp_x = j0.qmvp_x; B.qmvp_y = j0.qmvp_y; B.qp = j0.qp; B.ref_idx = j0.ref_idx;
        B.metric = j0.icost;
        // per-job predictors / ranges: the phase tables' extent (HierBundle::pred)
        int lxp = 0, lyp = 0;
        for (int i : taken)
            if (jobs[i].qmvp_x != j0.qmvp_x || jobs[i].qmvp_y != j0.qmvp_y)
                B.pred = 1;
        {
            int qx0 = INT32_MAX, qx1 = INT32_MIN, qy0 = INT32_MAX, qy1 = INT32_MIN;
            B.ulo_x = B.ulo_y = INT32_MAX; B.uhi_x = B.uhi_y = INT32_MIN;
            for (int i : taken)
            {
                const x265amd_me_job &j = jobs[i];
                const int qx = j.qmvp_x >> 2, qy = j.qmvp_y >> 2;
                qx0 = std::min(qx0, qx); qx1 = std::max(qx1, qx); qy0 = std::min(qy0, qy); qy1 = std::max(qy1, qy);
                B.ulo_x = std::min(B.ulo_x, j.mvmin_x - qx); B.uhi_x = std::max(B.uhi_x, j.mvmax_x - qx);
                B.ulo_y = std::min(B.ulo_y, j.mvmin_y - qy); B.uhi_y = std::max(B.uhi_y, j.mvmax_y - qy);
            }
            if (B.pred)
                for (int i : taken)
                {
                    const x265amd_me_job &j = jobs[i];
                    const int qx = j.qmvp_x >> 2, qy = j.qmvp_y >> 2;
                    if (j.mvmin_x - qx != B.ulo_x || j.mvmax_x - qx != B.uhi_x || j.mvmin_y - qy != B.ulo_y || j.mvmax_y - qy != B.uhi_y)
                        B.pred = 2;
                }
            B.dx0 = x0 - qx1; B.dy0 = y0 - qy1;
            B.lx = nxw + 4 + qx1 - qx0; B.ly = nyw + 4 + qy1 - qy0;
            lxp = (B.lx + 3) & ~3; lyp = (B.ly + 3) & ~3;
            // (the hierarchical kernel's 32-bit keys hold raster index + column offset of the job in their low bits)
            if (!had && B.pred && (int64_t)nxw * nyw + B.lx > (1 << HIER_RIDX_BITS))
                continue;
            if (had && B.pred)      // (the candidate-per-lane kernel has one form for per-job ranges and predictors)
                B.uniform = 0;
        }
        int n_rect = 0;
        for (int gk = 0; gk < 21; gk++)
        {
            for (const HierBundle::Rect &r : rects[gk])
                B.rect[n_rect++] = r;
            B.rect_end[gk] = n_rect;
        }
        // slabs of about one lane task per thread
        const int ppd = 4 / bpp;
        const int NC = had_nc(had_sad, bpp);                                    // candidates per lane task (me_had.h)
        const int XG = had ? (nxw + NC - 1) / NC : (nxw + 3) >> 2;              // lane tasks per task row
        const int RT = had ? 1 : HIER_R;                                        // candidate rows per task row
        const int nd = had ? (nxw + 67 + ppd - 1) / ppd + 1 : XG + 16;          // (the widest row read of had_load_row ends 11 pixels behind its first)
        int pitch = nd, best = 64;
        for (int p = nd; p < nd + 32; p++)
        {
            // consecutive task rows of one wavefront continue the bank sequence of the row before
            const int step = had ? (nxw + ppd - 1) / ppd : XG;
            const int d = ((RT * p - step) % 32 + 32) % 32, dd = std::min(d, 32 - d);
            if (dd < best)
            {
                best = dd;
                pitch = p;
            }
        }
        if (!had)
            pitch = hier_pitch(nd);
        // slabs of one lane task per thread (more per thread only when that would exceed the FullBest slots)
        const int nquads = (nyw + RT - 1) / RT, ntasks = nquads * XG;
        const int threads = had ? had_threads : HIER_THREADS;
        int per = threads;
        while ((ntasks + per - 1) / per > FB_STRIDE_PLAN)
            per += threads;

        if (had && per * NC > (1 << HAD_RIDX_BITS))
            continue;
        const int slot_waves = had ? HAD_WAVES : HIER_WAVES;
        const int fixed = (had ? (B.uniform ? 2 * (((nxw + 3) & ~3) + ((nyw + 3) & ~3)) : 8 * (lxp + lyp))
                               : B.pred ? 16 * (lxp + lyp) + 16 * (HIER_PUS + 1) + 32 : 2 * (((nxw + 3) & ~3) + ((nyw + 3) & ~3))) +
                          (had ? 4 * slot_waves * HIER_SMALL + 8 * slot_waves * 5 + 64 : hier_slot_bytes()) +
                          (n_rect ? 8 * (threads / 64) * HIER_MAX_RECT : 0) + (B.quads ? 4 * (threads / 64) * HIER_QUAD_PUS : 0);
        bool fits = true;
        std::vector<HierItem> mine;
        for (int t = 0, slot = 0; t < ntasks; t += per, slot++)
        {
            HierItem it;
            memset(&it, 0, sizeof(it));
            it.bundle = (int)bundles.size();
            it.t_lo = t;
            it.t_hi = std::min(ntasks, t + per);
            it.slot = slot;
            it.pitch = pitch;
            it.win_nd = nd;
            it.win_rows = ((it.t_hi - 1) / XG - it.t_lo / XG + 1) * RT + 63;
            const int bytes = 4 * it.win_rows * pitch + fixed;
            fits = fits && bytes <= FULL_LDS_MAX;
            lds_bytes = std::max(lds_bytes, bytes);
            mine.push_back(it);
        }
        if (!fits)
            continue;
        for (int i : taken)
            in_bundle[i] = 1;
        bundles.push_back(B);
        if (quads_out)
            quads_out->push_back(QJ);
        items.insert(items.end(), mine.begin(), mine.end());
        if (n_rect && had_threads_out)
            *had_threads_out |= 1 << 16;        // (bit 16: some bundle has rectangular PUs)
        if (B.quads && had_threads_out)
            *had_threads_out |= 1 << 17;        // (bit 17: some bundle has PUs with 4- or 12-pixel sides)
      }
    }
    if (bundles.empty())
        lds_bytes = 0;
}

int joblist_build(x265amd_ctx *c, x265amd_joblist *L, const x265amd_me_job *jobs, int n, int bpp)
{
    L->n = n;
    L->bpp = bpp;
    L->launches.clear();
    L->ran = false;
    if (int rc = ensure(c, L->jobs, (size_t)n * sizeof(x265amd_me_job))) return rc;
    if (int rc = ensure(c, L->results, (size_t)n * sizeof(x265amd_me_result))) return rc;
    HIPCHK(c, hipMemcpyAsync(L->jobs.p, jobs, (size_t)n * sizeof(x265amd_me_job), hipMemcpyHostToDevice, c->stream));

    std::vector<char> in_bundle(n, 0);
    std::vector<HierBundle> hbundles;
    std::vector<HierItem> hitems;
    L->n_hier_bundles = L->n_hier_items = L->hier_lds_bytes = 0;
    if (c->hier_ok && !(c->flags & 16)) // (tuning aid: X265AMD_FLAGS bit 4 switches the hierarchical kernel off)
        plan_hier(jobs, n, bpp, false, in_bundle, hbundles, hitems, L->hier_lds_bytes);
    std::vector<HierBundle> dbundles;
    std::vector<HierItem> ditems;
    std::vector<HierQuads> dquads;
    L->n_had_bundles = L->n_had_items = L->had_lds_bytes = 0;
    const std::vector<char> in_hier = in_bundle;
    if (c->hier_ok && !(c->flags & 32)) // (tuning aid: X265AMD_FLAGS bit 5 switches the candidate-per-lane kernel off)
        plan_hier(jobs, n, bpp, true, in_bundle, dbundles, ditems, L->had_lds_bytes, &L->had_threads,
                  (c->flags & 64) ? nullptr : &dquads);     // (bit 6: PUs with 4- or 12-pixel sides stay out of it)
    L->had_rects = ((L->had_threads >> 16) & 1) != 0;
    L->had_quad_pus = ((L->had_threads >> 17) & 1) != 0;
    L->had_threads &= 0xffff;
    if (!ditems.empty())
    {
        xcd_interleave(ditems);
        L->n_had_bundles = (int)dbundles.size();
        L->n_had_items = (int)ditems.size();
        if (int rc = ensure(c, L->had_bundles, dbundles.size() * sizeof(HierBundle))) return rc;
        if (int rc = ensure(c, L->had_items, ditems.size() * sizeof(HierItem))) return rc;
        if (int rc = ensure(c, L->had_fenc, dbundles.size() * (size_t)HAD_FENC_DW * 4)) return rc;
        if (int rc = ensure(c, L->fb_jobs, 2 * (size_t)n)) return rc;
        if (L->had_quad_pus)
        {
            if (int rc = ensure(c, L->had_quads, dquads.size() * sizeof(HierQuads))) return rc;
            HIPCHK(c, hipMemcpyAsync(L->had_quads.p, dquads.data(), dquads.size() * sizeof(HierQuads), hipMemcpyHostToDevice, c->stream));
        }
        L->had_max_depth = 12;
        for (const HierBundle &B : dbundles)
            if (B.metric != X265AMD_SAD)
                L->had_max_depth = 10;
        HIPCHK(c, hipMemcpyAsync(L->had_bundles.p, dbundles.data(), dbundles.size() * sizeof(HierBundle), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(L->had_items.p, ditems.data(), ditems.size() * sizeof(HierItem), hipMemcpyHostToDevice, c->stream));
        // (every SAD job the exhaustive kernels accept is planned below; of the SA8D jobs, those bundled here)
        // second half: the flags of a run whose planes are too deep for this kernel (its bundles fall to the general kernel's scan)
        std::vector<unsigned char> flags(2 * (size_t)n);
        for (int i = 0; i < n; i++)
        {
            flags[i] = full_kernel_eligible(jobs[i], bpp) || in_bundle[i];
            flags[n + i] = full_kernel_eligible(jobs[i], bpp) && !(in_bundle[i] && !in_hier[i]);
        }
        HIPCHK(c, hipMemcpyAsync(L->fb_jobs.p, flags.data(), flags.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));      // flags dies here
    }
    if (!hitems.empty())
    {
        // picture order, dealt to the XCDs in contiguous bands (bundles were made in (row, column) order)
        xcd_interleave(hitems);
        L->n_hier_bundles = (int)hbundles.size();
        L->n_hier_items = (int)hitems.size();
        if (int rc = ensure(c, L->hier_bundles, hbundles.size() * sizeof(HierBundle))) return rc;
        if (int rc = ensure(c, L->hier_items, hitems.size() * sizeof(HierItem))) return rc;
        if (int rc = ensure(c, L->hier_fenc, hbundles.size() * (size_t)HIER_FENC_DW * 4)) return rc;
        HIPCHK(c, hipMemcpyAsync(L->hier_bundles.p, hbundles.data(), hbundles.size() * sizeof(HierBundle), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(L->hier_items.p, hitems.data(), hitems.size() * sizeof(HierItem), hipMemcpyHostToDevice, c->stream));
    }
    if (getenv("X265AMD_PLAN_DEBUG"))       // tuning aid: what the planner made of the list
    {
        int modes[4] = { 0, 0, 0, 0 }, nb = 0;
        for (const HierBundle &B : hbundles)
            modes[B.pred ? 1 + B.pred : B.uniform ? 0 : 1]++;
        for (int i = 0; i < n; i++)
            nb += in_bundle[i];
        fprintf(stderr, "x265amd plan: %d jobs, %d in bundles; shared-window SAD pass: %zu bundles (%d one window, %d per-job ranges, %d per-job "
                "predictors, %d both) in %zu items, LDS %d; candidate-per-lane pass: %zu bundles in %zu items\n", n, nb, hbundles.size(),
                modes[0], modes[1], modes[2], modes[3], hitems.size(), L->hier_lds_bytes, dbundles.size(), ditems.size());
    }
    // job classes and launch order (after the bundles are known: a FULL job whose scan an exhaustive kernel runs leaves
    // only a handful of evaluations to the search kernels)
    {
        std::vector<int> order;
        for (int cls = 0; cls < JOB_CLASSES; cls++)
        {
            std::vector<int> seg;
            for (int i = 0; i < n; i++)
            {
                const int gc = group_job_class(jobs[i], full_kernel_eligible(jobs[i], bpp) || in_bundle[i], (c->flags & 128) ? 0 : c->lane_full_area);
                if ((gc >= 0 ? gc : job_class(jobs[i].w, jobs[i].h)) == cls)
                    seg.push_back(i);
            }
            L->cls_count[cls] = (int)seg.size();
            if (!(c->flags & 8))        // (tuning aid: X265AMD_FLAGS bit 3 keeps the caller's job order)
            {
                if (cls == LANE_CLASS || cls == ROW_CLASS)
                {
                    const size_t per_wave = cls == LANE_CLASS ? 16 : 4;
                    // one job per quad / per 16-lane row of lanes: a wavefront takes 16 / 4 consecutive entries.  Jobs of one shape and search kind
                    // together (the quads of a wavefront then run the same loops), each group in picture order (their
                    // row reads share cache lines); whole wavefronts are dealt to the XCDs
                    std::stable_sort(seg.begin(), seg.end(), [&](int a, int b) {
                        const x265amd_me_job &x = jobs[a], &y = jobs[b];
                        return std::make_tuple((int)x.ref_idx, (int)x.method, (int)x.subme, (int)x.w, (int)x.h, x.pu_y, x.pu_x) <
                               std::make_tuple((int)y.ref_idx, (int)y.method, (int)y.subme, (int)y.w, (int)y.h, y.pu_y, y.pu_x); });
                    std::vector<std::vector<int>> chunks;
                    for (size_t a = 0; a < seg.size(); a += per_wave)
                        chunks.emplace_back(seg.begin() + a, seg.begin() + std::min(seg.size(), a + per_wave));
                    // (only whole chunks may move: the last, partial one stays last)
                    std::vector<std::vector<int>> whole(chunks.begin(), chunks.end() - (chunks.empty() || chunks.back().size() == per_wave ? 0 : 1));
                    xcd_interleave(whole);
                    std::vector<int> flat;
                    for (auto &ch : whole)
                        flat.insert(flat.end(), ch.begin(), ch.end());
                    if (!chunks.empty() && chunks.back().size() != per_wave)
                        flat.insert(flat.end(), chunks.back().begin(), chunks.back().end());
                    seg.swap(flat);
                }
                else
                {
                    std::stable_sort(seg.begin(), seg.end(), [&](int a, int b) {
                        return std::make_tuple((int)jobs[a].ref_idx, jobs[a].pu_y, jobs[a].pu_x) <
                               std::make_tuple((int)jobs[b].ref_idx, jobs[b].pu_y, jobs[b].pu_x); });
                    xcd_interleave(seg);
                }
            }
            order.insert(order.end(), seg.begin(), seg.end());
        }
        if (int rc = ensure(c, L->idx, (size_t)n * sizeof(int))) return rc;
        L->order.swap(order);           // (the list keeps the host copy: it outlives the asynchronous upload)
        HIPCHK(c, hipMemcpyAsync(L->idx.p, L->order.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
        if (L->keep_hist)
        {
            if (int rc = ensure(c, L->hist, (size_t)n * sizeof(unsigned))) return rc;
            HIPCHK(c, hipMemsetAsync(L->hist.p, 0, (size_t)n * sizeof(unsigned), c->stream));
        }
    }
    std::vector<PlannedItem> planned;
    {
        std::map<std::tuple<int, int, int, int, int, int>, std::vector<int>> groups;
        for (int i = 0; i < n; i++)
        {
            const x265amd_me_job &j = jobs[i];
            if (!full_kernel_eligible(j, bpp) || in_bundle[i])
                continue;
            if (j.w * j.h <= 256)
            {
                auto fl = [](int v) { return v >= 0 ? v / PLAN_CELL : -((-v + PLAN_CELL - 1) / PLAN_CELL); };
                groups[std::make_tuple((int)j.ref_idx, (int)j.w | (j.h << 8), fl(j.pu_y), fl(j.pu_x),
                                       (j.pu_x + j.mvmin_x) & 3, 0)].push_back(i);
            }
            else
                plan_group(jobs, bpp, std::vector<int>(1, i), planned);
        }
        for (auto &g : groups)
            plan_group(jobs, bpp, g.second, planned);
    }
    L->has_fb = !planned.empty() || !hitems.empty() || !ditems.empty();
    if (L->has_fb)
    {
        std::vector<FullBest> fb((size_t)n * FB_STRIDE_PLAN);
        for (FullBest &f : fb)
        {
            f.cost = 0x7fffffff;
            f.mvx = f.mvy = 0;
        }
        if (int rc = ensure(c, L->fb, fb.size() * sizeof(FullBest))) return rc;
        HIPCHK(c, hipMemcpyAsync(L->fb.p, fb.data(), fb.size() * sizeof(FullBest), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));      // fb dies here
    }
    if (!planned.empty())
    {
        // class = smallest LDS size that holds the item; launches ordered largest class first
        auto cls = [](const PlannedItem &p) {
            for (int k = 0; k < 5; k++)
                if (p.lds_bytes <= k_lds_classes[k])
                    return k;
            return 4;
        };
        auto w4of = [&](const PlannedItem &p) { return jobs[p.members[0]].w >> 2; };
        // launch order: widest PUs / largest LDS class first (longest work items start first)
        std::stable_sort(planned.begin(), planned.end(), [&](const PlannedItem &a, const PlannedItem &b) {
            return std::make_pair(w4of(a), cls(a)) > std::make_pair(w4of(b), cls(b)); });
        for (size_t a = 0; a < planned.size();)
        {
            size_t b = a;
            int k = cls(planned[a]), w4 = w4of(planned[a]);
            while (b < planned.size() && cls(planned[b]) == k && w4of(planned[b]) == w4)
                b++;
            L->launches.push_back({ (int)a, (int)(b - a), k_lds_classes[k] & ~3, w4 });
            // inside a launch: picture order, dealt to the XCDs in contiguous bands
            std::vector<PlannedItem> seg(planned.begin() + a, planned.begin() + b);
            std::stable_sort(seg.begin(), seg.end(), [](const PlannedItem &x, const PlannedItem &y) {
                return std::make_pair(x.it.win_y0, x.it.win_x0) < std::make_pair(y.it.win_y0, y.it.win_x0); });
            xcd_interleave(seg);
            std::copy(seg.begin(), seg.end(), planned.begin() + a);
            a = b;
        }
        std::vector<FullItem> items;
        std::vector<int> item_jobs;
        for (PlannedItem &p : planned)
        {
            p.it.first = (int)item_jobs.size();
            item_jobs.insert(item_jobs.end(), p.members.begin(), p.members.end());
            items.push_back(p.it);
        }
        if (int rc = ensure(c, L->items, items.size() * sizeof(FullItem))) return rc;
        if (int rc = ensure(c, L->item_jobs, item_jobs.size() * sizeof(int))) return rc;
        HIPCHK(c, hipMemcpyAsync(L->items.p, items.data(), items.size() * sizeof(FullItem), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(L->item_jobs.p, item_jobs.data(), item_jobs.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));      // the host vectors die here
    return 0;
}

void joblist_free(x265amd_ctx *c, x265amd_joblist *L)
{
    hipStreamSynchronize(c->stream);
    for (int i = 0; i < x265amd_ctx::N_AUX; i++)
        hipStreamSynchronize(c->aux[i]);   // (a run on another stream set)
    if (c->out_stream)
        hipStreamSynchronize(c->out_stream);
    if (L->ev_ran) hipEventDestroy(L->ev_ran);
    if (L->ev_out) hipEventDestroy(L->ev_out);
    if (L->ev_done) hipEventDestroy(L->ev_done);
    DevBuf *bufs[] = { &L->jobs, &L->results, &L->fb, &L->items, &L->item_jobs, &L->idx, &L->hist,
                       &L->hier_bundles, &L->hier_items, &L->hier_fenc, &L->had_bundles, &L->had_items, &L->had_fenc, &L->had_quads,
                       &L->fb_jobs };
    for (DevBuf *b : bufs)
        if (b->p)
            hipFree(b->p);
    delete L;
}

// The candidate-per-lane kernel works on packed 16-bit differences and 21-bit costs: usable while every plane of
// the run is at most 10 bits deep (SA8D bundles) or 12 (SAD only)
bool had_usable(x265amd_ctx *c, const x265amd_joblist *L, const BatchParams &P)
{
    if (!L->n_had_items || c->h_planes[P.enc_plane].depth > L->had_max_depth)
        return false;
    for (int r = 0; r < P.n_refs; r++)
        if (c->h_planes[P.ref_planes[r]].depth > L->had_max_depth)
            return false;
    return true;
}

// the exhaustive-kernel launches of a planned list
int joblist_launch_full(x265amd_ctx *c, x265amd_joblist *L, const BatchParams &P)
{
    const bool had = had_usable(c, L, P);
    const int lanes = (int)std::min<size_t>(L->launches.size() + (L->n_hier_items ? 1 : 0) + (had ? 1 : 0), c->n_lanes);
    if (lanes > 1)
        if (int rc = fork_streams(c, lanes)) return rc;
    int k = 0;
    if (L->n_hier_items)
    {
        hipStream_t st = c->lane(k++);
        hipLaunchKernelGGL(k_pack_fenc, dim3(L->n_hier_bundles), dim3(256), 0, st, P, (const HierBundle*)L->hier_bundles.p,
                           L->n_hier_bundles, (uint32_t*)L->hier_fenc.p);
        hipLaunchKernelGGL(k_full_hier<false>, dim3(L->n_hier_items), dim3(HIER_THREADS), L->hier_lds_bytes, st, P,
                           (const x265amd_me_job*)L->jobs.p, (const HierBundle*)L->hier_bundles.p,
                           (const HierItem*)L->hier_items.p, L->n_hier_items, (const uint32_t*)L->hier_fenc.p,
                           (FullBest*)L->fb.p, FB_STRIDE_PLAN);
        HIPCHK(c, hipGetLastError());
    }
    if (had)
    {
        hipStream_t st = c->lane(k++);
#define XA_LAUNCH_HAD_V(PIX, RECTS, QUADS)                                                                         \
        hipLaunchKernelGGL((k_full_had<PIX, RECTS, QUADS>), dim3(L->n_had_items), dim3(L->had_threads), L->had_lds_bytes, st, P, \
                           (const x265amd_me_job*)L->jobs.p, (const HierBundle*)L->had_bundles.p,                  \
                           (const HierItem*)L->had_items.p, L->n_had_items, (const uint32_t*)L->had_fenc.p,        \
                           (const HierQuads*)L->had_quads.p, (FullBest*)L->fb.p, FB_STRIDE_PLAN)
#define XA_LAUNCH_HAD(PIX)                                                                                         \
        hipLaunchKernelGGL(k_pack_fenc16<PIX>, dim3(L->n_had_bundles), dim3(256), 0, st, P,                        \
                           (const HierBundle*)L->had_bundles.p, L->n_had_bundles, (uint32_t*)L->had_fenc.p);       \
        if (L->had_quad_pus)                                                                                       \
            XA_LAUNCH_HAD_V(PIX, true, true);                                                                      \
        else if (L->had_rects)                                                                                     \
            XA_LAUNCH_HAD_V(PIX, true, false);                                                                     \
        else                                                                                                       \
            XA_LAUNCH_HAD_V(PIX, false, false)
        if (L->bpp == 1) { XA_LAUNCH_HAD(uint8_t); } else { XA_LAUNCH_HAD(uint16_t); }
#undef XA_LAUNCH_HAD_V
#undef XA_LAUNCH_HAD
        HIPCHK(c, hipGetLastError());
    }
    for (const x265amd_joblist::Launch &la : L->launches)
    {
        hipStream_t st = c->lane(k++);
#define XA_LAUNCH_FULL_PIX(W4V, PIX)                                                                          \
        hipLaunchKernelGGL((k_full_items<W4V, PIX>), dim3(la.count), dim3(FULL_THREADS), la.lds_bytes, st, P, \
                           (const x265amd_me_job*)L->jobs.p, (const int*)L->item_jobs.p,                      \
                           (const FullItem*)L->items.p + la.first, la.count, (FullBest*)L->fb.p, FB_STRIDE_PLAN, \
                           la.lds_bytes / 4)
#define XA_LAUNCH_FULL(W4V) do { if (L->bpp == 1) XA_LAUNCH_FULL_PIX(W4V, uint8_t); else XA_LAUNCH_FULL_PIX(W4V, uint16_t); } while (0)
        switch (la.w4)
        {
        case 1:  XA_LAUNCH_FULL(1); break;
        case 2:  XA_LAUNCH_FULL(2); break;
        case 3:  XA_LAUNCH_FULL(3); break;
        case 4:  XA_LAUNCH_FULL(4); break;
        case 6:  XA_LAUNCH_FULL(6); break;
        case 8:  XA_LAUNCH_FULL(8); break;
        case 12: XA_LAUNCH_FULL(12); break;
        case 16: XA_LAUNCH_FULL(16); break;
        default: XA_LAUNCH_FULL(0); break;
        }
#undef XA_LAUNCH_FULL
#undef XA_LAUNCH_FULL_PIX
        HIPCHK(c, hipGetLastError());
    }
    if (lanes > 1)
        if (int rc = join_streams(c, lanes)) return rc;
    return 0;
}

// the general-kernel launches of a planned list
int joblist_launch_search(x265amd_ctx *c, x265amd_joblist *L, const BatchParams &P0, bool timing_events)
{
    BatchParams P = P0;
    P.hist = L->keep_hist ? (unsigned*)L->hist.p : nullptr;
    P.fb_jobs = !L->n_had_items ? nullptr : (const unsigned char*)L->fb_jobs.p + (had_usable(c, L, P) ? 0 : L->n);
    L->ran = true;
    return launch_general(c, P, L->bpp, (const x265amd_me_job*)L->jobs.p, L->n,
                          (const int*)L->idx.p, L->cls_count,
                          L->has_fb ? (const FullBest*)L->fb.p : nullptr, FB_STRIDE_PLAN,
                          (x265amd_me_result*)L->results.p, timing_events);
}

int joblist_launch(x265amd_ctx *c, x265amd_joblist *L, const BatchParams &P)
{
    if (L->out_pending)
    {
        HIPCHK(c, hipStreamWaitEvent(c->stream, L->ev_out, 0));      // the last download has read the result buffer
        L->out_pending = false;
    }
    if (int rc = record_start(c)) return rc;
    if (int rc = joblist_launch_full(c, L, P)) return rc;
    c->timed_full = L->has_fb;
    return joblist_launch_search(c, L, P, true);
}


} // namespace

// ---- scratch of the allocation-free entry point (x265amd_me_search_batch_device_noalloc) ------------
// One FullBest, one FullItem and one item-job index per job, for up to NOALLOC_MAX_JOBS jobs, in a
// file-scope device array: every GPU the library runs on holds its own copy (3.25 MiB, and 6.2 MiB of planner tables) at its own
// address until the process ends.  `scratch` is only a host-side stand-in for the symbol: the device
// address is looked up once per device (resolve_scratch, called by x265amd_create with that device
// current) and only that address is ever handed to kernels.
constexpr int NOALLOC_MAX_JOBS = 65536;
constexpr size_t NOALLOC_FB_OFF = 0;
constexpr size_t NOALLOC_ITEMS_OFF = NOALLOC_FB_OFF + sizeof(FullBest) * NOALLOC_MAX_JOBS;
constexpr size_t NOALLOC_IJOBS_OFF = NOALLOC_ITEMS_OFF + sizeof(FullItem) * NOALLOC_MAX_JOBS;
// ... and the device-side planner's tables (me_plan.h) for pictures of up to NOALLOC_MAX_CELLS 64x64 cells times reference
// planes (1080p: 510 per plane): 6.1 KiB per cell.  Larger pictures are searched without bundles, job by job.
constexpr int NOALLOC_MAX_CELLS = 1024;
constexpr size_t NOALLOC_PLAN_OFF = (NOALLOC_IJOBS_OFF + sizeof(int) * NOALLOC_MAX_JOBS + 255) & ~(size_t)255;
constexpr size_t NOALLOC_PLAN_BYTES = 256 * 6 + 4 * (size_t)NOALLOC_MAX_CELLS * HIER_PUS + sizeof(HierBundle) * (size_t)NOALLOC_MAX_CELLS +
                                      4 * ((size_t)NOALLOC_MAX_CELLS + 1) + (size_t)HIER_FENC_DW * 4 * NOALLOC_MAX_CELLS +
                                      NOALLOC_MAX_JOBS;
constexpr size_t NOALLOC_BYTES = NOALLOC_PLAN_OFF + NOALLOC_PLAN_BYTES;
static_assert(NOALLOC_ITEMS_OFF % 8 == 0 && NOALLOC_IJOBS_OFF % 4 == 0, "scratch carve-up alignment");
static __device__ __align__(256) unsigned char scratch[NOALLOC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[MAX_DEVICES];       // device address of `scratch` on device i (NULL: not looked up yet)
// One event per device, recorded behind every call that uses the device's scratch array; the next such call (from any
// context) makes its stream wait for it first, so calls that overlap in time are serialised on the GPU.  Created with
// the address look-up, so the entry point itself creates nothing.
hipEvent_t g_scratch_ev[MAX_DEVICES];

// with `device` current
hipError_t resolve_scratch(int device)
{
    if (device < 0 || device >= MAX_DEVICES)
        return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> g(g_scratch_mu);
    if (g_scratch_ptr[device])
        return hipSuccess;
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess)
        return e;
    if (!p)
        return hipErrorInvalidSymbol;
    e = hipEventCreateWithFlags(&g_scratch_ev[device], hipEventDisableTiming);
    if (e != hipSuccess)
        return e;
    g_scratch_ptr[device] = p;
    return hipSuccess;
}
} // namespace

extern "C" {

int x265amd_partition_from_sizes(int w, int h)
{
    int wi = -1, hi = -1;
    for (int i = 0; i < 8; i++)
    {
        if (k_dims[i] == w) wi = i;
        if (k_dims[i] == h) hi = i;
    }
    return (wi < 0 || hi < 0) ? -1 : (wi << 3) + hi;
}

int x265amd_create(int device, x265amd_ctx **out)
{
    if (!out)
        return X265AMD_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
        return X265AMD_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess)
        return X265AMD_ERR_NO_DEVICE;
    x265amd_ctx *c = new x265amd_ctx;
    c->device = device;
    memset(c->h_planes, 0, sizeof(c->h_planes));
    memset(c->used, 0, sizeof(c->used));
    memset(c->owned, 0, sizeof(c->owned));
    memset(c->base, 0, sizeof(c->base));
    hipError_t e;
    e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    for (int i = 0; i < 3 && e == hipSuccess; i++)
        e = hipEventCreate(&c->ev[i]);
    for (int i = 0; i < x265amd_ctx::N_AUX && e == hipSuccess; i++)
    {
        e = hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming);
    }
    if (e != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc(&c->d_planes, sizeof(c->h_planes))) != hipSuccess ||
        (e = hipHostMalloc(&c->h_planes_pin, sizeof(c->h_planes))) != hipSuccess ||
        (e = hipMalloc(&c->d_cost, (size_t)52 * 65536 * sizeof(uint16_t))) != hipSuccess ||
        (e = hipMalloc(&c->d_evals, EVAL_WORDS * sizeof(unsigned long long))) != hipSuccess)
    {
        fprintf(stderr, "x265amd_create: %s\n", hipGetErrorString(e));
        delete c;
        return X265AMD_ERR_HIP;
    }
    build_cost_tables(c);
    {
        unsigned mx = 0;
        for (int q = 0; q < 52; q++)
            for (uint16_t v : c->h_cost[q])
                mx = std::max(mx, (unsigned)v);
        c->hier_ok = 2 * mx <= HIER_MVCOST_MAX;
    }
    for (int q = 0; q < 52; q++)
        if ((e = hipMemcpy(c->d_cost + (size_t)q * 65536, c->h_cost[q].data(), 65536 * sizeof(uint16_t), hipMemcpyHostToDevice)) != hipSuccess)
        {
            delete c;
            return X265AMD_ERR_HIP;
        }
    hipMemset(c->d_evals, 0, EVAL_WORDS * sizeof(unsigned long long));
    // the exhaustive kernel uses up to 64 KiB of dynamic LDS
#define XA_FULL_KERNELS(PIX) (const void*)k_full_items<0, PIX>, (const void*)k_full_items<1, PIX>, (const void*)k_full_items<2, PIX>, \
                             (const void*)k_full_items<3, PIX>, (const void*)k_full_items<4, PIX>, (const void*)k_full_items<6, PIX>, \
                             (const void*)k_full_items<8, PIX>, (const void*)k_full_items<12, PIX>, (const void*)k_full_items<16, PIX>
    const void *full_kernels[] = { XA_FULL_KERNELS(uint8_t), XA_FULL_KERNELS(uint16_t), (const void*)k_full_hier<false>,
                                   (const void*)k_full_hier<true> };
#undef XA_FULL_KERNELS
    e = hipSuccess;
    for (const void *k : full_kernels)
        if (e == hipSuccess)
            e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, FULL_LDS_MAX);
    if (e != hipSuccess)
    {
        fprintf(stderr, "x265amd_create: kernels not loadable on this device: %s\n", hipGetErrorString(e));
        delete c;
        return X265AMD_ERR_NO_DEVICE;
    }
    if (const char *ev = getenv("X265AMD_FLAGS"))        // tuning aid, see BatchParams::flags
    {
        c->flags = atoi(ev);
        g_no_xcd_bands = (c->flags & 4) != 0;
        g_hier_one_pred = (c->flags & 256) != 0;
    }
    if (const char *ev = getenv("X265AMD_LANE_FULL"))    // tuning aid: see lane_full_area (at least LANE_MAX_AREA)
        c->lane_full_area = std::max(LANE_MAX_AREA, std::min(4096, atoi(ev)));
    if (const char *ev = getenv("X265AMD_LANES"))        // streams the independent kernels of a batch are spread over
        c->n_lanes = std::max(1, std::min(x265amd_ctx::N_AUX + 1, atoi(ev)));
    if ((e = resolve_scratch(device)) != hipSuccess)
    {
        fprintf(stderr, "x265amd_create: address of the scratch array: %s\n", hipGetErrorString(e));
        x265amd_destroy(c);
        return X265AMD_ERR_HIP;
    }

    *out = c;
    return X265AMD_OK;
}

void x265amd_destroy(x265amd_ctx *c)
{
    if (!c)
        return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->up_stream)
    {
        hipStreamSynchronize(c->up_stream);
        if (c->up_stream_owned)
            hipStreamDestroy(c->up_stream);
        hipEventDestroy(c->ev_mark);
    }
    for (int i = 0; i < MAX_PLANES; i++)
        if (c->used[i] && c->owned[i])
            hipFree(c->owned[i]);
    if (c->batch_list)
    {
        joblist_free(c, c->batch_list);
        c->batch_list = nullptr;
    }
    DevBuf *bufs[] = { &c->jobs, &c->results, &c->fullbest, &c->pairs, &c->costs, &c->stage, &c->items, &c->item_jobs, &c->plan_mem,
                       &c->look_pairs, &c->look_results, &c->look_progress, &c->field_results, &c->field_mvp, &c->field_col };
    for (DevBuf *b : bufs)
        if (b->p)
            hipFree(b->p);
    if (c->h_stage)
        hipHostFree(c->h_stage);
    for (int i = 0; i < 3; i++)
        if (c->ev[i])
            hipEventDestroy(c->ev[i]);
    hipFree(c->d_planes);
    if (c->h_planes_pin)
        hipHostFree(c->h_planes_pin);
    hipFree(c->d_cost);
    hipFree(c->d_evals);
    for (int i = 0; i < x265amd_ctx::N_AUX; i++)
    {
        if (c->aux[i]) hipStreamDestroy(c->aux[i]);
        if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->out_stream) hipStreamDestroy(c->out_stream);
    for (int i = 0; i < MAX_PLANES; i++)
    {
        if (c->plane_filled[i])
            hipEventDestroy(c->plane_filled[i]);
        for (int st = 0; st < 8; st++)
            if (c->plane_read[i][st])
                hipEventDestroy(c->plane_read[i][st]);
    }
    hipStreamDestroy(c->stream);
    delete c;
}

const char *x265amd_last_error(const x265amd_ctx *c) { return c ? c->err.c_str() : "no context"; }

int x265amd_sync(x265amd_ctx *c)
{
    if (!c)
        return X265AMD_ERR_INVALID;
    // (the streams are read under the lock - a call that queues on another stream set exchanges them while it holds it -
    // and waited for outside it, so that other threads can keep queueing)
    hipStream_t st[1 + x265amd_ctx::N_AUX], out, up;
    {
        std::lock_guard<std::mutex> g(c->mu);
        st[0] = c->stream;
        for (int i = 0; i < x265amd_ctx::N_AUX; i++)
            st[1 + i] = c->aux[i];                            // runs on the other stream sets
        out = c->out_stream;                                  // asynchronous result downloads
        up = c->up_stream;                                    // asynchronous plane refills
    }
    for (hipStream_t s : st)
        HIPCHK(c, hipStreamSynchronize(s));
    if (out)
        HIPCHK(c, hipStreamSynchronize(out));
    if (up)
        HIPCHK(c, hipStreamSynchronize(up));
    return 0;
}

void *x265amd_stream(x265amd_ctx *c)
{
    if (!c)
        return nullptr;
    std::lock_guard<std::mutex> g(c->mu);
    return (void*)c->stream;
}

int x265amd_debug_state(x265amd_ctx *c, char *buf, size_t n)
{
    if (!c || !buf || n == 0)
        return X265AMD_ERR_INVALID;
    // (no lock: meant to be called from a watchdog thread while another thread sits in x265amd_sync)
    auto q = [](hipError_t e) { return e == hipSuccess ? "done" : e == hipErrorNotReady ? "BUSY" : hipGetErrorString(e); };
    int at = snprintf(buf, n, "main stream %s", q(hipStreamQuery(c->stream)));
    for (int i = 0; i < x265amd_ctx::N_AUX && at < (int)n; i++)
        at += snprintf(buf + at, n - at, "; aux%d %s (join event %s)", i, q(hipStreamQuery(c->aux[i])), q(hipEventQuery(c->ev_join[i])));
    if (at < (int)n)
        at += snprintf(buf + at, n - at, "; fork event %s; lanes %d", q(hipEventQuery(c->ev_fork)), c->n_lanes);
    return 0;
}

int x265amd_wait_event_on(x265amd_ctx *c, void *event, int set)
{
    if (!c || !event || set < 0 || set > 7)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    StreamSet on(c, set);
    HIPCHK(c, hipStreamWaitEvent(c->stream, (hipEvent_t)event, 0));
    return 0;
}

int x265amd_record_event_on(x265amd_ctx *c, void *event, int set)
{
    if (!c || !event || set < 0 || set > 7)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    StreamSet on(c, set);
    HIPCHK(c, hipEventRecord((hipEvent_t)event, c->stream));
    return 0;
}

int x265amd_wait_event(x265amd_ctx *c, void *event) { return x265amd_wait_event_on(c, event, 0); }
int x265amd_record_event(x265amd_ctx *c, void *event) { return x265amd_record_event_on(c, event, 0); }

int x265amd_plane_geometry(x265amd_ctx *c, int id, int *width, int *height, int *margin_x, int *margin_y,
                           intptr_t *stride, int *bytes_per_px, int *bit_depth)
{
    if (!c || id < 0 || id >= MAX_PLANES || !c->used[id])
        return X265AMD_ERR_INVALID;
    const PlaneDesc &d = c->h_planes[id];
    if (width)        *width = d.width;
    if (height)       *height = d.height;
    if (margin_x)     *margin_x = d.margin_x;
    if (margin_y)     *margin_y = d.margin_y;
    if (stride)       *stride = d.stride;
    if (bytes_per_px) *bytes_per_px = d.bpp;
    if (bit_depth)    *bit_depth = d.depth;
    return 0;
}

int x265amd_plane_upload(x265amd_ctx *c, const void *host_base, int width, int height, int margin_x, int margin_y,
                         intptr_t stride, int bpp, int depth, int *plane_id)
{
    if (!c || !host_base || !plane_id)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_plane_args(c, width, height, margin_x, margin_y, stride, bpp, depth))
        return rc;
    size_t bytes = (size_t)(height + 2 * margin_y) * stride * bpp;
    void *d = nullptr;
    HIPCHK(c, hipMalloc(&d, bytes + 64));     // slack so that clamped dword reads at the very end stay inside
    hipError_t e = hipMemcpyAsync(d, host_base, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess)
    {
        hipFree(d);
        return fail(c, X265AMD_ERR_HIP, "plane upload", e);
    }
    int rc = add_plane(c, d, d, width, height, margin_x, margin_y, stride, bpp, depth, plane_id);
    if (rc)
        hipFree(d);
    return rc;
}

int x265amd_plane_wrap_device(x265amd_ctx *c, void *dev_base, int width, int height, int margin_x, int margin_y,
                              intptr_t stride, int bpp, int depth, int *plane_id)
{
    if (!c || !dev_base || !plane_id)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    if (int rc = check_plane_args(c, width, height, margin_x, margin_y, stride, bpp, depth))
        return rc;
    return add_plane(c, dev_base, nullptr, width, height, margin_x, margin_y, stride, bpp, depth, plane_id);
}

int x265amd_plane_create(x265amd_ctx *c, int width, int height, int margin_x, int margin_y, intptr_t stride, int bpp,
                         int depth, int *plane_id)
{
    if (!c || !plane_id)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_plane_args(c, width, height, margin_x, margin_y, stride, bpp, depth))
        return rc;
    size_t bytes = (size_t)(height + 2 * margin_y) * stride * bpp;
    void *d = nullptr;
    HIPCHK(c, hipMalloc(&d, bytes + 64));
    hipError_t e = hipMemsetAsync(d, 0, bytes + 64, c->stream);
    if (e != hipSuccess)
    {
        hipFree(d);
        return fail(c, X265AMD_ERR_HIP, "plane clear", e);
    }
    int rc = add_plane(c, d, d, width, height, margin_x, margin_y, stride, bpp, depth, plane_id);
    if (rc)
        hipFree(d);
    return rc;
}

int x265amd_plane_update_rows(x265amd_ctx *c, int id, const void *host_origin, intptr_t host_stride, int y0, int n_rows)
{
    if (!c || !host_origin || id < 0 || id >= MAX_PLANES || !c->used[id])
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const PlaneDesc &d = c->h_planes[id];
    if (d.lr_step || y0 < 0 || n_rows <= 0 || y0 + n_rows > d.height || host_stride < d.width)
        return fail(c, X265AMD_ERR_INVALID, "bad row range (or a half-resolution set)");
    char *org = (char*)const_cast<void*>(d.origin);
    HIPCHK(c, hipMemcpy2DAsync(org + (size_t)y0 * d.stride * d.bpp, (size_t)d.stride * d.bpp,
                               (const char*)host_origin + (size_t)y0 * host_stride * d.bpp, (size_t)host_stride * d.bpp,
                               (size_t)d.width * d.bpp, (size_t)n_rows, hipMemcpyHostToDevice, c->stream));
    const int first = y0 == 0 ? -d.margin_y : y0, last = y0 + n_rows == d.height ? d.height + d.margin_y : y0 + n_rows;
    const int threads = (last - first) * (d.width + 2 * d.margin_x);
    const dim3 grid((threads + 255) / 256), block(256);
    if (d.bpp == 1)
        hipLaunchKernelGGL(k_extend_rows<uint8_t>, grid, block, 0, c->stream, (uint8_t*)org, d.stride, d.width, d.height,
                           d.margin_x, d.margin_y, y0, n_rows);
    else
        hipLaunchKernelGGL(k_extend_rows<uint16_t>, grid, block, 0, c->stream, (uint16_t*)org, d.stride, d.width, d.height,
                           d.margin_x, d.margin_y, y0, n_rows);
    HIPCHK(c, hipGetLastError());
    // the host rows are borrowed only during the call when they are pageable; a pinned source must stay
    // valid until the stream reaches the copy, so wait here: row publishing is a per-CTU-row event
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int x265amd_plane_refill_async(x265amd_ctx *c, int id, const void *host_base)
{
    if (!c || !host_base || id < 0 || id >= MAX_PLANES || !c->used[id])
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const PlaneDesc &d = c->h_planes[id];
    if (d.lr_step)
        return fail(c, X265AMD_ERR_INVALID, "a half-resolution set is refilled with x265amd_lowres_update");
    if (!c->up_stream)
    {
        // (a process has few hardware queues, and streams beyond them share one - in packet order, which would put the
        // searches behind the uploads: take the context's last auxiliary stream when the launches do not use it)
        if (c->n_lanes <= x265amd_ctx::N_AUX)
            c->up_stream = c->aux[x265amd_ctx::N_AUX - 1];
        else
        {
            HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
            c->up_stream_owned = true;
        }
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_mark, hipEventDisableTiming));
    }
    if (!c->plane_filled[id])
        HIPCHK(c, hipEventCreateWithFlags(&c->plane_filled[id], hipEventDisableTiming));
    // behind everything queued so far (the searches that read the plane's old content), beside everything queued later
    // that does not name this plane
    HIPCHK(c, hipEventRecord(c->ev_mark, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->up_stream, c->ev_mark, 0));
    // ... and behind the runs on the other stream sets that named this plane
    for (int st = 0; st < 8; st++)
        if (c->read_pending[id] & (1u << st))
            HIPCHK(c, hipStreamWaitEvent(c->up_stream, c->plane_read[id][st], 0));
    c->read_pending[id] = 0;
    const size_t bytes = (size_t)(d.hi - d.lo) * d.bpp;
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, const_cast<void*>(host_base), 0) == hipSuccess && mapped)
    {
        // page-locked, device-mapped memory: read over the bus by a copy kernel (no DMA submission on the host's path)
        hipLaunchKernelGGL(k_copy_bytes, dim3(COPY_WGS), dim3(256), 0, c->up_stream,
                           (const unsigned char*)mapped, (unsigned char*)c->base[id], bytes);
        HIPCHK(c, hipGetLastError());
    }
    else
    {
        (void)hipGetLastError();
        HIPCHK(c, hipMemcpyAsync(c->base[id], host_base, bytes, hipMemcpyHostToDevice, c->up_stream));
    }
    HIPCHK(c, hipEventRecord(c->plane_filled[id], c->up_stream));
    c->fill_pending[id] = 255;
    return 0;
}

void *x265amd_plane_device_ptr(x265amd_ctx *c, int id)
{
    if (!c || id < 0 || id >= MAX_PLANES || !c->used[id])
        return nullptr;
    return c->base[id];
}

int x265amd_plane_release(x265amd_ctx *c, int id)
{
    if (!c || id < 0 || id >= MAX_PLANES || !c->used[id])
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < x265amd_ctx::N_AUX; i++)
        HIPCHK(c, hipStreamSynchronize(c->aux[i]));           // (runs on the other stream sets may read the plane)
    if (c->up_stream)
        HIPCHK(c, hipStreamSynchronize(c->up_stream));
    c->fill_pending[id] = 0;
    c->read_pending[id] = 0;
    if (c->owned[id])
        hipFree(c->owned[id]);
    c->used[id] = false;
    c->owned[id] = nullptr;
    c->base[id] = nullptr;
    memset(&c->h_planes[id], 0, sizeof(PlaneDesc));
    c->planes_dirty = true;
    return 0;
}

// ---- lookahead (half-resolution) path ------------------------------------------------------

namespace {

// The chained kernels report a row that gave up waiting by OR-ing into an error word: one word for the lookahead
// searches, one for the predictor-chained fields, zeroed when first allocated and then only when they are read out.
constexpr int ERR_WORD_LOOKAHEAD = 0, ERR_WORD_FIELD = 1;
int ensure_error_words(x265amd_ctx *c)
{
    if (c->look_progress.p)
        return 0;
    if (int rc = ensure(c, c->look_progress, 2 * sizeof(int))) return rc;
    HIPCHK(c, hipMemsetAsync(c->look_progress.p, 0, 2 * sizeof(int), c->stream));
    return 0;
}

// geometry of Lowres::create (common/lowres.cpp:30-58) for a source plane
struct LowresGeom { int width, lines, stride, cu_w, cu_h; int64_t elems; };

LowresGeom lowres_geom(const PlaneDesc &s)
{
    LowresGeom g;
    int width = s.width / 2, lines = s.height / 2;
    g.stride = width + 2 * s.margin_x;
    if (g.stride & 31)
        g.stride += 32 - (g.stride & 31);
    g.cu_w = (width + 7) >> 3;
    g.cu_h = (lines + 7) >> 3;
    g.width = g.cu_w * 8;
    g.lines = g.cu_h * 8;
    g.elems = (int64_t)g.stride * (g.lines + 2 * s.margin_y);
    return g;
}

int lowres_fill(x265amd_ctx *c, int lowres_plane, int src_plane)
{
    const PlaneDesc &s = c->h_planes[src_plane], &d = c->h_planes[lowres_plane];
    const int rows = d.height + 2 * d.margin_y;
    const int threads = (d.stride / 4) * rows;
    const dim3 grid((threads + 255) / 256), block(256);
    if (s.bpp == 1)
        hipLaunchKernelGGL(k_lowres_init<uint8_t>, grid, block, 0, c->stream, (const uint8_t*)s.origin, s.lo, s.hi, s.stride,
                           (uint8_t*)c->base[lowres_plane], d.stride, rows, d.width, d.height, d.margin_x, d.margin_y, d.lr_step);
    else
        hipLaunchKernelGGL(k_lowres_init<uint16_t>, grid, block, 0, c->stream, (const uint16_t*)s.origin, s.lo, s.hi, s.stride,
                           (uint16_t*)c->base[lowres_plane], d.stride, rows, d.width, d.height, d.margin_x, d.margin_y, d.lr_step);
    HIPCHK(c, hipGetLastError());
    return 0;
}

int check_lowres_source(x265amd_ctx *c, int src_plane)
{
    if (src_plane < 0 || src_plane >= MAX_PLANES || !c->used[src_plane] || c->h_planes[src_plane].lr_step)
        return fail(c, X265AMD_ERR_INVALID, "source is not a registered full-resolution plane");
    const PlaneDesc &s = c->h_planes[src_plane];
    if (s.width < 16 || s.height < 16)
        return fail(c, X265AMD_ERR_INVALID, "source picture too small");
    // the downscale reads up to 16 pixels beyond the picture; searches leave the half-resolution
    // picture by up to 8 + 3 pixels plus one for the neighbouring half-pel plane
    if (s.margin_x < 16 || s.margin_y < 16)
        return fail(c, X265AMD_ERR_RANGE, "source margins below 16 pixels");
    return 0;
}

} // namespace

int x265amd_lowres_create(x265amd_ctx *c, int src_plane, int *lowres_plane)
{
    if (!c || !lowres_plane)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_lowres_source(c, src_plane)) return rc;
    const PlaneDesc s = c->h_planes[src_plane];
    const LowresGeom lg = lowres_geom(s);
    void *d = nullptr;
    HIPCHK(c, hipMalloc(&d, (size_t)(4 * lg.elems) * s.bpp + 64));
    int id = -1;
    int rc = add_plane(c, d, d, lg.width, lg.lines, s.margin_x, s.margin_y, lg.stride, s.bpp, s.depth, &id);
    if (rc)
    {
        hipFree(d);
        return rc;
    }
    c->h_planes[id].lr_step = lg.elems;
    rc = lowres_fill(c, id, src_plane);
    if (rc)
    {
        // give the slot and the memory back
        hipStreamSynchronize(c->stream);
        hipFree(d);
        c->used[id] = false;
        c->owned[id] = nullptr;
        c->base[id] = nullptr;
        memset(&c->h_planes[id], 0, sizeof(PlaneDesc));
        c->planes_dirty = true;
        return rc;
    }
    *lowres_plane = id;
    return 0;
}

int x265amd_lowres_update(x265amd_ctx *c, int lowres_plane, int src_plane)
{
    if (!c)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_lowres_source(c, src_plane)) return rc;
    if (lowres_plane < 0 || lowres_plane >= MAX_PLANES || !c->used[lowres_plane] || !c->h_planes[lowres_plane].lr_step)
        return fail(c, X265AMD_ERR_INVALID, "not a half-resolution plane set");
    const PlaneDesc &s = c->h_planes[src_plane], &d = c->h_planes[lowres_plane];
    const LowresGeom lg = lowres_geom(s);
    if (lg.width != d.width || lg.lines != d.height || lg.stride != d.stride || s.bpp != d.bpp ||
        s.margin_x != d.margin_x || s.margin_y != d.margin_y)
        return fail(c, X265AMD_ERR_INVALID, "source geometry does not match the half-resolution set");
    return lowres_fill(c, lowres_plane, src_plane);
}

int x265amd_lowres_geometry(x265amd_ctx *c, int lowres_plane, int *width, int *lines, intptr_t *stride, int *cu_w, int *cu_h)
{
    if (!c || lowres_plane < 0 || lowres_plane >= MAX_PLANES || !c->used[lowres_plane] || !c->h_planes[lowres_plane].lr_step)
        return X265AMD_ERR_INVALID;
    const PlaneDesc &d = c->h_planes[lowres_plane];
    if (width)  *width = d.width;
    if (lines)  *lines = d.height;
    if (stride) *stride = d.stride;
    if (cu_w)   *cu_w = d.width / 8;
    if (cu_h)   *cu_h = d.height / 8;
    return 0;
}

int x265amd_plane_download(x265amd_ctx *c, int plane, int which, void *host, size_t bytes)
{
    if (!c || !host || plane < 0 || plane >= MAX_PLANES || !c->used[plane])
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    const PlaneDesc &d = c->h_planes[plane];
    if (which < 0 || which > (d.lr_step ? 3 : 0))
        return fail(c, X265AMD_ERR_INVALID, "no such plane in the set");
    const size_t have = (size_t)(d.hi - d.lo) * d.bpp;
    if (bytes > have)
        return fail(c, X265AMD_ERR_INVALID, "download larger than the plane");
    HIPCHK(c, hipMemcpyAsync(host, (const char*)c->base[plane] + (size_t)which * d.lr_step * d.bpp, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int x265amd_lookahead_me(x265amd_ctx *c, int n_pairs, const int *enc_lowres, const int *ref_lowres, int merange,
                         int method, int subme, int qp, x265amd_me_result *out)
{
    if (!c || !enc_lowres || !ref_lowres || n_pairs < 0)
        return X265AMD_ERR_INVALID;
    if (n_pairs == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (method < X265AMD_DIA || method > X265AMD_FULL || subme < 0 || subme > 7 || qp < 0 || qp > 51 || merange < 0 || merange > 32767)
        return fail(c, X265AMD_ERR_INVALID, "lookahead search parameter out of range");
    std::vector<LookPair> hp(n_pairs);
    const PlaneDesc *first = nullptr;
    for (int i = 0; i < n_pairs; i++)
    {
        for (int id : { enc_lowres[i], ref_lowres[i] })
        {
            if (id < 0 || id >= MAX_PLANES || !c->used[id] || !c->h_planes[id].lr_step)
                return fail(c, X265AMD_ERR_INVALID, "lookahead search needs half-resolution plane sets");
            const PlaneDesc &d = c->h_planes[id];
            if (!first)
                first = &d;
            if (d.width != first->width || d.height != first->height || d.stride != first->stride || d.bpp != first->bpp)
                return fail(c, X265AMD_ERR_INVALID, "half-resolution sets of one call must share their geometry");
        }
        hp[i].enc_plane = enc_lowres[i];
        hp[i].ref_plane = ref_lowres[i];
    }
    const int cu_w = first->width / 8, cu_h = first->height / 8;
    const size_t n_res = (size_t)n_pairs * cu_w * cu_h;
    if (int rc = sync_planes(c)) return rc;
    if (int rc = ensure(c, c->look_pairs, hp.size() * sizeof(LookPair))) return rc;
    if (int rc = ensure(c, c->look_results, n_res * sizeof(x265amd_me_result))) return rc;
    if (int rc = ensure_error_words(c)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->look_pairs.p, hp.data(), hp.size() * sizeof(LookPair), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // hp goes out of scope
    HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)c->look_results.p, LOOK_UNSET, n_res * 2, c->stream));   // "not searched yet"
    if (int rc = record_start(c)) return rc;
    if (c->timing)
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    int *error = (int*)c->look_progress.p + ERR_WORD_LOOKAHEAD;
    const uint16_t *cost = c->d_cost + (size_t)qp * 65536 + 32768;
    const dim3 grid(n_pairs * cu_h), block(64);
    if (first->bpp == 1)
        hipLaunchKernelGGL((k_lookahead<JobLds<uint8_t, 1, 64, true, true>>), grid, block, 0, c->stream, c->d_planes, cost,
                           (const LookPair*)c->look_pairs.p, cu_w, cu_h, merange, method, subme, qp, c->flags,
                           (x265amd_me_result*)c->look_results.p, error, c->d_evals);
    else
        hipLaunchKernelGGL((k_lookahead<JobLds<uint16_t, 1, 64, true, true>>), grid, block, 0, c->stream, c->d_planes, cost,
                           (const LookPair*)c->look_pairs.p, cu_w, cu_h, merange, method, subme, qp, c->flags,
                           (x265amd_me_result*)c->look_results.p, error, c->d_evals);
    HIPCHK(c, hipGetLastError());
    if (c->timing)
    {
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
        c->timed_any = true;
    }
    if (!out)
        return 0;
    int h_err = 0;
    HIPCHK(c, hipMemcpyAsync(out, c->look_results.p, n_res * sizeof(x265amd_me_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&h_err, error, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_err)
    {
        HIPCHK(c, hipMemsetAsync(error, 0, sizeof(int), c->stream));
        return fail(c, X265AMD_ERR_HIP, "lookahead search: a CU row waited too long for the row below it");
    }
    return 0;
}

const x265amd_me_result *x265amd_lookahead_results_device(const x265amd_ctx *c)
{
    return c ? (const x265amd_me_result*)c->look_results.p : nullptr;
}

extern "C++" {
namespace {
template <typename PIX>
void launch_field(x265amd_ctx *c, hipStream_t st, int cls, int n_wgs, int cp_waves, const uint16_t *cost, const FieldDesc *fields, int n_fields,
                  int pic_w, int pic_h, int merange, int method, int subme, int qp, const x265amd_me_result *col,
                  x265amd_me_result *res, int *mvp, int *error)
{
#define XA_FIELD(NW, MAXG) hipLaunchKernelGGL((k_field<JobLds<PIX, NW, MAXG, false, NW == 1, NW == 1>>), dim3(n_wgs), dim3(NW == 1 ? 64 * cp_waves : NW * 64), 0, st, c->d_planes, cost, \
                                              fields, n_fields, pic_w, pic_h, 64, merange, method, subme, qp, c->flags,   \
                                              col, res, mvp, error, c->d_evals)
    if (cls == 0)      XA_FIELD(1, 64);
    else if (cls == 1) XA_FIELD(1, 256);
    else               XA_FIELD(8, 1024);
#undef XA_FIELD
}

// PU grid of a field: PUs per row, PU rows (whole CUs only: an odd last PU row / column of a 2NxN / Nx2N grid is left out; the
// asymmetric modes 4..7 take the CU's size and have two PU rows (4, 5) or two PU columns (6, 7) per CU)
bool field_grid(int pic_w, int pic_h, int pu_w, int pu_h, int part_mode, int &gw, int &gh)
{
    gw = pic_w / pu_w; gh = pic_h / pu_h;
    if (part_mode == 1) gh &= ~1;
    if (part_mode == 2) gw &= ~1;
    if (part_mode == 4 || part_mode == 5) gh *= 2;
    if (part_mode == 6 || part_mode == 7) gw *= 2;
    return gw >= 1 && gh >= 1;
}
bool field_two_rows(int part_mode) { return part_mode == 1 || part_mode == 4 || part_mode == 5; }   // upper / lower parts
} // namespace
} // extern "C++"

int x265amd_field_me_multi(x265amd_ctx *c, int n_fields, const int *enc_planes, const int *ref_planes, const int *pu_w, const int *pu_h,
                           const int *part_mode, int merange, int method, int subme, int qp, x265amd_me_result *out, int16_t *out_mvp)
{
    return x265amd_field_me_multi_tmvp(c, n_fields, enc_planes, ref_planes, pu_w, pu_h, part_mode, nullptr, merange, method, subme, qp, out,
                                       out_mvp);
}

int x265amd_field_me_multi_tmvp(x265amd_ctx *c, int n_fields, const int *enc_planes, const int *ref_planes, const int *pu_w, const int *pu_h,
                                const int *part_mode, const x265amd_me_result *col, int merange, int method, int subme, int qp,
                                x265amd_me_result *out, int16_t *out_mvp)
{
    if (!c || n_fields < 1 || n_fields > 4096 || !enc_planes || !ref_planes || !pu_w || !pu_h)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (method < X265AMD_DIA || method > X265AMD_FULL || subme < 0 || subme > 7 || qp < 0 || qp > 51 || merange < 0 || merange > 32767)
        return fail(c, X265AMD_ERR_INVALID, "field search parameter out of range");
    std::vector<size_t> first(n_fields + 1, 0);         // first result of field f
    for (int f = 0; f < n_fields; f++)
    {
        const int pm = part_mode ? part_mode[f] : 0;
        if (!valid_dim(pu_w[f]) || !valid_dim(pu_h[f]) || pm < 0 || pm > 7 || pm == 3 || (method == X265AMD_UMH && pu_w[f] == 4 && pu_h[f] == 4) ||
            (pm == 1 && 2 * pu_h[f] > 64) || (pm == 2 && 2 * pu_w[f] > 64) ||
            (pm >= 4 && (pu_w[f] != pu_h[f] || (pu_w[f] != 16 && pu_w[f] != 32 && pu_w[f] != 64))))
            return fail(c, X265AMD_ERR_INVALID, "field search parameter out of range");
        const int refs[1] = { ref_planes[f] };
        if (int rc = check_batch_planes(c, enc_planes[f], refs, 1)) return rc;
        const PlaneDesc &ef = c->h_planes[enc_planes[f]], &r = c->h_planes[ref_planes[f]], &e0 = c->h_planes[enc_planes[0]];
        if (ef.lr_step || r.lr_step || ef.width != r.width || ef.height != r.height)
            return fail(c, X265AMD_ERR_INVALID, "field search needs two full-resolution planes of one picture size");
        if (ef.width != e0.width || ef.height != e0.height || ef.bpp != e0.bpp)
            return fail(c, X265AMD_ERR_INVALID, "the fields of one call must share picture size and pixel type");
        // clipMv lets a block reach 64 + 8 - 1 pixels beyond the picture; plus pattern overshoot and the filter apron
        if (r.margin_x < 71 + 3 + 4 || r.margin_y < 71 + 3 + 4)
            return fail(c, X265AMD_ERR_RANGE, "reference margins below 78 pixels");
        int gw, gh;
        if (!field_grid(ef.width, ef.height, pu_w[f], pu_h[f], pm, gw, gh))
            return fail(c, X265AMD_ERR_INVALID, "picture smaller than one CU");
        first[f + 1] = first[f] + (size_t)gw * gh;
    }
    const size_t n = first[n_fields];
    const PlaneDesc &e = c->h_planes[enc_planes[0]];
    if (int rc = sync_planes(c)) return rc;
    if (int rc = ensure(c, c->field_results, n * sizeof(x265amd_me_result))) return rc;
    if (int rc = ensure(c, c->field_mvp, n * sizeof(int))) return rc;
    if (int rc = ensure_error_words(c)) return rc;
    // One launch per kernel class (PUs of at most 256 / 1024 / 4096 pixels; the larger part of an asymmetric CU, three quarters of
    // it, decides): fields of different PU sizes and partition modes share a launch through their descriptors.  A field's time is
    // the length of its chain, not its work, so all fields of the call are in flight side by side, the launches on different
    // streams; inside a launch the longest chains come first (a chain's length in PU steps is a row's PUs plus the two-CU lag of
    // every further row, and a step takes about 12 + 0.75 sqrt(PU area) microseconds - measured 16 us at 8x8, 24 at 16x16, 39 at
    // 32x32, 62 at 64x64).
    const size_t col_per = (size_t)((e.width + 15) >> 4) * ((e.height + 15) >> 4);
    std::vector<FieldDesc> desc[3];
    std::vector<double> len(n_fields);
    for (int cls = 0; cls < 3; cls++)
    {
        std::vector<int> of_cls;
        for (int f = 0; f < n_fields; f++)
        {
            const int pm = part_mode ? part_mode[f] : 0;
            if ((pm >= 4 ? job_class(pu_w[f], pu_h[f] - (pu_h[f] >> 2)) : job_class(pu_w[f], pu_h[f])) != cls)
                continue;
            int gw, gh;
            field_grid(e.width, e.height, pu_w[f], pu_h[f], pm, gw, gh);
            const int parts = field_two_rows(pm) ? 2 : 1;
            // (the two parts of an asymmetric CU together have the CU's area: half of it each on average)
            const double area = (double)pu_w[f] * pu_h[f] * (pm >= 4 ? 0.5 : 1.0);
            len[f] = parts * (gw + 2 * (gh / parts)) * (12.0 + 0.75 * sqrt(area));
            of_cls.push_back(f);
        }
        std::stable_sort(of_cls.begin(), of_cls.end(), [&](int a, int b) { return len[a] > len[b]; });
        int wg = 0;
        for (int f : of_cls)
        {
            const int pm = part_mode ? part_mode[f] : 0;
            int gw, gh;
            field_grid(e.width, e.height, pu_w[f], pu_h[f], pm, gw, gh);
            FieldDesc d;
            d.enc_plane = enc_planes[f]; d.ref_plane = ref_planes[f];
            d.pu_w = pu_w[f]; d.pu_h = pu_h[f]; d.gw = gw; d.wg_rows = field_two_rows(pm) ? gh / 2 : gh; d.part_mode = pm;
            d.first_wg = wg; d.first_out = (int32_t)first[f]; d.first_col = col ? (int32_t)(col_per * f) : -1;
            wg += d.wg_rows;
            desc[cls].push_back(d);
        }
    }
    std::vector<FieldDesc> all_desc;
    size_t desc_at[3];
    int n_launch = 0;
    for (int cls = 0; cls < 3; cls++)
    {
        desc_at[cls] = all_desc.size();
        all_desc.insert(all_desc.end(), desc[cls].begin(), desc[cls].end());
        n_launch += desc[cls].empty() ? 0 : 1;
    }
    if (int rc = ensure(c, c->look_pairs, all_desc.size() * sizeof(FieldDesc))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->look_pairs.p, all_desc.data(), all_desc.size() * sizeof(FieldDesc), hipMemcpyHostToDevice, c->stream));
    // the co-located motion: one grid of 16x16 blocks per field (the caller's array may die after the call)
    if (col)
    {
        if (int rc = ensure(c, c->field_col, col_per * n_fields * sizeof(x265amd_me_result))) return rc;
        HIPCHK(c, hipMemcpyAsync(c->field_col.p, col, col_per * n_fields * sizeof(x265amd_me_result), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));      // the descriptors die here
    HIPCHK(c, hipMemsetD32Async((hipDeviceptr_t)c->field_results.p, LOOK_UNSET, n * 2, c->stream));   // "not searched yet"
    if (int rc = record_start(c)) return rc;
    if (c->timing)
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    const uint16_t *cost = c->d_cost + (size_t)qp * 65536 + 32768;
    const int n_streams = n_launch;
    if (n_streams > 1)
    {
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int i = 0; i + 1 < n_streams; i++)
            HIPCHK(c, hipStreamWaitEvent(c->aux[i], c->ev_fork, 0));
    }
    // Wavefronts per workgroup of the two candidate-parallel classes: eight price a round of eight candidates in the time of one,
    // but a call with many PU rows would no longer fit the GPU at once (rows dispatched late start their chains late): four then
    // (MI355X, 1080p, one session: one 16x16 field 4.36 ms with eight against 4.80 with four, one 32x32 field 3.97 against 4.56; eight
    // 16x16 fields in one call, 536 workgroups, 5.27 either way; sixteen, 1072 workgroups, 9.7 against 5.8).
    int wgs_total = 0;
    for (int cls = 0; cls < 3; cls++)
        for (const FieldDesc &d : desc[cls])
            wgs_total += d.wg_rows;
    int cp_waves = wgs_total <= 600 ? 8 : 4;
    if (const char *ev = getenv("X265AMD_CP_WAVES"))      // (tuning aid: 1 .. 8)
        cp_waves = std::min(8, std::max(1, atoi(ev)));
    // (the class of the smallest PUs has the longest chains: it takes the main stream and starts first)
    int used = 0;
    for (int cls = 0; cls < 3; cls++)
    {
        if (desc[cls].empty())
            continue;
        hipStream_t st = used == 0 ? c->stream : c->aux[used - 1];
        used++;
        const FieldDesc &last = desc[cls].back();
        const int n_wgs = last.first_wg + last.wg_rows;
        const FieldDesc *dd = (const FieldDesc*)c->look_pairs.p + desc_at[cls];
        x265amd_me_result *res = (x265amd_me_result*)c->field_results.p;
        int *mvp = (int*)c->field_mvp.p;
        int *err = (int*)c->look_progress.p + ERR_WORD_FIELD;
        const x265amd_me_result *dcol = col ? (const x265amd_me_result*)c->field_col.p : nullptr;
        if (e.bpp == 1)
            launch_field<uint8_t>(c, st, cls, n_wgs, cp_waves, cost, dd, (int)desc[cls].size(), e.width, e.height, merange, method, subme, qp, dcol, res, mvp, err);
        else
            launch_field<uint16_t>(c, st, cls, n_wgs, cp_waves, cost, dd, (int)desc[cls].size(), e.width, e.height, merange, method, subme, qp, dcol, res, mvp, err);
        HIPCHK(c, hipGetLastError());
    }
    for (int i = 0; i + 1 < n_streams; i++)
    {
        HIPCHK(c, hipEventRecord(c->ev_join[i], c->aux[i]));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[i], 0));
    }
    if (c->timing)
    {
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
        c->timed_any = true;
    }
    if (!out)
        return 0;
    int h_err = 0;
    HIPCHK(c, hipMemcpyAsync(out, c->field_results.p, n * sizeof(x265amd_me_result), hipMemcpyDeviceToHost, c->stream));
    if (out_mvp)
        HIPCHK(c, hipMemcpyAsync(out_mvp, c->field_mvp.p, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&h_err, (int*)c->look_progress.p + ERR_WORD_FIELD, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_err)
    {
        HIPCHK(c, hipMemsetAsync((int*)c->look_progress.p + ERR_WORD_FIELD, 0, sizeof(int), c->stream));
        return fail(c, X265AMD_ERR_HIP, "field search: a PU row waited too long for the row above it");
    }
    return 0;
}

int x265amd_field_me_batch(x265amd_ctx *c, int n_fields, const int *enc_planes, const int *ref_planes, int pu_w, int pu_h, int merange,
                           int method, int subme, int qp, x265amd_me_result *out, int16_t *out_mvp)
{
    if (n_fields < 1 || n_fields > 4096)
        return X265AMD_ERR_INVALID;
    const std::vector<int> w(n_fields, pu_w), h(n_fields, pu_h);
    return x265amd_field_me_multi(c, n_fields, enc_planes, ref_planes, w.data(), h.data(), nullptr, merange, method, subme, qp, out, out_mvp);
}

int x265amd_field_me(x265amd_ctx *c, int enc_plane, int ref_plane, int pu_w, int pu_h, int merange, int method, int subme, int qp,
                     x265amd_me_result *out, int16_t *out_mvp)
{
    return x265amd_field_me_batch(c, 1, &enc_plane, &ref_plane, pu_w, pu_h, merange, method, subme, qp, out, out_mvp);
}

const x265amd_me_result *x265amd_field_results_device(const x265amd_ctx *c)
{
    return c ? (const x265amd_me_result*)c->field_results.p : nullptr;
}

int x265amd_lookahead_status(x265amd_ctx *c)
{
    if (!c)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->look_progress.p)
        return 0;                       // no lookahead search was ever queued
    // the kernels OR their failures into one word per kind; they are cleared only here (and by a synchronous call that
    // reports them itself), so an asynchronous call's failure survives later calls until it is asked for
    int h_err[2] = { 0, 0 };
    HIPCHK(c, hipMemcpyAsync(h_err, c->look_progress.p, sizeof(h_err), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemsetAsync(c->look_progress.p, 0, sizeof(h_err), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_err[ERR_WORD_LOOKAHEAD] && h_err[ERR_WORD_FIELD])
        return fail(c, X265AMD_ERR_HIP, "lookahead search: a CU row waited too long for the row below it; field search: a PU row waited too long for the row above it");
    if (h_err[ERR_WORD_LOOKAHEAD])
        return fail(c, X265AMD_ERR_HIP, "lookahead search: a CU row waited too long for the row below it");
    if (h_err[ERR_WORD_FIELD])
        return fail(c, X265AMD_ERR_HIP, "field search: a PU row waited too long for the row above it");
    return 0;
}

int x265amd_ipfilter(x265amd_ctx *c, int kind, const void *src, intptr_t src_stride, void *dst, intptr_t dst_stride,
                     int w, int h, int frac, int bpp, int depth)
{
    if (!c || !src || !dst)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (kind < 0 || kind > 3 || w <= 0 || h <= 0 || w > 4096 || h > 4096 || frac < 0 || frac > 3 ||
        (bpp != 1 && bpp != 2) || depth < 8 || depth > 8 * bpp || depth > 14)
        return fail(c, X265AMD_ERR_INVALID, "bad filter arguments");
    // stage the block with the 8-tap apron (3 before, 4 behind) in the filtered direction only - the
    // reference reads no more - plus zeroed slack for the 4-wide groups' loads
    const bool horiz = kind == 0 || kind == 2;
    const int se = kind == 3 ? 2 : bpp, de = kind == 2 ? 2 : bpp;          // bytes per source / output element
    const int ax0 = horiz ? 3 : 0, ax1 = horiz ? 4 : 0, ay0 = horiz ? 0 : 3, ay1 = horiz ? 0 : 4;
    const int sw = ax0 + w + ax1, sh = ay0 + h + ay1, pitch = ((sw + 3) & ~3) + 16;
    std::vector<char> hs((size_t)pitch * (sh + 1) * se, 0);
    for (int y = 0; y < sh; y++)
        memcpy(&hs[((size_t)y * pitch + 4 - ax0) * se],
               (const char*)src + ((intptr_t)(y - ay0) * src_stride - ax0) * se, (size_t)sw * se);
    const size_t out_bytes = (size_t)w * h * de;
    if (int rc = ensure(c, c->stage, hs.size() + 64)) return rc;
    if (int rc = ensure(c, c->costs, out_bytes)) return rc;
    HIPCHK(c, hipMemcpyAsync(c->stage.p, hs.data(), hs.size(), hipMemcpyHostToDevice, c->stream));
    const char *origin = (const char*)c->stage.p + ((size_t)ay0 * pitch + 4) * se;     // element (0,0): column 4 of row ay0
    const dim3 grid((((w + 3) >> 2) * h + 255) / 256), block(256);
    if (bpp == 1)
        hipLaunchKernelGGL(k_ipfilter<uint8_t>, grid, block, 0, c->stream, kind, (const void*)origin, pitch, c->costs.p, w, h, frac, depth);
    else
        hipLaunchKernelGGL(k_ipfilter<uint16_t>, grid, block, 0, c->stream, kind, (const void*)origin, pitch, c->costs.p, w, h, frac, depth);
    HIPCHK(c, hipGetLastError());
    std::vector<char> ho(out_bytes);
    HIPCHK(c, hipMemcpyAsync(ho.data(), c->costs.p, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int y = 0; y < h; y++)
        memcpy((char*)dst + (intptr_t)y * dst_stride * de, &ho[(size_t)y * w * de], (size_t)w * de);
    return 0;
}

int x265amd_cost_batch(x265amd_ctx *c, int kind, int w, int h, int n, const x265amd_block_pair *pairs, int32_t *out)
{
    if (!c || !pairs || !out || n < 0)
        return X265AMD_ERR_INVALID;
    if (n == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (kind < X265AMD_SAD || kind > X265AMD_SA8D || !valid_dim(w) || !valid_dim(h))
        return fail(c, X265AMD_ERR_INVALID, "bad cost kind or block size");
    int bpp = 0;
    for (int i = 0; i < n; i++)
    {
        const x265amd_block_pair &p = pairs[i];
        if (p.enc_plane < 0 || p.enc_plane >= MAX_PLANES || !c->used[p.enc_plane] ||
            p.ref_plane < 0 || p.ref_plane >= MAX_PLANES || !c->used[p.ref_plane])
            return fail(c, X265AMD_ERR_INVALID, "block pair names an unregistered plane");
        const PlaneDesc &e = c->h_planes[p.enc_plane], &r = c->h_planes[p.ref_plane];
        if (e.bpp != r.bpp || (bpp && e.bpp != bpp))
            return fail(c, X265AMD_ERR_INVALID, "mixed pixel sizes in one cost batch");
        bpp = e.bpp;
        // blocks must lie inside the padded planes
        if (p.enc_x < -e.margin_x || p.enc_y < -e.margin_y || p.enc_x + w > e.width + e.margin_x || p.enc_y + h > e.height + e.margin_y ||
            p.ref_x < -r.margin_x || p.ref_y < -r.margin_y || p.ref_x + w > r.width + r.margin_x || p.ref_y + h > r.height + r.margin_y)
            return fail(c, X265AMD_ERR_RANGE, "block outside the padded plane");
    }
    if (int rc = sync_planes(c)) return rc;
    if (int rc = ensure(c, c->pairs, (size_t)n * sizeof(x265amd_block_pair))) return rc;
    if (int rc = ensure(c, c->costs, (size_t)n * sizeof(int32_t))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pairs.p, pairs, (size_t)n * sizeof(x265amd_block_pair), hipMemcpyHostToDevice, c->stream));
    if (bpp == 1)
        hipLaunchKernelGGL(k_cost<uint8_t>, dim3(n), dim3(64), 0, c->stream, c->d_planes, kind, w, h,
                           (const x265amd_block_pair*)c->pairs.p, n, (int32_t*)c->costs.p);
    else
        hipLaunchKernelGGL(k_cost<uint16_t>, dim3(n), dim3(64), 0, c->stream, c->d_planes, kind, w, h,
                           (const x265amd_block_pair*)c->pairs.p, n, (int32_t*)c->costs.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->costs.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int x265amd_mc_cost_batch(x265amd_ctx *c, int kind, int w, int h, int n, const x265amd_mc_pair *pairs, int32_t *out)
{
    if (!c || !pairs || !out || n < 0)
        return X265AMD_ERR_INVALID;
    if (n == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (kind < X265AMD_SAD || kind > X265AMD_SA8D || !valid_dim(w) || !valid_dim(h))
        return fail(c, X265AMD_ERR_INVALID, "bad cost kind or block size");
    int bpp = 0;
    for (int i = 0; i < n; i++)
    {
        const x265amd_mc_pair &p = pairs[i];
        if (p.enc_plane < 0 || p.enc_plane >= MAX_PLANES || !c->used[p.enc_plane] ||
            p.ref_plane < 0 || p.ref_plane >= MAX_PLANES || !c->used[p.ref_plane])
            return fail(c, X265AMD_ERR_INVALID, "pair names an unregistered plane");
        const PlaneDesc &e = c->h_planes[p.enc_plane], &r = c->h_planes[p.ref_plane];
        if (e.bpp != r.bpp || (bpp && e.bpp != bpp) || r.lr_step)
            return fail(c, X265AMD_ERR_INVALID, "mixed pixel sizes in one batch, or a half-resolution reference");
        bpp = e.bpp;
        if (p.qmv_x < -32768 || p.qmv_x > 32767 || p.qmv_y < -32768 || p.qmv_y > 32767)
            return fail(c, X265AMD_ERR_INVALID, "MV does not fit 16 bits");
        const int rx = p.pu_x + (p.qmv_x >> 2), ry = p.pu_y + (p.qmv_y >> 2);
        if (p.pu_x < -e.margin_x || p.pu_y < -e.margin_y || p.pu_x + w > e.width + e.margin_x || p.pu_y + h > e.height + e.margin_y ||
            rx - 3 < -r.margin_x || ry - 3 < -r.margin_y || rx + w + 4 > r.width + r.margin_x || ry + h + 4 > r.height + r.margin_y)
            return fail(c, X265AMD_ERR_RANGE, "block (with the filter apron) outside the padded plane");
    }
    if (int rc = sync_planes(c)) return rc;
    if (int rc = ensure(c, c->pairs, (size_t)n * sizeof(x265amd_mc_pair))) return rc;
    if (int rc = ensure(c, c->costs, (size_t)n * sizeof(int32_t))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pairs.p, pairs, (size_t)n * sizeof(x265amd_mc_pair), hipMemcpyHostToDevice, c->stream));
    if (bpp == 1)
        hipLaunchKernelGGL(k_mc_cost<uint8_t>, dim3(n), dim3(64), 0, c->stream, c->d_planes, kind, w, h,
                           (const x265amd_mc_pair*)c->pairs.p, n, (int32_t*)c->costs.p);
    else
        hipLaunchKernelGGL(k_mc_cost<uint16_t>, dim3(n), dim3(64), 0, c->stream, c->d_planes, kind, w, h,
                           (const x265amd_mc_pair*)c->pairs.p, n, (int32_t*)c->costs.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->costs.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int x265amd_bipred_source_batch(x265amd_ctx *c, int dst_plane, int w, int h, int n, const x265amd_mc_pair *pairs)
{
    if (!c || !pairs || n < 0)
        return X265AMD_ERR_INVALID;
    if (n == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!valid_dim(w) || !valid_dim(h) || dst_plane < 0 || dst_plane >= MAX_PLANES || !c->used[dst_plane] || c->h_planes[dst_plane].lr_step)
        return fail(c, X265AMD_ERR_INVALID, "bad block size or destination plane");
    const PlaneDesc &d = c->h_planes[dst_plane];
    for (int i = 0; i < n; i++)
    {
        const x265amd_mc_pair &p = pairs[i];
        if (p.enc_plane < 0 || p.enc_plane >= MAX_PLANES || !c->used[p.enc_plane] ||
            p.ref_plane < 0 || p.ref_plane >= MAX_PLANES || !c->used[p.ref_plane])
            return fail(c, X265AMD_ERR_INVALID, "pair names an unregistered plane");
        const PlaneDesc &e = c->h_planes[p.enc_plane], &r = c->h_planes[p.ref_plane];
        if (e.bpp != r.bpp || e.bpp != d.bpp || r.lr_step || e.lr_step)
            return fail(c, X265AMD_ERR_INVALID, "mixed pixel sizes, or a half-resolution plane");
        if (p.qmv_x < -32768 || p.qmv_x > 32767 || p.qmv_y < -32768 || p.qmv_y > 32767)
            return fail(c, X265AMD_ERR_INVALID, "MV does not fit 16 bits");
        const int rx = p.pu_x + (p.qmv_x >> 2), ry = p.pu_y + (p.qmv_y >> 2);
        if (p.pu_x < -e.margin_x || p.pu_y < -e.margin_y || p.pu_x + w > e.width + e.margin_x || p.pu_y + h > e.height + e.margin_y ||
            p.pu_x < -d.margin_x || p.pu_y < -d.margin_y || p.pu_x + w > d.width + d.margin_x || p.pu_y + h > d.height + d.margin_y ||
            rx - 3 < -r.margin_x || ry - 3 < -r.margin_y || rx + w + 4 > r.width + r.margin_x || ry + h + 4 > r.height + r.margin_y)
            return fail(c, X265AMD_ERR_RANGE, "block (with the filter apron) outside a padded plane");
    }
    if (int rc = sync_planes(c)) return rc;
    if (int rc = ensure(c, c->pairs, (size_t)n * sizeof(x265amd_mc_pair))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pairs.p, pairs, (size_t)n * sizeof(x265amd_mc_pair), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // (pairs is borrowed only during the call)
    const long long threads = (long long)n * (w >> 2) * h;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    if (d.bpp == 1)
        hipLaunchKernelGGL(k_bipred_src<uint8_t>, grid, block, 0, c->stream, c->d_planes, dst_plane, w, h, (const x265amd_mc_pair*)c->pairs.p, n);
    else
        hipLaunchKernelGGL(k_bipred_src<uint16_t>, grid, block, 0, c->stream, c->d_planes, dst_plane, w, h, (const x265amd_mc_pair*)c->pairs.p, n);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int validate_jobs(x265amd_ctx *c, const x265amd_me_job *jobs, int n, int n_refs, bool *any_full)
{
    *any_full = false;
    for (int i = 0; i < n; i++)
    {
        const x265amd_me_job &j = jobs[i];
        if (!valid_dim(j.w) || !valid_dim(j.h))
            return fail(c, X265AMD_ERR_INVALID, "job size is not a partition size");
        if (j.method == X265AMD_UMH && j.w == 4 && j.h == 4)
            return fail(c, X265AMD_ERR_INVALID, "UMH does not support 4x4 (the reference asserts, encoder/motion.cpp:447)");
        if (j.method == X265AMD_BIPRED && (j.icost != X265AMD_SAD || j.mvmin_x > j.mvmax_x || j.mvmin_y > j.mvmax_y))
            return fail(c, X265AMD_ERR_INVALID, "a bidirectional-refinement job needs a non-empty range and the SAD metric");
        if (!job_fields_valid(j, n_refs))       // (the same predicate the device-array entry points apply per job)
            return fail(c, X265AMD_ERR_INVALID, "job field out of range");
        if (j.method == X265AMD_FULL || j.method == X265AMD_BIPRED)
            *any_full = true;
    }
    return 0;
}

// Every block a job can evaluate must lie inside the padded planes, as in the reference: the
// search bounds, plus the 3 pixels a pattern may step beyond them before its range check, plus the
// 8-tap filter's apron of 3 pixels before and 4 behind the block (encoder/motion.cpp:364,407,625;
// common/ipfilter.cpp).  (The kernels clamp addresses, so a violation would not fault: it would
// silently measure other pixels than the reference.)
static int check_job_ranges(x265amd_ctx *c, int enc_plane, const int *ref_planes, const x265amd_me_job *jobs, int n)
{
    const PlaneDesc &e = c->h_planes[enc_plane];
    for (int i = 0; i < n; i++)
    {
        const x265amd_me_job &j = jobs[i];
        const PlaneDesc &r = c->h_planes[ref_planes[j.ref_idx]];
        if (j.pu_x < -e.margin_x || j.pu_y < -e.margin_y || j.pu_x + j.w > e.width + e.margin_x || j.pu_y + j.h > e.height + e.margin_y)
            return fail(c, X265AMD_ERR_RANGE, "PU outside the padded encoder plane");
        if (j.mvmin_x > j.mvmax_x || j.mvmin_y > j.mvmax_y)
            continue;           // empty range: only the clipped predictors are measured, at the clamped bounds
        if (j.pu_x + j.mvmin_x - 6 < -r.margin_x || j.pu_y + j.mvmin_y - 6 < -r.margin_y ||
            j.pu_x + j.w + j.mvmax_x + 7 > r.width + r.margin_x || j.pu_y + j.h + j.mvmax_y + 7 > r.height + r.margin_y)
            return fail(c, X265AMD_ERR_RANGE, "a job's search window reaches outside the reference plane's margins");
    }
    return 0;
}

int x265amd_me_search_batch(x265amd_ctx *c, int enc_plane, const int *ref_planes, int n_refs,
                            const x265amd_me_job *jobs, int n, x265amd_me_result *out)
{
    if (!c || !jobs || !out || n < 0)
        return X265AMD_ERR_INVALID;
    if (n == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_batch_planes(c, enc_plane, ref_planes, n_refs)) return rc;
    if (int rc = wait_plane_fills(c, enc_plane, ref_planes, n_refs)) return rc;
    bool any_full = false;
    if (int rc = validate_jobs(c, jobs, n, n_refs, &any_full)) return rc;
    if (int rc = check_job_ranges(c, enc_plane, ref_planes, jobs, n)) return rc;
    if (int rc = sync_planes(c)) return rc;
    if (!c->batch_list)
    {
        c->batch_list = new x265amd_joblist;
        c->batch_list->keep_hist = false;
    }
    x265amd_joblist *L = c->batch_list;
    int rc = joblist_build(c, L, jobs, n, c->h_planes[enc_plane].bpp);
    if (!rc)
    {
        BatchParams P = make_params(c, enc_plane, ref_planes, n_refs);
        rc = joblist_launch(c, L, P);
    }
    if (!rc)
    {
        hipError_t e = hipMemcpyAsync(out, L->results.p, (size_t)n * sizeof(x265amd_me_result), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess)
            e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess)
            rc = fail(c, X265AMD_ERR_HIP, "result download", e);
    }
    return rc;
}

int x265amd_me_search_batch_device(x265amd_ctx *c, int enc_plane, const int *ref_planes, int n_refs,
                                   const x265amd_me_job *d_jobs, int n, x265amd_me_result *d_out)
{
    if (!c || !d_jobs || !d_out || n < 0)
        return X265AMD_ERR_INVALID;
    if (n == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (int rc = check_batch_planes(c, enc_plane, ref_planes, n_refs)) return rc;
    if (int rc = wait_plane_fills(c, enc_plane, ref_planes, n_refs)) return rc;
    if (int rc = sync_planes(c)) return rc;
    BatchParams P = make_params(c, enc_plane, ref_planes, n_refs);
    if (int rc = ensure(c, c->fullbest, (size_t)n * sizeof(FullBest))) return rc;
    if (int rc = ensure(c, c->items, (size_t)n * sizeof(FullItem))) return rc;
    if (int rc = ensure(c, c->item_jobs, (size_t)n * sizeof(int))) return rc;
    const PlaneDesc &e = c->h_planes[enc_plane];
    const int cells_x = (e.width + 63) / 64, cells_y = (e.height + 63) / 64;
    unsigned char *plan_mem = nullptr;
    if (e.bpp == 1 && c->hier_ok && (int64_t)cells_x * cells_y * n_refs <= (1 << 20))
    {
        if (int rc = ensure(c, c->plan_mem, plan_bytes(cells_x * cells_y * n_refs, n))) return rc;
        plan_mem = (unsigned char*)c->plan_mem.p;
    }
    return launch_search_unplanned(c, P, e.bpp, d_jobs, n, d_out,
                                   (FullBest*)c->fullbest.p, (FullItem*)c->items.p, (int*)c->item_jobs.p, plan_mem, cells_x, cells_y);
}

int x265amd_me_search_batch_device_noalloc(x265amd_ctx *c, int enc_plane, const int *ref_planes, int n_refs,
                                           const x265amd_me_job *d_jobs, int n, x265amd_me_result *d_out)
{
    if (!c || !d_jobs || !d_out || n < 0)
        return X265AMD_ERR_INVALID;
    if (n == 0)
        return 0;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (n > NOALLOC_MAX_JOBS)
    {
        char msg[160];
        snprintf(msg, sizeof(msg), "%d jobs exceed the %d the allocation-free entry point has scratch for; split the batch", n, NOALLOC_MAX_JOBS);
        return fail(c, X265AMD_ERR_INVALID, msg);
    }
    if (int rc = check_batch_planes(c, enc_plane, ref_planes, n_refs)) return rc;
    if (int rc = wait_plane_fills(c, enc_plane, ref_planes, n_refs)) return rc;
    if (int rc = sync_planes(c)) return rc;          // (a copy from pinned memory; nothing is allocated)
    // every context on this GPU shares the array: order this call behind the last one that used it
    std::lock_guard<std::mutex> gs(g_scratch_mu);
    unsigned char *sp = (unsigned char*)g_scratch_ptr[c->device];
    if (!sp)
        return fail(c, X265AMD_ERR_HIP, "the scratch array's device address is unknown");
    HIPCHK(c, hipStreamWaitEvent(c->stream, g_scratch_ev[c->device], 0));
    BatchParams P = make_params(c, enc_plane, ref_planes, n_refs);
    const PlaneDesc &e = c->h_planes[enc_plane];
    const int cells_x = (e.width + 63) / 64, cells_y = (e.height + 63) / 64;
    const bool plan = (int64_t)cells_x * cells_y * n_refs <= NOALLOC_MAX_CELLS && plan_bytes(cells_x * cells_y * n_refs, n) <= NOALLOC_PLAN_BYTES;
    const int rc = launch_search_unplanned(c, P, e.bpp, d_jobs, n, d_out,
                                           (FullBest*)(sp + NOALLOC_FB_OFF), (FullItem*)(sp + NOALLOC_ITEMS_OFF),
                                           (int*)(sp + NOALLOC_IJOBS_OFF), plan ? sp + NOALLOC_PLAN_OFF : nullptr, cells_x, cells_y);
    HIPCHK(c, hipEventRecord(g_scratch_ev[c->device], c->stream));
    return rc;
}

int x265amd_joblist_create(x265amd_ctx *c, const x265amd_me_job *jobs, int n, int n_refs, int bytes_per_px,
                           x265amd_joblist **out)
{
    if (!c || !jobs || !out || n <= 0 || (bytes_per_px != 1 && bytes_per_px != 2))
        return X265AMD_ERR_INVALID;
    *out = nullptr;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    bool any_full = false;
    if (int rc = validate_jobs(c, jobs, n, n_refs, &any_full)) return rc;
    x265amd_joblist *L = new x265amd_joblist;
    int rc = joblist_build(c, L, jobs, n, bytes_per_px);
    if (rc)
    {
        joblist_free(c, L);
        return rc;
    }
    L->n_refs = n_refs;
    L->h_jobs.assign(jobs, jobs + n);
    *out = L;
    return 0;
}

namespace {
// the main stream waits for the list's last run if that ran on the other stream set
int wait_list_done(x265amd_ctx *c, x265amd_joblist *L)
{
    if (L->last_set >= 0 && L->last_set != c->cur_set)
        HIPCHK(c, hipStreamWaitEvent(c->stream, L->ev_done, 0));
    return 0;
}
} // namespace

static int joblist_run_on(x265amd_ctx *c, x265amd_joblist *L, int enc_plane, const int *ref_planes, int n_refs,
                          x265amd_me_result *out, int set)
{
    if (!c || !L || set < 0 || set > 7)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    StreamSet on(c, set);
    if (!L->ev_done)
        HIPCHK(c, hipEventCreateWithFlags(&L->ev_done, hipEventDisableTiming));
    if (int rc = wait_list_done(c, L)) return rc;
    if (int rc = check_batch_planes(c, enc_plane, ref_planes, n_refs)) return rc;
    if (int rc = wait_plane_fills(c, enc_plane, ref_planes, n_refs)) return rc;
    if (c->h_planes[enc_plane].bpp != L->bpp)
        return fail(c, X265AMD_ERR_INVALID, "job list was planned for a different pixel size");
    if (n_refs < L->n_refs)
        return fail(c, X265AMD_ERR_INVALID, "job list was created for more reference planes than this run supplies");
    {
        // the jobs' windows against these planes' margins: checked once per plane geometry
        std::vector<int64_t> geom;
        auto push = [&](const PlaneDesc &d) {
            for (int64_t v : { (int64_t)d.width, (int64_t)d.height, (int64_t)d.margin_x, (int64_t)d.margin_y })
                geom.push_back(v);
        };
        push(c->h_planes[enc_plane]);
        for (int i = 0; i < L->n_refs; i++)
            push(c->h_planes[ref_planes[i]]);
        if (geom != L->checked_geom)
        {
            if (int rc = check_job_ranges(c, enc_plane, ref_planes, L->h_jobs.data(), L->n)) return rc;
            L->checked_geom.swap(geom);
        }
    }
    if (int rc = sync_planes(c)) return rc;
    BatchParams P = make_params(c, enc_plane, ref_planes, n_refs);
    if (int rc = joblist_launch(c, L, P)) return rc;
    HIPCHK(c, hipEventRecord(L->ev_done, c->stream));
    L->last_set = set;
    if (set != 0 && set != 4)
        for (int k = -1; k < n_refs; k++)
        {
            const int id = k < 0 ? enc_plane : ref_planes[k];
            if (!c->plane_read[id][set])
                HIPCHK(c, hipEventCreateWithFlags(&c->plane_read[id][set], hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(c->plane_read[id][set], c->stream));
            c->read_pending[id] |= (unsigned char)(1u << set);
        }
    if (out)
    {
        HIPCHK(c, hipMemcpyAsync(out, L->results.p, (size_t)L->n * sizeof(x265amd_me_result), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

int x265amd_joblist_run(x265amd_ctx *c, x265amd_joblist *L, int enc_plane, const int *ref_planes, int n_refs,
                        x265amd_me_result *out)
{
    return joblist_run_on(c, L, enc_plane, ref_planes, n_refs, out, 0);
}

int x265amd_joblist_run_overlapped(x265amd_ctx *c, x265amd_joblist *L, int enc_plane, const int *ref_planes, int n_refs, int set)
{
    return joblist_run_on(c, L, enc_plane, ref_planes, n_refs, nullptr, set);
}

const x265amd_me_result *x265amd_joblist_results_device(const x265amd_joblist *L)
{
    return L ? (const x265amd_me_result*)L->results.p : nullptr;
}

int x265amd_joblist_results(x265amd_ctx *c, x265amd_joblist *L, x265amd_me_result *out)
{
    if (!c || !L || !out)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!L->ran)
        return fail(c, X265AMD_ERR_INVALID, "the job list has not run yet");
    if (int rc = wait_list_done(c, L)) return rc;
    HIPCHK(c, hipMemcpyAsync(out, L->results.p, (size_t)L->n * sizeof(x265amd_me_result), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

namespace {
// reorder every class of a list: 8 groups by predicted length, longest first; inside a group the planned (picture / XCD
// band) order stays
int order_by_length(x265amd_ctx *c, x265amd_joblist *L, const std::vector<unsigned> &len)
{
    int off = 0;
    for (int cls = 0; cls < JOB_CLASSES; cls++)
    {
        const int cnt = L->cls_count[cls];
        if (cnt > 1)
        {
            unsigned mx = 1;
            for (int i = 0; i < cnt; i++)
                mx = std::max(mx, len[L->order[off + i]]);
            auto group = [&](int id) { return 7 - (int)((uint64_t)len[id] * 8 / ((uint64_t)mx + 1)); };
            std::stable_sort(L->order.begin() + off, L->order.begin() + off + cnt,
                             [&](int a, int b) { return group(a) < group(b); });
        }
        off += cnt;
    }
    HIPCHK(c, hipMemcpy(L->idx.p, L->order.data(), (size_t)L->n * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}
} // namespace

int x265amd_joblist_eval_counts(x265amd_ctx *c, x265amd_joblist *L, uint32_t *out)
{
    if (!c || !L || !out)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!L->ran || !L->keep_hist)
        return fail(c, X265AMD_ERR_INVALID, "the job list has not run yet");
    if (int rc = wait_list_done(c, L)) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, L->hist.p, (size_t)L->n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

int x265amd_joblist_order_by_length(x265amd_ctx *c, x265amd_joblist *L, const uint32_t *length)
{
    if (!c || !L || !length)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // the index list may be in use
    return order_by_length(c, L, std::vector<unsigned>(length, length + L->n));
}

int x265amd_joblist_order_by_field(x265amd_ctx *c, x265amd_joblist *L, const x265amd_me_result *field, int cu_w, int cu_h)
{
    if (!c || !L || !field || cu_w <= 0 || cu_h <= 0)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (L->h_jobs.empty())
        return fail(c, X265AMD_ERR_INVALID, "the list keeps no host copy of its jobs");
    HIPCHK(c, hipStreamSynchronize(c->stream));          // the index list may be in use
    // a pattern search walks from its predictor to where the block went: its length grows with the distance between the
    // two.  The half-resolution field says where the block went (its MVs are q-pel of the half-resolution picture: x 2).
    std::vector<unsigned> len(L->n);
    for (int i = 0; i < L->n; i++)
    {
        const x265amd_me_job &j = L->h_jobs[i];
        const int cx = std::min(cu_w - 1, std::max(0, (j.pu_x + j.w / 2) >> 4)), cy = std::min(cu_h - 1, std::max(0, (j.pu_y + j.h / 2) >> 4));
        const x265amd_me_result &f = field[cx + (size_t)cy * cu_w];
        const int dx = 2 * f.mvx - j.qmvp_x, dy = 2 * f.mvy - j.qmvp_y;
        len[i] = (unsigned)(std::abs(dx) + std::abs(dy));
    }
    return order_by_length(c, L, len);
}

int x265amd_joblist_results_async(x265amd_ctx *c, x265amd_joblist *L, x265amd_me_result *out)
{
    if (!c || !L || !out)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!L->ran)
        return fail(c, X265AMD_ERR_INVALID, "the job list has not run yet");
    // (queued on the stream set the list last ran on, right behind that run: the other sets' frames are not held up)
    StreamSet on(c, L->last_set > 0 ? L->last_set : 0);
    if (int rc = wait_list_done(c, L)) return rc;
    // Page-locked, device-mapped host memory is written by a small kernel behind the run on the main stream (a few
    // microseconds; queueing a DMA copy costs the host about 0.1 ms per call on this platform, and one more stream
    // would share a hardware queue with the caller's upload stream).  Anything else: hipMemcpyAsync on a stream of
    // its own, so that the batches queued next do not wait for it.
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, out, 0) == hipSuccess && mapped)
    {
        static_assert(sizeof(x265amd_me_result) == 8, "one result = one 64-bit word");
        hipLaunchKernelGGL(k_copy_words, dim3((L->n + 255) / 256), dim3(256), 0, c->stream,
                           (const unsigned long long*)L->results.p, (unsigned long long*)mapped, L->n);
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    (void)hipGetLastError();
    if (!c->out_stream)
        HIPCHK(c, hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
    if (!L->ev_ran)
    {
        HIPCHK(c, hipEventCreateWithFlags(&L->ev_ran, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&L->ev_out, hipEventDisableTiming));
    }
    HIPCHK(c, hipEventRecord(L->ev_ran, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->out_stream, L->ev_ran, 0));
    HIPCHK(c, hipMemcpyAsync(out, L->results.p, (size_t)L->n * sizeof(x265amd_me_result), hipMemcpyDeviceToHost, c->out_stream));
    HIPCHK(c, hipEventRecord(L->ev_out, c->out_stream));
    L->out_pending = true;
    return 0;
}

#ifndef XA_SRC_HASH
#define XA_SRC_HASH "unknown"
#endif
const char *x265amd_build_info(void) { return "x265_amd " XA_SRC_HASH; }

// Launch order by job length.  A pattern search's length is data dependent (a far-moving block takes
// three times the evaluations of a still one) and the general kernel's launch ends with its longest
// jobs; starting those first shortens the launch.  The lengths are those of the list's last run -
// consecutive frames of a sequence move alike - grouped coarsely so that jobs of similar length keep
// their picture order and XCD banding.
int x265amd_joblist_rebalance(x265amd_ctx *c, x265amd_joblist *L)
{
    if (!c || !L)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!L->ran || L->n == 0)
        return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned> hist(L->n);
    HIPCHK(c, hipMemcpy(hist.data(), L->hist.p, (size_t)L->n * sizeof(unsigned), hipMemcpyDeviceToHost));
    return order_by_length(c, L, hist);
}

int x265amd_joblist_destroy(x265amd_ctx *c, x265amd_joblist *L)
{
    if (!c || !L)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    joblist_free(c, L);
    return 0;
}

int x265amd_eval_count(x265amd_ctx *c, int reset, uint64_t *evals)
{
    if (!c)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<unsigned long long> t(EVAL_WORDS);
    HIPCHK(c, hipMemcpy(t.data(), c->d_evals, EVAL_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long v = 0;
    for (int i = 0; i < EVAL_SLOTS; i++)
        v += t[(size_t)i * 16];
    if (evals)
        *evals = v;
#ifdef XA_PROFILE
    {
        const unsigned long long *q = &t[(size_t)EVAL_SLOTS * 16];
        fprintf(stderr, "phase clocks: bind %llu  mvp %llu  predictors %llu  pattern %llu  subpel %llu  publish %llu\n",
                q[0], q[1], q[2], q[3], q[4], q[5]);
        if (q[15])
            fprintf(stderr, "lookahead rows (sampled), clocks: cache PU %llu  wait for the row below %llu  mvp %llu  predictors %llu  pattern %llu  subpel %llu  publish %llu  | whole CU loop %llu\n",
                    q[8], q[9], q[10], q[11], q[12], q[13], q[14], q[15]);
        const unsigned long long *cq = &t[(size_t)(EVAL_SLOTS + 1) * 16];
        const unsigned long long *sq = &t[(size_t)(EVAL_SLOTS + 2) * 16];
        if (sq[7])
            fprintf(stderr, "chained field (sampled rows): sum of (clocks / 64)^2 per phase: cache %llu  wait %llu  mvp %llu  predictors %llu  pattern %llu  subpel %llu  "
                            "publish %llu  work (loop - wait) %llu; largest per PU: cache %llu  wait %llu  mvp %llu  predictors %llu  pattern %llu  subpel %llu  publish %llu  work %llu\n",
                    sq[0], sq[1], sq[2], sq[3], sq[4], sq[5], sq[6], sq[7], sq[8], sq[9], sq[10], sq[11], sq[12], sq[13], sq[14], sq[15]);
        if (cq[0])
            fprintf(stderr, "half-resolution sub-pel calls (sampled rows): %llu calls; clocks per call: set-up %llu  call -> entry %llu  MV costs %llu  "
                            "plane reads %llu  reduce %llu  return %llu  unpack %llu\n", cq[0], cq[1] / cq[0], cq[2] / cq[0], cq[3] / cq[0],
                    cq[4] / cq[0], cq[5] / cq[0], cq[6] / cq[0], cq[7] / cq[0]);
        if (cq[8] || cq[11])
            fprintf(stderr, "evaluation calls (sampled jobs): SAD %llu calls, set-up %llu + call %llu clocks each; other %llu calls, set-up %llu + call %llu clocks each\n",
                    cq[8], cq[9] / (cq[8] ? cq[8] : 1), cq[10] / (cq[8] ? cq[8] : 1), cq[11], cq[12] / (cq[11] ? cq[11] : 1), cq[13] / (cq[11] ? cq[11] : 1));
        unsigned long long fp[8], zero[8] = { 0 };
        HIPCHK(c, hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_full_prof), sizeof(fp)));
        if (fp[4])
            fprintf(stderr, "exhaustive (sampled wave 0 of %llu items): total %llu  window staging %llu  job staging %llu  scan %llu clocks per item; %.2f us per item by the 100 MHz counter -> the clock counter runs at %.0f MHz\n",
                    fp[4], fp[0] / fp[4], fp[1] / fp[4], fp[2] / fp[4], fp[3] / fp[4], fp[7] / (double)fp[4] / 100.0,
                    fp[7] ? fp[0] / (fp[7] / 100.0) : 0.0);
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_full_prof), zero, sizeof(zero)));
        HIPCHK(c, hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_hier_prof), sizeof(fp)));
        if (fp[4])
            fprintf(stderr, "shared-window pass (wave 0 of %llu items), clocks per item: total %llu  staging %llu  cell loop %llu  epilogue %llu\n",
                    fp[4], fp[0] / fp[4], fp[1] / fp[4], fp[2] / fp[4], fp[3] / fp[4]);
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_hier_prof), zero, sizeof(zero)));
        if (const char *path = getenv("X265AMD_HIER_TRACE"))        // per work item: start, staged, end
        {
            std::vector<unsigned long long> tr(3 * 16384);
            HIPCHK(c, hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_hier_trace), tr.size() * 8));
            if (FILE *fo = fopen(path, "wb"))
            {
                fwrite(tr.data(), 8, tr.size(), fo);
                fclose(fo);
            }
        }
    }
#endif
    if (reset)
        HIPCHK(c, hipMemset(c->d_evals, 0, EVAL_WORDS * sizeof(unsigned long long)));
    return 0;
}

int x265amd_set_timing(x265amd_ctx *c, int enable)
{
    if (!c)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    c->timing = enable != 0;
    c->timed_any = false;
    return 0;
}

int x265amd_kernel_times(x265amd_ctx *c, float ms[2])
{
    if (!c || !ms)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    ms[0] = ms[1] = 0.f;
    if (!c->timing || !c->timed_any)
        return fail(c, X265AMD_ERR_INVALID, "no timed batch (call x265amd_set_timing first)");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->ev[2]));
    if (c->timed_full)
        HIPCHK(c, hipEventElapsedTime(&ms[0], c->ev[0], c->ev[1]));
    HIPCHK(c, hipEventElapsedTime(&ms[1], c->ev[1], c->ev[2]));
    return 0;
}

int x265amd_mvcost_table(x265amd_ctx *c, int qp, uint16_t *out)
{
    if (!c || !out || qp < 0 || qp > 51)
        return X265AMD_ERR_INVALID;
    std::lock_guard<std::mutex> g(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    // read back what the kernels actually use
    HIPCHK(c, hipMemcpy(out, c->d_cost + (size_t)qp * 65536, 65536 * sizeof(uint16_t), hipMemcpyDeviceToHost));
    return 0;
}

// BitCost::bitcost (reference: encoder/bitcost.h:49-53)
int x265amd_bitcost(x265amd_ctx *c, int mvp_x, int mvp_y, int mv_x, int mv_y)
{
    if (!c)
        return X265AMD_ERR_INVALID;
    int ax = abs(mv_x - mvp_x), ay = abs(mv_y - mvp_y);
    if (ax > 16383 || ay > 16383)
        return X265AMD_ERR_INVALID;
    return (uint16_t)(c->bitsizes[(ax << 1) + !!(mv_x < mvp_x)] + c->bitsizes[(ay << 1) + !!(mv_y < mvp_y)] + 0.5f);
}

} // extern "C"

// -------------------------------------------------------------------------------------------
// Per-call primitives with the reference's signatures (common/primitives.h:180-184).
// A process-wide context per pixel size; each call packs its blocks (any caller stride, e.g. the
// TestBench's odd 59) into pinned staging, copies them to the device as small stride-w planes
// and runs k_cost on one pair per reference pointer.

namespace {

struct PrimCtx
{
    x265amd_ctx *ctx = nullptr;
    std::mutex   mu;
    int          bpp = 0;
};
PrimCtx g_prim[3];

// copy `rows` rows of `row_elems` pixels with stride `stride` into pinned staging at byte offset `at`
size_t stage_block(x265amd_ctx *c, size_t at, const void *src, intptr_t stride, int row_elems, int rows, int bpp)
{
    char *dst = (char*)c->h_stage + at;
    for (int y = 0; y < rows; y++)
        memcpy(dst + (size_t)y * row_elems * bpp, (const char*)src + (intptr_t)y * stride * bpp, (size_t)row_elems * bpp);
    return at + (size_t)rows * row_elems * bpp;
}

// fenc (stride fs) against nref blocks (stride rs); results to res[0..nref)
int prim_eval(int bpp, int kind, int w, int h, const void *fenc, intptr_t fs, const void *const *fref, int nref, intptr_t rs, int *res)
{
    PrimCtx &pc = g_prim[bpp];
    if (!pc.ctx)
    {
        fprintf(stderr, "x265_amd: primitive called before x265amd_setup_primitives()\n");
        abort();
    }
    std::lock_guard<std::mutex> g(pc.mu);
    x265amd_ctx *c = pc.ctx;
    hipSetDevice(c->device);
    // staged layout: [fenc w x h][ref0 w x h]...; each becomes a margin-less plane of stride w
    size_t blk = (size_t)w * h * bpp;
    size_t need = blk * (1 + nref) + 64;
    if (c->h_stage_cap < need)
    {
        if (c->h_stage) hipHostFree(c->h_stage);
        c->h_stage_cap = need < (1 << 16) ? (1 << 16) : need;
        if (hipHostMalloc(&c->h_stage, c->h_stage_cap) != hipSuccess) { fprintf(stderr, "x265_amd: pinned alloc failed\n"); abort(); }
    }
    if (ensure(c, c->stage, c->h_stage_cap)) { fprintf(stderr, "x265_amd: %s\n", c->err.c_str()); abort(); }
    size_t at = stage_block(c, 0, fenc, fs, w, h, bpp);
    for (int k = 0; k < nref; k++)
        at = stage_block(c, at, fref[k], rs, w, h, bpp);
    hipMemcpyAsync(c->stage.p, c->h_stage, at, hipMemcpyHostToDevice, c->stream);
    // temporary planes: slots at the top of the table; the valid range of each is the whole staging buffer
    int ids[5];
    x265amd_block_pair pairs[4];
    {
        std::lock_guard<std::mutex> g2(c->mu);
        for (int k = 0; k <= nref; k++)
        {
            int id = MAX_PLANES - 1 - k;
            ids[k] = id;
            PlaneDesc &d = c->h_planes[id];
            int64_t first = (int64_t)((blk * k) / bpp);
            d.origin = (const char*)c->stage.p + blk * k;
            d.lo = -first;
            d.hi = (int64_t)(c->stage.cap / bpp) - first;
            d.stride = w;
            d.width = w;
            d.height = h;
            d.bpp = bpp;
            d.depth = bpp == 1 ? 8 : 10;
            d.margin_x = d.margin_y = 0;
            c->used[id] = true;
            c->owned[id] = nullptr;
        }
        c->planes_dirty = true;
        if (sync_planes(c)) { fprintf(stderr, "x265_amd: %s\n", c->err.c_str()); abort(); }
    }
    for (int k = 0; k < nref; k++)
        pairs[k] = x265amd_block_pair{ ids[0], 0, 0, ids[k + 1], 0, 0 };
    int32_t out[4];
    int rc = x265amd_cost_batch(c, kind, w, h, nref, pairs, out);
    if (rc) { fprintf(stderr, "x265_amd: primitive failed: %s\n", x265amd_last_error(c)); abort(); }
    for (int k = 0; k < nref; k++)
        res[k] = out[k];
    return 0;
}

template <int BPP, int KIND, int PART>
int prim_cmp(void *fenc, intptr_t fs, void *fref, intptr_t rs)
{
    int r;
    const void *refs[1] = { fref };
    prim_eval(BPP, KIND, k_dims[PART >> 3], k_dims[PART & 7], fenc, fs, refs, 1, rs, &r);
    return r;
}
template <int BPP, int PART>
void prim_x3(void *fenc, void *r0, void *r1, void *r2, intptr_t rs, int *res)
{
    const void *refs[3] = { r0, r1, r2 };
    prim_eval(BPP, X265AMD_SAD, k_dims[PART >> 3], k_dims[PART & 7], fenc, X265AMD_FENC_STRIDE, refs, 3, rs, res);
}
template <int BPP, int PART>
void prim_x4(void *fenc, void *r0, void *r1, void *r2, void *r3, intptr_t rs, int *res)
{
    const void *refs[4] = { r0, r1, r2, r3 };
    prim_eval(BPP, X265AMD_SAD, k_dims[PART >> 3], k_dims[PART & 7], fenc, X265AMD_FENC_STRIDE, refs, 4, rs, res);
}

template <int BPP, int PART>
struct Fill
{
    static void run(x265amd_primitives *p)
    {
        p->sad[PART] = prim_cmp<BPP, X265AMD_SAD, PART>;
        p->satd[PART] = prim_cmp<BPP, X265AMD_SATD, PART>;
        p->sa8d_inter[PART] = prim_cmp<BPP, X265AMD_SA8D, PART>;
        p->sad_x3[PART] = prim_x3<BPP, PART>;
        p->sad_x4[PART] = prim_x4<BPP, PART>;
        Fill<BPP, PART - 1>::run(p);
    }
};
template <int BPP>
struct Fill<BPP, -1> { static void run(x265amd_primitives*) {} };

} // namespace

extern "C" int x265amd_setup_primitives(x265amd_primitives *p, int device, int bytes_per_px)
{
    if (!p || (bytes_per_px != 1 && bytes_per_px != 2))
        return X265AMD_ERR_INVALID;
    PrimCtx &pc = g_prim[bytes_per_px];
    {
        std::lock_guard<std::mutex> g(pc.mu);
        if (!pc.ctx)
        {
            int rc = x265amd_create(device, &pc.ctx);
            if (rc)
                return rc;
            pc.bpp = bytes_per_px;
        }
    }
    if (bytes_per_px == 1)
        Fill<1, 63>::run(p);
    else
        Fill<2, 63>::run(p);
    // sa8d[] for square intra blocks equals sa8d_inter of the square partition
    // (reference: common/pixel.cpp:750-754 vs :756-826)
    for (int i = 0; i < 5; i++)
        p->sa8d[i] = p->sa8d_inter[x265amd_partition_from_sizes(4 << i, 4 << i)];
    return X265AMD_OK;
}
