// __global__ kernels of the motion-estimation engine (gfx950).
#pragma once

#include "me_search.h"

namespace xa {

constexpr int MAX_PLANES = 256;
constexpr int FULL_MAX_CAND = 256;       // candidates per axis the exhaustive kernel handles (larger ranges: general kernel)

struct BatchParams
{
    const PlaneDesc *planes;             // device array of registered planes
    const uint16_t  *cost[52];           // per-QP MV cost tables (centre pointers); NULL if not built
    int32_t          enc_plane;
    int32_t          ref_planes[X265AMD_MAX_REFS];
    int32_t          n_refs;
    int32_t          flags;               // JF_* bits applied to every job (tuning aids)
    unsigned        *hist;                // optional: evaluations spent on job i, written to hist[i]
    const unsigned char *fb_jobs;         // optional: fb_jobs[i] != 0 = an exhaustive kernel ran job i's integer stage
                                          // (without it: exactly the jobs full_kernel_eligible accepts)
    int32_t          lane_on;             // the kernels of me_lane.h take the jobs group_job_class names (0: none), not k_search
};

// The field checks every entry path applies to a job: the host entry points refuse a batch that fails them
// (X265AMD_ERR_INVALID); the device-array entry points, whose jobs the host cannot see, apply them per job
// on the device: a failing job gets the result (MV 0, cost X265AMD_ERR_INVALID) and is never searched.
__device__ __host__ inline bool job_dim_valid(int v) { return v == 4 || v == 8 || v == 12 || v == 16 || v == 24 || v == 32 || v == 48 || v == 64; }
template <class J>
__device__ __host__ inline bool job_fields_valid(const J &j, int n_refs)
{
    if (!job_dim_valid(j.w) || !job_dim_valid(j.h))
        return false;
    if (j.method > X265AMD_BIPRED || j.subme > 7 || j.qp > 51 || j.numc > X265AMD_MAX_MVC ||
        j.ref_idx >= n_refs || j.icost > X265AMD_SA8D || j.merange < 0)
        return false;
    if (j.method == X265AMD_UMH && j.w == 4 && j.h == 4)      // (the reference asserts, encoder/motion.cpp:447)
        return false;
    if (j.method == X265AMD_BIPRED && (j.icost != X265AMD_SAD || j.mvmin_x > j.mvmax_x || j.mvmin_y > j.mvmax_y))
        return false;
    return true;
}

// Does the dedicated exhaustive kernel handle this job?  (SAD metric, bounded range)
template <class J>
__device__ __host__ inline bool full_kernel_eligible(const J &j, int bpp)
{
    int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    return (j.method == X265AMD_FULL || j.method == X265AMD_BIPRED) && j.icost == X265AMD_SAD && (bpp == 1 || bpp == 2) &&
           nx > 0 && ny > 0 && nx <= FULL_MAX_CAND && ny <= FULL_MAX_CAND;
}

// Can the candidate-per-lane exhaustive kernel (me_had.h) take this job?  SA8D or SAD metric; bounded range.  (The caller also checks that the PU is a square of a 64x64 cell's
// quadtree and, at launch, the planes' bit depth: at most 10 for SA8D, 12 for SAD.)
template <class J>
__device__ __host__ inline bool had_kernel_eligible(const J &j, int bpp)
{
    int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    if (nx <= 0 || ny <= 0 || nx > FULL_MAX_CAND || ny > FULL_MAX_CAND || (bpp != 1 && bpp != 2))
        return false;
    if (j.icost == X265AMD_SA8D)
        return j.method == X265AMD_FULL;
    return full_kernel_eligible(j, bpp);
}

// ------------------------------------------------------------------------------------------
// General search: any method / sub-pel level / pixel type.  Jobs come in three classes by PU area:
// up to 256 pixels and up to 1024 pixels on one wavefront (one and up to four passes over the PU),
// larger on four or eight (the host picks: four beside launches of the small classes, whose
// throughput wants the registers and wavefront slots a large PU's waiting workers hold; eight
// alone).  `idx` (optional) lists the jobs of this launch; without it every job is visited and
// those of the other classes are skipped.
constexpr int JOB_CLASSES = 5;           // 0 .. 2: the general kernel's classes; LANE_CLASS / ROW_CLASS: the quad- / row-per-job kernel's jobs
constexpr int LANE_CLASS = 3, ROW_CLASS = 4;
constexpr int LANE_MAX_AREA = 64;        // PU pixels of a quad-per-job search (four 4x4 tiles): 8x8, 8x4, 4x8, 16x4, 4x16, 4x4, 12x4, 4x12
constexpr int ROW_MAX_AREA = 1024;       // PU pixels of a row-per-job search: what is left of a FULL job whose scan an exhaustive kernel ran (off by default)
constexpr int EVAL_SLOTS = 256;          // evaluation counters (summed on read-out)
__device__ __host__ inline int job_class(int w, int h) { return w * h <= 256 ? 0 : w * h <= 1024 ? 1 : 2; }
// Which of the kernels of me_lane.h takes this job, if any: LANE_CLASS (four lanes per job): any search of a PU of at most
// LANE_MAX_AREA pixels; ROW_CLASS (sixteen lanes per job): a FULL job of a PU of at most `lane_on` pixels whose scan ran
// in an exhaustive kernel (`covered`): only the predictor checks, the slab combine and the sub-pel rounds are left of it.
// `lane_on`: 0 = neither (a reference of the batch is a half-resolution set), else the row class's largest area (at
// least LANE_MAX_AREA, which switches that class off).  Returns -1 for a job of the general kernel.
template <class J>
__device__ __host__ inline int group_job_class(const J &j, bool covered, int lane_on)
{
    const int area = j.w * j.h;
    if (!lane_on)
        return -1;
    if (area <= LANE_MAX_AREA)
        return LANE_CLASS;
    return (covered && j.method == X265AMD_FULL && area <= lane_on) ? ROW_CLASS : -1;
}
template <class J>
__device__ __host__ inline bool lane_job(const J &j, bool covered, int lane_on) { return group_job_class(j, covered, lane_on) >= 0; }

template <class L>
__global__ void __launch_bounds__(L::NT) __attribute__((amdgpu_waves_per_eu(L::NW == 1 && L::INL ? 8 : 1))) k_search(BatchParams P, const x265amd_me_job *jobs, const int *idx, int n, int cls,
                                                  const FullBest *fb, int fb_stride, x265amd_me_result *out,
                                                  unsigned long long *evals)
{
    __shared__ __align__(16) L lds;
    if ((int)blockIdx.x >= n)
        return;
    const int id = idx ? idx[blockIdx.x] : (int)blockIdx.x;
    JobRegs j;
    j.load(jobs + id);
    if (!idx && !job_fields_valid(j, P.n_refs))
    {
        // (only the device-array entry points come without an index list; planned lists were checked on the host)
        if (cls == 0 && threadIdx.x == 0)
        {
            out[id].mvx = out[id].mvy = 0;
            out[id].cost = X265AMD_ERR_INVALID;
        }
        return;
    }
    const bool covered = fb && (P.fb_jobs ? P.fb_jobs[id] != 0 : full_kernel_eligible(j, (int)sizeof(typename L::PIX)));
    // (every launch visits its whole list - or every job, without one - and skips the jobs of the other classes)
    if (job_class(j.w, j.h) != cls || (!idx && lane_job(j, covered, P.lane_on)))
        return;
    const PlaneDesc &enc = P.planes[P.enc_plane];
    const PlaneDesc &ref = P.planes[P.ref_planes[j.ref_idx]];
    if (L::NW > 1 && Dev<L>::wave_id() != 0)
    {
        Dev<L>::worker((uint32_t)(uintptr_t)(XA_LDS L*)&lds);
        return;
    }
    Wave<L> W;
    W.stamp(0);
    W.bind(lds, enc, ref, j.pu_x, j.pu_y, j.pu_x, j.pu_y, j.w, j.h, P.cost[j.qp], P.flags);
    W.stamp(1);
    Search<L> S(W, j);
    int ox, oy;
    const FullBest *f = covered ? fb + (size_t)id * fb_stride : nullptr;
    int cost = S.run(f, fb_stride, ox, oy);
    Dev<L>::post(W.base, Dev<L>::OP_EXIT);
    // (evaluation counts go to one of EVAL_SLOTS counters, 128 bytes apart, to spread the atomics)
    Dev<L>::publish(out + id, ox, oy, cost, evals + (blockIdx.x % EVAL_SLOTS) * 16, W.evals, P.hist ? P.hist + id : nullptr);
#ifdef XA_PROFILE
    W.stamp(6);
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0)      // (a sample: every job would serialise on the atomics)
        for (int k = 0; k < 6; k++)
            atomicAdd(evals + EVAL_SLOTS * 16 + k, (unsigned long long)(W.tstamp[k + 1] - W.tstamp[k]));
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0)
        for (int k = 0; k < 6; k++)
            atomicAdd(evals + (EVAL_SLOTS + 1) * 16 + 8 + k, (unsigned long long)W.cgen[k]);
#endif
}

// ------------------------------------------------------------------------------------------
// Batched block costs: one wavefront per (enc block, ref block) pair.
template <typename PIX>
__global__ void __launch_bounds__(64) k_cost(const PlaneDesc *planes, int kind, int w, int h,
                                             const x265amd_block_pair *pairs, int n, int32_t *out)
{
    typedef JobLds<PIX, 1, 1024> L;
    __shared__ __align__(16) L lds;
    int id = blockIdx.x;
    if (id >= n)
        return;
    const x265amd_block_pair &p = pairs[id];
    Wave<L> W;
    W.bind(lds, planes[p.enc_plane], planes[p.ref_plane], p.enc_x, p.enc_y, p.ref_x, p.ref_y, w, h, nullptr,
           JF_NO_WINDOW | JF_NO_SLICES);
    int c = W.cost_direct(kind);
    if (threadIdx.x == 0)
        out[id] = c;
}

// ------------------------------------------------------------------------------------------
// Batched costs of motion-compensated blocks: one wavefront per (PU, q-pel MV): the PU at (enc_x, enc_y)
// against the prediction block of the same position displaced by the MV in the reference plane,
// interpolated as TComPrediction::xPredInterLumaBlk / subpelCompare do (TComPrediction.cpp:464-496,
// encoder/motion.cpp:1140-1165).
template <typename PIX>
__global__ void __launch_bounds__(64) k_mc_cost(const PlaneDesc *planes, int kind, int w, int h,
                                                const x265amd_mc_pair *pairs, int n, int32_t *out)
{
    typedef JobLds<PIX, 1, 1024> L;
    __shared__ __align__(16) L lds;
    int id = blockIdx.x;
    if (id >= n)
        return;
    const x265amd_mc_pair &p = pairs[id];
    Wave<L> W;
    W.bind(lds, planes[p.enc_plane], planes[p.ref_plane], p.pu_x, p.pu_y, p.pu_x, p.pu_y, w, h, nullptr,
           JF_NO_WINDOW | JF_NO_SLICES);
    int c = W.subpel_cost(XA_S16(p.qmv_x), XA_S16(p.qmv_y), kind, false);
    if (threadIdx.x == 0)
        out[id] = c;
}

// ------------------------------------------------------------------------------------------
// The source block of a bidirectional-refinement search, made on the device: dst = ClipY(2 * org - pred) over the PU
// (TComYuv::removeHighFreq, Lib/TLibCommon/TComYuv.cpp:673-702, luma part), pred = the other list's motion-compensated
// prediction of the PU at its q-pel MV (TComPrediction::xPredInterLumaBlk, Lib/TLibCommon/TComPrediction.cpp:464-496; the
// same four filter primitives as subpelCompare).  One thread per 4 adjacent pixels; reads of the reference are clamped
// into its allocation like every other read of the library.
template <typename PIX>
__global__ void __launch_bounds__(256) k_bipred_src(const PlaneDesc *planes, int dst_plane, int w, int h,
                                                   const x265amd_mc_pair *pairs, int n)
{
    typedef Dev<JobLds<PIX, 1, 64>> D;
    constexpr int DW = D::DW;
    const int ngx = w >> 2, G = ngx * h;
    const int t = (int)(blockIdx.x * 256 + threadIdx.x);
    if (t >= n * G)
        return;
    const int item = t / G, g = t - item * G, y = g / ngx, x = (g - y * ngx) * 4;
    const x265amd_mc_pair &p = pairs[item];
    const PlaneDesc &org = planes[p.enc_plane], &ref = planes[p.ref_plane], &dst = planes[dst_plane];
    const int qx = XA_S16(p.qmv_x), qy = XA_S16(p.qmv_y), xf = qx & 3, yf = qy & 3;
    const int depth = DW == 1 ? 8 : ref.depth, maxval = (1 << depth) - 1;
    const int64_t lo = ref.lo, hi = ref.hi;
    const PIX *r0 = (const PIX*)ref.origin;
    const int64_t off = (int64_t)(p.pu_y + y + (qy >> 2)) * ref.stride + p.pu_x + x + (qx >> 2);
    auto clampo = [&](int64_t o, int len) { o = o < lo ? lo : o; return o > hi - len ? hi - len : o; };
    auto rd4 = [&](int64_t o, uint32_t *r) {
        const PIX *q = r0 + clampo(o, 4);
        r[0] = ld_u32(q);
        r[1] = DW == 2 ? ld_u32((const uint8_t*)q + 4) : 0;
    };
    auto rd12 = [&](int64_t o, uint32_t *r) {
        const PIX *q = r0 + clampo(o, 12);
#pragma unroll
        for (int i = 0; i < 3 * DW; i++)
            r[i] = ld_u32((const uint8_t*)q + 4 * i);
    };
    int pred[4];
    if (!xf && !yf)
    {
        uint32_t d[2];
        rd4(off, d);
        const P4 v = D::unpack4(d);
#pragma unroll
        for (int i = 0; i < 4; i++)
            pred[i] = v.v[i];
    }
    else if (!yf)
    {
        uint32_t px[6];
        rd12(off - 3, px);
        D::h_pp(px, xf, depth, pred);
    }
    else if (!xf)
    {
        uint32_t rw[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++)
            rd4(off + (int64_t)(k - 3) * ref.stride, rw[k]);
        D::v_pp(rw, yf, depth, pred);
    }
    else
    {
        uint32_t im[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            uint32_t px[6];
            rd12(off + (int64_t)(k - 3) * ref.stride - 3, px);
            int sm[4];
            D::h_ps(px, xf, depth, sm);
            im[k][0] = __builtin_amdgcn_perm((uint32_t)sm[1], (uint32_t)sm[0], 0x05040100);
            im[k][1] = __builtin_amdgcn_perm((uint32_t)sm[3], (uint32_t)sm[2], 0x05040100);
        }
        D::v_sp(im, yf, depth, pred);
    }
    const PIX *o = (const PIX*)org.origin + (int64_t)(p.pu_y + y) * org.stride + p.pu_x + x;
    PIX *d = (PIX*)const_cast<void*>(dst.origin) + (int64_t)(p.pu_y + y) * dst.stride + p.pu_x + x;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int v = 2 * (int)o[i] - pred[i];
        d[i] = (PIX)(v < 0 ? 0 : v > maxval ? maxval : v);
    }
}

// ------------------------------------------------------------------------------------------
// The four 8-tap luma filter primitives on their own (common/primitives.h:186-188, 263-265;
// common/ipfilter.cpp:38-139, 265-309, 354-425), through the same device functions the search's
// prediction code uses.  kind 0 = ipfilter_pp[H_8], 1 = ipfilter_pp[V_8], 2 = ipfilter_ps[H_8]
// (pixel -> short), 3 = ipfilter_sp[V_8] (short -> pixel).  `src` addresses element (0,0) of the block
// inside a staged copy that holds 3 rows / columns before and 4 (+3 of load slack) behind it; one
// thread produces 4 adjacent outputs; `dst` is a dense w x h array.
template <typename PIX>
__global__ void __launch_bounds__(256) k_ipfilter(int kind, const void *src, int sstride, void *dst, int w, int h, int frac, int depth)
{
    typedef Dev<JobLds<PIX, 1, 64>> D;
    const int ngx = (w + 3) >> 2;
    const int t = (int)(blockIdx.x * 256 + threadIdx.x);
    if (t >= ngx * h)
        return;
    const int y = t / ngx, x = (t - y * ngx) * 4;
    int out[4];
    if (kind == 0 || kind == 2)
    {
        const PIX *p = (const PIX*)src + (int64_t)y * sstride + x - 3;
        uint32_t px[6];
#pragma unroll
        for (int i = 0; i < 3 * D::DW; i++)
            px[i] = ld_u32((const uint8_t*)p + 4 * i);
        if (kind == 0) D::h_pp(px, frac, depth, out);
        else           D::h_ps(px, frac, depth, out);
    }
    else if (kind == 1)
    {
        uint32_t rw[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const PIX *p = (const PIX*)src + (int64_t)(y + k - 3) * sstride + x;
            rw[k][0] = ld_u32(p);
            rw[k][1] = D::DW == 2 ? ld_u32((const uint8_t*)p + 4) : 0;
        }
        D::v_pp(rw, frac, depth, out);
    }
    else
    {
        uint32_t rw[8][2];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const uint64_t d = ld_u64((const int16_t*)src + (int64_t)(y + k - 3) * sstride + x);
            rw[k][0] = (uint32_t)d;
            rw[k][1] = (uint32_t)(d >> 32);
        }
        D::v_sp(rw, frac, depth, out);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if (x + i < w)
        {
            if (kind == 2) ((int16_t*)dst)[(int64_t)y * w + x + i] = (int16_t)out[i];
            else           ((PIX*)dst)[(int64_t)y * w + x + i] = (PIX)out[i];
        }
}

} // namespace xa
