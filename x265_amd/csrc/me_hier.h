// Exhaustive (FULL_SEARCH) integer stage for ALL square PUs of a 64x64 picture cell from ONE pass
// over the search window (8-bit pixels, SAD metric).
//
// SAD is additive over a partition of the block: the SAD of a 16x16 PU at candidate m is the sum of
// the SADs of its four 8x8 quarters at the same m, and so on up to 64x64 (the reference relies on the
// same identity when it builds large-block primitives from small ones, common/x86/asm-primitives.cpp:
// 50-59).  A "bundle" is the set of FULL-search jobs whose PUs are the squares of the quadtree of one
// 64x64 cell (8x8, 16x16, 32x32, 64x64; any subset) and that share reference, QP and predictor, hence
// one search window and one MV-cost slice.  Per candidate the kernel measures only the 8x8 cells -
// a quarter of the pixel-SADs of four independent scans - and sums them up the quadtree in Z-order.
//
// Mapping.  A work item = (bundle, range of lane tasks); one 128-thread workgroup stages the window rows
// those tasks touch into LDS.  A lane task = 4 horizontally adjacent candidates x HIER_R candidate rows:
// v_qsad_pk_u16_u8 prices the 4 byte-shifted candidates of one 4-pixel group per instruction, every
// window dword pair read from LDS feeds HIER_R candidate rows, and the encoder pixels arrive as SCALAR
// operands (the bundle's 64x64 block is re-packed cell by cell in Z-order by k_pack_fenc, so one
// s_load_dwordx16 fetches a whole 8x8 cell).  Packed 16-bit partial sums hold up to 16x16 (<= 65280);
// 32x32 and 64x64 continue in 32 bits.
//
// Winner per PU = min over (cost, raster index), the candidate the reference's strict-`<` raster scan
// keeps (encoder/motion.cpp:738-771): as soon as a PU's sums are complete its 4 x HIER_R candidate keys
// are reduced inside the lane, across the wavefront (DPP) and into a per-wavefront LDS slot - no key
// lives in registers across cells.  Keys of 8x8 / 16x16 PUs are 32-bit (cost < 2^17 in the top bits,
// 15-bit raster index of the bundle window below); invalid candidates (beyond the window, or outside
// a job's own clipped range) get a key at or above HIER_KEY_NONE, which no real candidate reaches.  Slabs of one bundle write
// partial winners (FullBest) that the general kernel combines, like the slabs of me_full.h.
#pragma once

#include "me_full.h"

namespace xa {

#ifdef XA_PROFILE
static __device__ unsigned long long g_hier_prof[8];
static __device__ unsigned long long g_hier_trace[3 * 16384];   // per work item of a launch: start, staged, end (100 MHz counter)     // developer aid: clocks {item total, staging, cell loop, epilogue}, items
#endif
#ifndef XA_HIER_R
#define XA_HIER_R 4
#endif
#ifndef XA_HIER_THREADS
#define XA_HIER_THREADS 128
#endif
constexpr int HIER_R = XA_HIER_R;                 // candidate rows per lane task
static_assert(4 * HIER_R <= 32, "hier_lane_min64 keeps the slot of a lane's candidate in 5 bits");
constexpr int HIER_THREADS = XA_HIER_THREADS;
constexpr int HIER_WAVES = HIER_THREADS / 64;
#ifndef XA_HIER_PF
#define XA_HIER_PF 4
#endif
constexpr int HIER_PF = XA_HIER_PF;        // window rows of the next cell fetched into registers a cell ahead
#ifndef XA_HIER_WPS
#define XA_HIER_WPS 4
#endif
constexpr int HIER_WAVES_PER_SIMD = XA_HIER_WPS;   // occupancy the kernel is compiled for
constexpr int HIER_SLOT_ROWS = 4;         // winner slots of a small PU per wavefront: one per 16-lane row
constexpr int HIER_PUS = 85;              // 64 + 16 + 4 + 1 squares of the quadtree
constexpr int HIER_SMALL = 80;            // PUs 0..79 (8x8, 16x16) use 32-bit keys
constexpr int HIER_RIDX_BITS = 15;        // raster index bits: window candidates <= 2^15
constexpr int HIER_FENC_DW = 1024;        // packed encoder block: 64 cells x 16 dwords
constexpr int HIER_MAX_RECT = 128;        // rectangular PUs of one bundle (me_had.h; every 8-aligned HEVC shape of a CTU: 124)
// 32-bit keys: (SAD + mvcost) << 15 | raster index.  The kernel is used only while no MV-cost table entry
// exceeds HIER_MVCOST_MAX / 2 (the reference's tables peak at 3381, encoder/bitcost.cpp:48-56 with the
// largest lambda), so a valid key stays below (65280 + 8191) << 15 | 32767 < HIER_KEY_NONE, and every
// key at or above HIER_KEY_NONE marks "not a candidate".
constexpr uint32_t HIER_MVCOST_MAX = 8191;
constexpr uint32_t HIER_KEY_NONE = 0xE0000000u;

struct HierBundle
{
    int32_t  ctu_x, ctu_y;      // picture position of the 64x64 cell
    int32_t  mvx0, mvy0;        // first candidate MV (full-pel) of the bundle window = union of the jobs' ranges
    int32_t  nxw, nyw;          // candidates per row / candidate rows of the window
    int32_t  qmvp_x, qmvp_y;    // the jobs' common predictor (q-pel) and QP
    int32_t  qp, ref_idx;
    int32_t  uniform;           // every job's range equals the window (no per-job masking)
    // Jobs with DIFFERENT predictors in one bundle (me_hier.h only; pred = 0: one predictor, qmvp_x / qmvp_y above).  A job's
    // MV cost depends on the candidate only through d = mv - (qmvp >> 2) and the phase qmvp & 3, so the kernel keeps four
    // phase tables per axis over d = d0 .. d0 + len - 1 and every job reads them at its own offset.  Candidates whose d lies
    // outside [ulo, uhi] (the union of the jobs' ranges, relative to their predictors) are no candidates of any job;
    // pred = 1: every job's range is exactly [ulo, uhi] around its predictor, pred = 2: some are smaller (compared per job).
    int32_t  pred;
    int32_t  dx0, dy0, lx, ly;
    int32_t  ulo_x, uhi_x, ulo_y, uhi_y;
    int32_t  metric;            // X265AMD_SAD / X265AMD_SA8D: the jobs' integer-stage metric (me_had.h; me_hier.h is SAD only)
    uint64_t cells;             // bit z: 8x8 cell z (Z-order) is covered by some job
    int32_t  job[HIER_PUS];     // job of square k (0..63: 8x8 in Z-order, 64..79: 16x16, 80..83: 32x32, 84: 64x64), -1: none
    // Rectangular PUs (me_had.h only; `kind` as had_rect_value documents, bit 31: wide keys), sorted by where they are
    // complete: rect_end[q], q = 0..15: PUs inside 16x16 block q (Z-order); rect_end[16 + b]: PUs of 32x32 block b;
    // rect_end[20]: PUs made of 16x16 values of the whole cell.  rect_end[g] = end of group g in rect[].
    int32_t  rect_end[21];
    int32_t  quads;             // me_had.h only: the bundle has PUs with 4- or 12-pixel sides (their jobs: HierQuads of this bundle)
    // bundles formed on the device (me_plan.h): work items (slabs) of the bundle, lane tasks per work item, LDS row pitch and
    // the LDS bytes its largest work item needs; n_slabs = 0: this cell of the planner's table holds no bundle
    int32_t  n_slabs, per, pitch, lds_need;
    struct Rect { int32_t job; uint32_t kind; } rect[HIER_MAX_RECT];
};

// The PUs of a bundle with 4- or 12-pixel sides (me_had.h), -1: no such job.  Per 8x8 cell z (Z-order) its halves:
// 0 = 8x4 rows 0-3, 1 = 8x4 rows 4-7, 2 = 4x8 columns 0-3, 3 = 4x8 columns 4-7.  Per 16x16 block q (Z-order) the
// asymmetric partitions: 0 = 16x4 at row 0, 1 = 16x12 at row 4, 2 = 16x12 at row 0, 3 = 16x4 at row 12,
// 4 = 4x16 at column 0, 5 = 12x16 at column 4, 6 = 12x16 at column 0, 7 = 4x16 at column 12.
struct HierQuads
{
    int32_t cell[64][4];
    int32_t blk[16][8];
};
constexpr int HIER_QUAD_PUS = 64 * 4 + 16 * 8;

// A work item: lane tasks [t_lo, t_hi) of a bundle, in raster order of (row quad, x group)
struct HierItem
{
    int32_t bundle;
    int32_t t_lo, t_hi;
    int32_t slot;               // FullBest slot (slab number)
    int32_t pitch, win_nd;      // LDS row pitch / staged dwords per window row
    int32_t win_rows;           // window rows staged, from candidate row HIER_R * (t_lo / XG)
    int32_t pad;
};

#define XA_CONST __attribute__((address_space(4)))

// Z-order cell index -> cell coordinates inside the 64x64 block
__device__ __host__ __forceinline__ void hier_cell_xy(int z, int &cx, int &cy)
{
    cx = (z & 1) | ((z >> 1) & 2) | ((z >> 2) & 4);
    cy = ((z >> 1) & 1) | ((z >> 2) & 2) | ((z >> 3) & 4);
}

// Re-pack the 64x64 encoder block of every bundle: cell z (Z-order) -> 16 consecutive dwords
// (row r, half k at [2 r + k]).
__global__ void __launch_bounds__(256) k_pack_fenc(BatchParams P, const HierBundle *bundles, int n, uint32_t *packed)
{
    const int b = blockIdx.x;
    if (b >= n)
        return;
    const PlaneDesc &enc = P.planes[P.enc_plane];
    const HierBundle &B = bundles[b];
    if (B.nxw <= 0)         // (a cell of the device planner's table that holds no bundle)
        return;
    const uint8_t *org = (const uint8_t*)enc.origin;
    for (int i = threadIdx.x; i < HIER_FENC_DW; i += 256)
    {
        const int z = i >> 4, r = (i >> 1) & 7, k = i & 1;
        int cx, cy;
        hier_cell_xy(z, cx, cy);
        int64_t o = (int64_t)(B.ctu_y + cy * 8 + r) * enc.stride + B.ctu_x + cx * 8 + 4 * k;
        o = o < enc.lo ? enc.lo : (o > enc.hi - 4 ? enc.hi - 4 : o);
        packed[(size_t)b * HIER_FENC_DW + i] = ld_u32(org + o);
    }
}

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t hier_min_dpp(uint32_t v)
{
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)~0u, (int)v, CTRL, ROW_MASK, 0xf, false);
    return o < v ? o : v;
}
// minimum over the wavefront; valid in lane 63
__device__ __forceinline__ uint32_t hier_wave_min_last(uint32_t v)
{
    v = hier_min_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v = hier_min_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v = hier_min_dpp<0x141>(v);           // row_half_mirror
    v = hier_min_dpp<0x140>(v);           // row_mirror
    v = hier_min_dpp<0x142, 0xa>(v);      // row_bcast15 into rows 1 and 3
    v = hier_min_dpp<0x143, 0xc>(v);      // row_bcast31 into rows 2 and 3
    return v;
}
// minimum over every 16-lane row of the wavefront, in all lanes of the row
__device__ __forceinline__ uint32_t hier_row_min(uint32_t v)
{
    v = hier_min_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
    v = hier_min_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
    v = hier_min_dpp<0x141>(v);           // row_half_mirror
    v = hier_min_dpp<0x140>(v);           // row_mirror
    return v;
}

// LDS bytes of a work item besides its window and MV-cost tables: 32-bit winner slots [wavefront][16-lane row][80],
// 64-bit slots [wavefront][5], padding of the carve-up
__device__ __host__ constexpr int hier_slot_bytes()
{
    return 4 * HIER_WAVES * HIER_SLOT_ROWS * HIER_SMALL + 8 * HIER_WAVES * 5 + 64;
}
// LDS row pitch (dwords) of a window of nd dwords per row.  (Measured on the MI355X: the pitch makes no difference to the
// kernel's time - its LDS reads are far from the LDS's rate - so the smallest one is taken, odd so that rows shift banks.)
__device__ __host__ constexpr int hier_pitch(int nd) { return nd | 1; }

// D = S0.u16 (low / high half) * S1.u16 + S2.u32 in one instruction
__device__ __forceinline__ uint32_t hier_mad_lo(uint32_t p, uint32_t mul, uint32_t add)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(p), "s"(mul), "v"(add));
    return r;
}
__device__ __forceinline__ uint32_t hier_mad_hi(uint32_t p, uint32_t mul, uint32_t add)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(p), "s"(mul), "v"(add));
    return r;
}
// the same with the result saturated at 2^32 - 1 (the VOP3 clamp bit saturates this instruction on gfx950:
// tools/mad_clamp_probe.hip)
__device__ __forceinline__ uint32_t hier_mad_lo_sat(uint32_t p, uint32_t mul, uint32_t add)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(p), "s"(mul), "v"(add));
    return r;
}
__device__ __forceinline__ uint32_t hier_mad_hi_sat(uint32_t p, uint32_t mul, uint32_t add)
{
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0] clamp" : "=v"(r) : "v"(p), "s"(mul), "v"(add));
    return r;
}

// D = min(S0 + S1, 2^32 - 1)
__device__ __forceinline__ uint32_t hier_add_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The lane's minimum key of a small PU (8x8, 16x16) from packed 16-bit sums p[i] (4 candidates of row i).  A key is
// (cost << 15) + raster index = (SAD << 15) + column term + row term: cxk[x] = (x cost << 15) + column tag of candidate
// column x, cyk[i] = (y cost << 15) + raster index of the row's first candidate (less the tag of column 0); either is
// HIER_KEY_NONE or more for a column / row that is no candidate of this job.  The row term is the same for the 4
// candidates of a row, so it is added after their minimum; both steps saturate, which keeps a struck-out candidate at or
// above HIER_KEY_NONE (a real key stays below it: sums <= 65280, MV costs <= HIER_MVCOST_MAX).
__device__ __forceinline__ uint32_t hier_lane_min32(const unsigned long long *p, const uint32_t *cxk, const uint32_t *cyk)
{
    const uint32_t mul = 1u << HIER_RIDX_BITS;
    uint32_t m = ~0u;
#pragma unroll
    for (int i = 0; i < HIER_R; i++)
    {
        const uint32_t lo = (uint32_t)p[i], hi = (uint32_t)(p[i] >> 32);
        const uint32_t k0 = hier_mad_lo_sat(lo, mul, cxk[0]), k1 = hier_mad_hi_sat(lo, mul, cxk[1]);
        const uint32_t k2 = hier_mad_lo_sat(hi, mul, cxk[2]), k3 = hier_mad_hi_sat(hi, mul, cxk[3]);
        uint32_t r = k0 < k1 ? k0 : k1;
        r = k2 < r ? k2 : r;
        r = k3 < r ? k3 : r;
        r = hier_add_sat(r, cyk[i]);
        m = r < m ? r : m;
    }
    return m;
}

// The lane's best candidate of a 32x32 / 64x64 PU from 32-bit sums, as a 64-bit key (cost << 32 | raster index) for
// the cross-lane reduction; column / row terms as above (only their cost fields and "at or above HIER_KEY_NONE" are
// used).  Inside the lane the 4 x HIER_R candidates are compared as (cost << 5 | slot), slot = 4 * row + x being their
// raster order.
__device__ __forceinline__ unsigned long long hier_lane_min64(const uint32_t (*s)[4], const uint32_t *cxk, const uint32_t *cyk,
                                                              int x0, int y0, int nxw)
{
    constexpr uint32_t BIG = 0x3ffffffu;            // "not a candidate": above every real cost (< 2^21)
    uint32_t m = ~0u;
#pragma unroll
    for (int i = 0; i < HIER_R; i++)
#pragma unroll
        for (int x = 0; x < 4; x++)
        {
            const uint32_t t = hier_add_sat(cxk[x], cyk[i]);
            const uint32_t mv = t < HIER_KEY_NONE ? t >> HIER_RIDX_BITS : BIG;
            const uint32_t k = ((s[i][x] + mv) << 5) | (uint32_t)(i * 4 + x);
            m = k < m ? k : m;
        }
    const uint32_t cost = m >> 5, slot = m & 31;
    if (cost >= BIG)
        return ~0ull;
    return ((unsigned long long)cost << 32) | (uint32_t)((y0 + (int)(slot >> 2)) * nxw + x0 + (int)(slot & 3));
}

// One predictor for the whole bundle: the lane's column / row terms cxs / cys serve every job; a job whose own range is
// smaller than the window (MASK) strikes out the columns / rows outside it.
template <bool MASK>
__device__ __forceinline__ void hier_plain_terms(const x265amd_me_job *jobs, int jid, const HierBundle &B, int x0, int y0,
                                                 const uint32_t *cxs, const uint32_t *cys, uint32_t *cxk, uint32_t *cyk)
{
#pragma unroll
    for (int x = 0; x < 4; x++)
        cxk[x] = cxs[x];
#pragma unroll
    for (int i = 0; i < HIER_R; i++)
        cyk[i] = cys[i];
    if (MASK)
    {
        const x265amd_me_job &J = jobs[jid];
        const int jx0 = J.mvmin_x - B.mvx0, jx1 = J.mvmax_x - B.mvx0, jy0 = J.mvmin_y - B.mvy0, jy1 = J.mvmax_y - B.mvy0;
#pragma unroll
        for (int x = 0; x < 4; x++)
            cxk[x] = (x0 + x >= jx0 && x0 + x <= jx1) ? cxs[x] : HIER_KEY_NONE;
#pragma unroll
        for (int i = 0; i < HIER_R; i++)
            cyk[i] = (y0 + i >= jy0 && y0 + i <= jy1) ? cys[i] : HIER_KEY_NONE;
    }
}

// Bundles formed on the device (me_plan.h): bundle c is cell c of the picture's cell table and has n_slabs <= PLAN_SLABS work
// items; offsets[c] = work items of the cells before c (offsets[n_cells] = all of them).  The launch covers
// n_cells x PLAN_SLABS workgroups; workgroup b takes work item plan_item_of_block(b, total) - the XCD-interleaved order of the
// planned lists (xcd_interleave in x265_amd.hip: 8 * PLAN_BANDS bands of consecutive work items, XCD b % 8 takes bands x,
// x + 8, ... in turn) - and the workgroups beyond the last work item, all at the end of the launch, leave at once.
constexpr int PLAN_SLABS = 16;              // most work items per bundle (= FB_STRIDE_PLAN of the host planner)
// LDS size classes of the launches over device-formed bundles (a bundle runs in the smallest class that holds its largest
// work item): 8, 6 and 4 workgroups per CU.  1080p +-64 needs 19904 bytes with one predictor, about 26 K with one per PU.
constexpr int PLAN_LDS_CLASSES = 3;
__device__ __host__ constexpr int plan_lds_class(int k) { return k == 0 ? 19968 : k == 1 ? 27136 : 40960; }
constexpr int PLAN_BANDS = 2;
__device__ __forceinline__ int plan_item_of_block(int b, int total)
{
    const int per = (total + 8 * PLAN_BANDS - 1) / (8 * PLAN_BANDS);
    const int x = b & 7, t = b >> 3, g = t / per, k = t - g * per;
    const int i = (g * 8 + x) * per + k;
    return g < PLAN_BANDS && i < total ? i : -1;
}

// What hier_item_body keeps in LDS per PU of a bundle with several predictors
struct HierPredPu
{
    uint32_t tx, ty;            // where candidate column 0 / row 0 of the window lies in the job's phase table (~0u: no such job)
    uint32_t xr, yr;            // the job's own range in window coordinates: first | (last - first) << 16
};

// the job's column / row terms of the lane's 4 x HIER_R candidates (bundles with several predictors)
template <bool KILL>
__device__ __forceinline__ void hier_pred_terms(const uint32_t *s_tx, const uint32_t *s_ty, const HierPredPu &d, int x0, int y0,
                                                const uint32_t *ry, uint32_t *cxk, uint32_t *cyk)
{
    const uint32_t *tx = s_tx + d.tx + x0, *ty = s_ty + d.ty + y0;
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        cxk[x] = tx[x];
        if (KILL)
            cxk[x] = (uint32_t)(x0 + x) - (d.xr & 0xffffu) <= (d.xr >> 16) ? cxk[x] : HIER_KEY_NONE;
    }
#pragma unroll
    for (int i = 0; i < HIER_R; i++)
    {
        cyk[i] = hier_add_sat(ty[i], ry[i]);
        if (KILL)
            cyk[i] = (uint32_t)(y0 + i) - (d.yr & 0xffffu) <= (d.yr >> 16) ? cyk[i] : HIER_KEY_NONE;
    }
}

// One 8x8 cell against the lane's 4 x RL candidates: window row r meets encoder row r - i in candidate row i.  w holds the first
// HIER_PF window rows (fetched a cell ahead), the others are read here; p[i] = packed 16-bit SADs of row i's 4 candidates.
template <int RL>
__device__ __forceinline__ void hier_measure(const unsigned long long (*w)[2], const uint32_t *row, int pitch, const uint32_t *f,
                                             unsigned long long *p)
{
#pragma unroll
    for (int r = 0; r < 8 + RL - 1; r++)
    {
        const unsigned long long d0 = r < HIER_PF ? w[r][0] : lds_pair(row + r * pitch);
        const unsigned long long d1 = r < HIER_PF ? w[r][1] : lds_pair(row + r * pitch + 1);
#pragma unroll
        for (int i = 0; i < RL; i++)
        {
            const int e = r - i;
            if (e >= 0 && e < 8)
            {
                p[i] = __builtin_amdgcn_qsad_pk_u16_u8(d0, f[2 * e], p[i]);
                p[i] = __builtin_amdgcn_qsad_pk_u16_u8(d1, f[2 * e + 1], p[i]);
            }
        }
    }
}

// MODE 0: one predictor, every job's range is the window; 1: one predictor, per-job ranges; 2: per-job predictors, every job's
// range is the same around its predictor; 3: per-job predictors and ranges
template <int MODE, bool KEYS>
__device__ __forceinline__ void hier_item_body(const BatchParams &P, const x265amd_me_job *jobs, const HierBundle &B, const HierItem &it,
                                               const XA_CONST uint32_t *pf, FullBest *fb, int fb_stride, uint32_t *lds)
{
    constexpr int R = HIER_R;
    constexpr bool MASK = MODE == 1, PRED = MODE >= 2, KILL = MODE == 3;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const PlaneDesc &ref = P.planes[P.ref_planes[B.ref_idx]];
    const int nxw = B.nxw, nyw = B.nyw, pitch = it.pitch;
    const int XG = (nxw + 3) >> 2;
    const int yq_lo = it.t_lo / XG;                 // first row quad of this slab = first staged candidate row / R

    // LDS: window | cx[nxw] cy[nyw] (u16) | 32-bit slots [waves][16-lane rows][80] | 64-bit slots [waves][5]
    // (the window area is rounded up to an even dword count, so the 64-bit slots behind it are 8-byte aligned)
    // several predictors: window (a multiple of 4 dwords) | HierPredPu[85] | the slots | phase tables x[4][lxp] y[4][lyp] (u32)
    uint32_t *s_win = lds;
    uint16_t *s_cx = (uint16_t*)(s_win + ((it.win_rows * pitch + 1) & ~1));
    uint16_t *s_cy = s_cx + ((nxw + 3) & ~3);
    uint32_t *s_slot = (uint32_t*)(s_cy + ((nyw + 3) & ~3));
    HierPredPu *s_pd = (HierPredPu*)(s_win + ((it.win_rows * pitch + 3) & ~3));
    const int lxp = (B.lx + 3) & ~3, lyp = (B.ly + 3) & ~3;
    if (PRED)
        s_slot = (uint32_t*)(s_pd + HIER_PUS + 1);
    unsigned long long *s_big = (unsigned long long*)(s_slot + HIER_WAVES * HIER_SLOT_ROWS * HIER_SMALL);
    uint32_t *s_tx = (uint32_t*)(s_big + HIER_WAVES * 5 + 1), *s_ty = s_tx + 4 * lxp;

#ifdef XA_PROFILE
    const long long t_begin = XA_TICK();
    const unsigned long long r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- stage the window rows of this slab, the MV-cost slices, and clear the slots
    {
        const uint8_t *r0 = (const uint8_t*)ref.origin;
        const int64_t lo = ref.lo, hi = ref.hi - 4;
        const int wx0 = B.ctu_x + B.mvx0, wy0 = B.ctu_y + B.mvy0 + R * yq_lo;
        const int nd = it.win_nd;
        const int64_t first = (int64_t)wy0 * ref.stride + wx0, last = (int64_t)(wy0 + it.win_rows - 1) * ref.stride + wx0 + 4 * (nd - 1);
        const int chunks = (nd + 3) >> 2;               // 16-byte pieces of a window row
        if (first >= lo && last + 12 <= hi && chunks <= 64)
        {
            // the whole window (and the up to 12 bytes a row's last 16-byte piece reads beyond it) lies inside the plane
            // allocation: plain 32-bit offsets, 16 bytes per lane, 64 / chunks window rows per load instruction
            const uint8_t *w0 = r0 + first;
            const int stride = ref.stride;
            const int rpi = 64 / chunks, rr = lane / chunks, ch = lane - rr * chunks;
            struct __attribute__((packed)) U4 { uint32_t v[4]; };
            if (rr < rpi)
#pragma unroll 6
                for (int r = wv * rpi + rr; r < it.win_rows; r += HIER_WAVES * rpi)
                {
                    const XA_GLOBAL U4 *src = (const XA_GLOBAL U4*)(uintptr_t)(w0 + r * stride + 16 * ch);
                    const uint32_t d0 = src->v[0], d1 = src->v[1], d2 = src->v[2], d3 = src->v[3];
                    uint32_t *dst = s_win + r * pitch + 4 * ch;
                    dst[0] = d0;
                    if (4 * ch + 1 < nd) dst[1] = d1;
                    if (4 * ch + 2 < nd) dst[2] = d2;
                    if (4 * ch + 3 < nd) dst[3] = d3;
                }
        }
        else
        {
            for (int r = wv; r < it.win_rows; r += HIER_WAVES)
                for (int c = lane; c < nd; c += 64)
                {
                    int64_t a = (int64_t)(wy0 + r) * ref.stride + wx0 + 4 * c;
                    a = a < lo ? lo : (a > hi ? hi : a);
                    s_win[r * pitch + c] = ld_u32(r0 + a);
                }
        }
        const uint16_t *cost = P.cost[B.qp];
        if (PRED)
        {
            // phase table entry t of an axis: the cost of d = d0 + t at phase ph, shifted into the key's cost field; the
            // columns carry t itself as their tag (a job's columns are t = its offset + x: the same order as x)
            for (int i = tid; i < 4 * lxp; i += HIER_THREADS)
            {
                const int ph = i / lxp, t = i - ph * lxp, d = B.dx0 + t;
                s_tx[i] = (d >= B.ulo_x && d <= B.uhi_x) ? ((uint32_t)cost[(int16_t)(4 * d - ph)] << HIER_RIDX_BITS) + (uint32_t)t : HIER_KEY_NONE;
            }
            for (int i = tid; i < 4 * lyp; i += HIER_THREADS)
            {
                const int ph = i / lyp, t = i - ph * lyp, d = B.dy0 + t;
                s_ty[i] = (d >= B.ulo_y && d <= B.uhi_y) ? (uint32_t)cost[(int16_t)(4 * d - ph)] << HIER_RIDX_BITS : HIER_KEY_NONE;
            }
            if (tid <= HIER_PUS)
            {
                HierPredPu d = { ~0u, ~0u, 0, 0 };
                const int jid = tid < HIER_PUS ? B.job[tid] : -1;
                if (jid >= 0)
                {
                    const x265amd_me_job &J = jobs[jid];
                    // (window column 0 is mv = mvx0, i.e. d = mvx0 - (qmvp_x >> 2))
                    d.tx = (uint32_t)((J.qmvp_x & 3) * lxp + (B.mvx0 - (J.qmvp_x >> 2) - B.dx0));
                    d.ty = (uint32_t)((J.qmvp_y & 3) * lyp + (B.mvy0 - (J.qmvp_y >> 2) - B.dy0));
                    d.xr = (uint32_t)(J.mvmin_x - B.mvx0) | (uint32_t)(J.mvmax_x - J.mvmin_x) << 16;
                    d.yr = (uint32_t)(J.mvmin_y - B.mvy0) | (uint32_t)(J.mvmax_y - J.mvmin_y) << 16;
                }
                s_pd[tid] = d;
            }
        }
        else
        {
            for (int i = tid; i < nxw; i += HIER_THREADS)
                s_cx[i] = cost[(int16_t)((int16_t)((B.mvx0 + i) << 2) - B.qmvp_x)];
            for (int i = tid; i < nyw; i += HIER_THREADS)
                s_cy[i] = cost[(int16_t)((int16_t)((B.mvy0 + i) << 2) - B.qmvp_y)];
        }
        for (int i = tid; i < HIER_WAVES * HIER_SLOT_ROWS * HIER_SMALL; i += HIER_THREADS)
            s_slot[i] = ~0u;
        if (tid < HIER_WAVES * 5)
            s_big[tid] = ~0ull;
    }
    __syncthreads();
#ifdef XA_PROFILE
    const long long t_staged = XA_TICK();
    const unsigned long long r_staged = __builtin_amdgcn_s_memrealtime();
#endif

    // which squares of the quadtree have a job, as wave-uniform masks (no look-up per cell in the loop below):
    // has8 bit z: 8x8 cell z; hasb bits 0..15: 16x16 blocks, 16..19: 32x32 blocks, 20: the 64x64 block
    const uint64_t has8 = __ballot(B.job[lane] >= 0);
    const uint32_t hasb = (uint32_t)__ballot(lane < HIER_PUS - 64 && B.job[64 + (lane < HIER_PUS - 64 ? lane : 0)] >= 0);
    const uint64_t cells = B.cells;
    // The last task row of a window of 4 k + 1 candidate rows (+-64, +-16: 129, 33) holds ONE candidate row: a wavefront whose
    // lane tasks all lie in it measures that row only (a quarter of the quad-SADs).  A work item that straddles the start of
    // that task row gives it to its last wavefront and the tasks before it to the first.
    const int n_trows = (nyw + R - 1) / R, bottom0 = (n_trows - 1) * XG;
    const bool thin = R > 1 && nyw - R * (n_trows - 1) == 1;
    const bool split = thin && it.t_lo < bottom0 && bottom0 < it.t_hi && bottom0 - it.t_lo <= 64 * (HIER_WAVES - 1) && it.t_hi - bottom0 <= 64;
    const int swv = __builtin_amdgcn_readfirstlane(wv);
    for (int base = it.t_lo; base < it.t_hi; base += HIER_THREADS)
    {
        const int wave_first = split ? (swv == HIER_WAVES - 1 ? bottom0 : it.t_lo + 64 * swv) : base + 64 * swv;
        const int task = wave_first + lane;
        const bool active = task < (split && swv != HIER_WAVES - 1 ? bottom0 : it.t_hi);
        const bool one_row = thin && wave_first >= bottom0;
        const int tk = active ? task : it.t_lo;
        const int yq = tk / XG, xg = tk - yq * XG;
        const int x0 = 4 * xg, y0 = R * yq;
        uint32_t cxs[4], cys[R];        // one predictor: column / row terms of the lane's candidates' keys (hier_lane_min32)
        uint32_t ry[R];                 // several predictors: raster index of the first candidate of the lane's rows
#pragma unroll
        for (int i = 0; i < R; i++)
        {
            const int y = y0 + i;
            const bool vy = active && y < nyw;
            ry[i] = vy ? (uint32_t)(y * nxw) : HIER_KEY_NONE;
            if (!PRED)
                cys[i] = vy ? ((uint32_t)s_cy[y < nyw ? y : nyw - 1] << HIER_RIDX_BITS) + (uint32_t)(y * nxw) : HIER_KEY_NONE;
        }
        if (!PRED)
        {
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const int xx = x0 + x;
                cxs[x] = (active && xx < nxw) ? ((uint32_t)s_cx[xx < nxw ? xx : nxw - 1] << HIER_RIDX_BITS) + (uint32_t)xx : HIER_KEY_NONE;
            }
        }
        const uint32_t *wbase = s_win + (R * (yq - yq_lo)) * pitch + xg;

        uint32_t a64[R][4], a32[R][4];
        unsigned long long a16[R];
#pragma unroll
        for (int i = 0; i < R; i++)
        {
            a16[i] = 0;
#pragma unroll
            for (int x = 0; x < 4; x++)
                a64[i][x] = a32[i][x] = 0;
        }

        // the covered cells in Z-order; the next cell's encoder pixels (scalar loads) and window rows (LDS reads into
        // registers) are fetched while the current cell is measured and priced, so that a wavefront on its own - the
        // last work items of a launch - does not wait for the LDS at every cell
        uint64_t rem = cells;
        int z = rem ? __builtin_ctzll(rem) : 64;
        uint32_t f[16];
        {
            const XA_CONST uint32_t *fe = pf + (z & 63) * 16;
#pragma unroll
            for (int e = 0; e < 16; e++)
                f[e] = fe[e];
        }
        // (HIER_PF of the cell's 8 + R - 1 window rows are fetched a cell ahead; the others follow while the first ones are measured)
        unsigned long long w[HIER_PF ? HIER_PF : 1][2];     // window rows of the cell: dword pairs (k, k + 1) and (k + 1, k + 2)
        const uint32_t *row;
        {
            int cx, cy;
            hier_cell_xy(z & 63, cx, cy);
            row = wbase + (cy * 8) * pitch + cx * 2;
#pragma unroll
            for (int r = 0; r < HIER_PF; r++)
            {
                w[r][0] = lds_pair(row + r * pitch);
                w[r][1] = lds_pair(row + r * pitch + 1);
            }
        }
#pragma nounroll
        while (z < 64)
        {
            rem &= rem - 1;
            const int nz = rem ? __builtin_ctzll(rem) : 64;
            uint32_t fn[16];
            {
                const XA_CONST uint32_t *fe = pf + (nz & 63) * 16;
#pragma unroll
                for (int e = 0; e < 16; e++)
                    fn[e] = fe[e];
            }
            unsigned long long p[R];
#pragma unroll
            for (int i = 0; i < R; i++)
                p[i] = 0;
            if (one_row)
                hier_measure<1>(w, row, pitch, f, p);       // (the other rows' sums stay 0 and their keys are struck out)
            else
                hier_measure<R>(w, row, pitch, f, p);
            asm volatile("" ::: "memory");      // (the reads below start before the pricing, not behind it)
            {
                int cx, cy;
                hier_cell_xy(nz & 63, cx, cy);
                row = wbase + (cy * 8) * pitch + cx * 2;
#pragma unroll
                for (int r = 0; r < HIER_PF; r++)
                {
                    w[r][0] = lds_pair(row + r * pitch);
                    w[r][1] = lds_pair(row + r * pitch + 1);
                }
            }
            asm volatile("" ::: "memory");
            // 8x8 PU of this cell
            if ((has8 >> z) & 1)
            {
                uint32_t cxk[4], cyk[R];
                if (PRED)
                    hier_pred_terms<KILL>(s_tx, s_ty, s_pd[z], x0, y0, ry, cxk, cyk);
                else
                    hier_plain_terms<MASK>(jobs, MASK ? B.job[z] : 0, B, x0, y0, cxs, cys, cxk, cyk);
                const uint32_t m = hier_row_min(hier_lane_min32(p, cxk, cyk));
                if ((lane & 15) == 15)
                    atomicMin(s_slot + (wv * HIER_SLOT_ROWS + (lane >> 4)) * HIER_SMALL + z, m);    // (ds_min_u32: no read-back to wait for)
            }
#pragma unroll
            for (int i = 0; i < R; i++)
                a16[i] += p[i];         // four 16-bit lanes; no carry between them: every partial 16x16 sum <= 65280
            if ((nz >> 2) != (z >> 2))      // the 16x16 block is complete
            {
                const int q = z >> 2;
                if ((hasb >> q) & 1)
                {
                    uint32_t cxk[4], cyk[R];
                    if (PRED)
                        hier_pred_terms<KILL>(s_tx, s_ty, s_pd[64 + q], x0, y0, ry, cxk, cyk);
                    else
                        hier_plain_terms<MASK>(jobs, MASK ? B.job[64 + q] : 0, B, x0, y0, cxs, cys, cxk, cyk);
                    const uint32_t m = hier_row_min(hier_lane_min32(a16, cxk, cyk));
                    if ((lane & 15) == 15)
                        atomicMin(s_slot + (wv * HIER_SLOT_ROWS + (lane >> 4)) * HIER_SMALL + 64 + q, m);
                }
#pragma unroll
                for (int i = 0; i < R; i++)
                {
                    const uint32_t lo = (uint32_t)a16[i], hi = (uint32_t)(a16[i] >> 32);
                    a32[i][0] += lo & 0xffffu; a32[i][1] += lo >> 16;
                    a32[i][2] += hi & 0xffffu; a32[i][3] += hi >> 16;
                    a16[i] = 0;
                }
                if ((nz >> 4) != (z >> 4))  // the 32x32 block is complete
                {
                    const int q32 = z >> 4;
                    if ((hasb >> (16 + q32)) & 1)
                    {
                        uint32_t cxk[4], cyk[R];
                        if (PRED)
                            hier_pred_terms<KILL>(s_tx, s_ty, s_pd[80 + q32], x0, y0, ry, cxk, cyk);
                        else
                            hier_plain_terms<MASK>(jobs, MASK ? B.job[80 + q32] : 0, B, x0, y0, cxs, cys, cxk, cyk);
                        const unsigned long long m = wave_min_u64(hier_lane_min64(a32, cxk, cyk, x0, y0, nxw));
                        if (lane == 0)
                            atomicMin(s_big + wv * 5 + q32, m);         // (ds_min_u64)
                    }
#pragma unroll
                    for (int i = 0; i < R; i++)
#pragma unroll
                        for (int x = 0; x < 4; x++)
                        {
                            a64[i][x] += a32[i][x];
                            a32[i][x] = 0;
                        }
                }
            }
#pragma unroll
            for (int e = 0; e < 16; e++)
                f[e] = fn[e];
            z = nz;
        }
        if ((hasb >> 20) & 1)
        {
            uint32_t cxk[4], cyk[R];
            if (PRED)
                hier_pred_terms<KILL>(s_tx, s_ty, s_pd[84], x0, y0, ry, cxk, cyk);
            else
                hier_plain_terms<MASK>(jobs, MASK ? B.job[84] : 0, B, x0, y0, cxs, cys, cxk, cyk);
            const unsigned long long m = wave_min_u64(hier_lane_min64(a64, cxk, cyk, x0, y0, nxw));
            if (lane == 0)
                atomicMin(s_big + wv * 5 + 4, m);
        }
    }
#ifdef XA_PROFILE
    const long long t_looped = XA_TICK();
#endif
    __syncthreads();

    // ---- one thread per PU combines the wavefronts' winners and writes the slab's partial result
    if (tid < HIER_PUS)
    {
        const int jid = B.job[tid];
        if (jid >= 0)
        {
            FullBest f;
            int cost = 0x7fffffff, ridx = -1;
            if (tid < HIER_SMALL)
            {
                uint32_t m = ~0u;
                for (int w = 0; w < HIER_WAVES * HIER_SLOT_ROWS; w++)
                {
                    const uint32_t v = s_slot[w * HIER_SMALL + tid];
                    m = v < m ? v : m;
                }
                if (m < HIER_KEY_NONE)
                {
                    cost = (int)(m >> HIER_RIDX_BITS);
                    ridx = (int)(m & ((1u << HIER_RIDX_BITS) - 1));
                    if (PRED)       // (the key's low bits hold row start + column tag = raster index + the job's table offset)
                        ridx -= (int)(s_pd[tid].tx % (uint32_t)lxp);
                }
            }
            else
            {
                unsigned long long m = ~0ull;
                for (int w = 0; w < HIER_WAVES; w++)
                {
                    const unsigned long long v = s_big[w * 5 + (tid - HIER_SMALL)];
                    m = v < m ? v : m;
                }
                if (m != ~0ull)
                {
                    cost = (int)(m >> 32);
                    ridx = (int)(m & 0xffffffffu);
                }
            }
            f.cost = cost;
            f.mvx = ridx < 0 ? 0 : (int16_t)(B.mvx0 + ridx % nxw);
            f.mvy = ridx < 0 ? 0 : (int16_t)(B.mvy0 + ridx / nxw);
            if (KEYS)
            {
                // one key per job for all slabs: cost, then raster order of the MV (what combine_slabs picks among slab slots)
                if (ridx >= 0)
                    atomicMin((unsigned long long*)(fb + (size_t)jid * fb_stride),
                              ((unsigned long long)(uint32_t)cost << 32) | ((uint32_t)(f.mvy + 32768) << 16) | (uint32_t)(f.mvx + 32768));
            }
            else
                fb[(size_t)jid * fb_stride + it.slot] = f;
        }
    }
#ifdef XA_PROFILE
    if (tid == 0)
    {
        const long long t_end = XA_TICK();
        atomicAdd(&g_hier_prof[0], (unsigned long long)(t_end - t_begin));
        atomicAdd(&g_hier_prof[1], (unsigned long long)(t_staged - t_begin));
        atomicAdd(&g_hier_prof[2], (unsigned long long)(t_looped - t_staged));
        atomicAdd(&g_hier_prof[3], (unsigned long long)(t_end - t_looped));
        atomicAdd(&g_hier_prof[4], 1ull);
        if (blockIdx.x < 16384)
        {
            g_hier_trace[3 * blockIdx.x] = r_begin;
            g_hier_trace[3 * blockIdx.x + 1] = r_staged;
            g_hier_trace[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        }
    }
#endif
}

// KEYS = false: the work items of a planned list, results in one FullBest slot per (job, slab).  KEYS = true: bundles
// formed on the device - `items` is the offsets array described above, n_items the number of cells, a job's slabs meet in
// one 64-bit key, and fb_stride = -(LDS size class of this launch, plan_lds_class): every class has a launch of its own.
template <bool KEYS>
__global__ void __launch_bounds__(HIER_THREADS) __attribute__((amdgpu_waves_per_eu(HIER_WAVES_PER_SIMD, HIER_WAVES_PER_SIMD))) k_full_hier(BatchParams P, const x265amd_me_job *jobs, const HierBundle *bundles,
                                                            const HierItem *items, int n_items, const uint32_t *packed,
                                                            FullBest *fb, int fb_stride)
{
    extern __shared__ __align__(16) uint32_t lds[];
    if (KEYS)
    {
        const int *offsets = (const int*)items;
        const int id = plan_item_of_block(blockIdx.x, offsets[n_items]);
        if (id < 0)
            return;
        int lo = 0, hi = n_items;               // the cell with offsets[cell] <= id < offsets[cell + 1]
        while (hi - lo > 1)
        {
            const int mid = (lo + hi) >> 1;
            if (offsets[mid] <= id) lo = mid; else hi = mid;
        }
        const HierBundle &B = bundles[lo];
        const int cls = -fb_stride;             // this launch's LDS class
        if (B.lds_need > plan_lds_class(cls) || (cls > 0 && B.lds_need <= plan_lds_class(cls - 1)))
            return;
        const int XG = (B.nxw + 3) >> 2, ntasks = ((B.nyw + HIER_R - 1) / HIER_R) * XG;
        HierItem it;
        it.bundle = lo;
        it.t_lo = (id - offsets[lo]) * B.per;
        it.t_hi = it.t_lo + B.per < ntasks ? it.t_lo + B.per : ntasks;
        it.slot = 0;
        it.pitch = B.pitch;
        it.win_nd = XG + 16;
        it.win_rows = ((it.t_hi - 1) / XG - it.t_lo / XG + 1) * HIER_R + 63;
        it.pad = 0;
        const XA_CONST uint32_t *pf = (const XA_CONST uint32_t*)(uintptr_t)(packed + (size_t)lo * HIER_FENC_DW);
        if (B.pred == 1)
            hier_item_body<2, true>(P, jobs, B, it, pf, fb, 1, lds);
        else if (B.pred)
            hier_item_body<3, true>(P, jobs, B, it, pf, fb, 1, lds);
        else if (B.uniform)
            hier_item_body<0, true>(P, jobs, B, it, pf, fb, 1, lds);
        else
            hier_item_body<1, true>(P, jobs, B, it, pf, fb, 1, lds);
        return;
    }
    const int id = blockIdx.x;
    if (id >= n_items)
        return;
    const HierItem &it = items[id];
    const HierBundle &B = bundles[it.bundle];
    const XA_CONST uint32_t *pf = (const XA_CONST uint32_t*)(uintptr_t)(packed + (size_t)it.bundle * HIER_FENC_DW);
    if (B.pred == 1)
        hier_item_body<2, false>(P, jobs, B, it, pf, fb, fb_stride, lds);
    else if (B.pred)
        hier_item_body<3, false>(P, jobs, B, it, pf, fb, fb_stride, lds);
    else if (B.uniform)
        hier_item_body<0, false>(P, jobs, B, it, pf, fb, fb_stride, lds);
    else
        hier_item_body<1, false>(P, jobs, B, it, pf, fb, fb_stride, lds);
}

} // namespace xa
