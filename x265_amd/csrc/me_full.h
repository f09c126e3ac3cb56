// Exhaustive (FULL_SEARCH) integer stage (SAD metric), driven by WORK ITEMS.  The description below
// is the 8-bit form; 16-bit planes use the same items, windows and lane tasks with two pixels per
// dword and v_sad_u16 (scan_job16).
//
// A work item is a set of jobs whose search windows overlap (PUs of one size inside one cell of
// the picture) plus a range of candidate rows.  One 256-thread workgroup executes one item:
//   1. the union of the jobs' reference windows is staged into LDS once (row-coalesced dword reads);
//   2. small PUs (>= 4 jobs in the item): every wavefront takes whole jobs ("wave mode");
//      large PUs: all four wavefronts split the candidates of one job ("block mode"); a large
//      job is cut into several items by candidate rows (slabs) so that the machine stays full;
//   3. a lane task = 4 horizontally consecutive candidates x 2 consecutive candidate rows.  Per
//      window row the lane holds W4+1 consecutive dwords and issues v_qsad_pk_u16_u8 (4
//      byte-shifted 4-pixel SADs per instruction -- no byte realignment) against the matching
//      encoder-block dword, read as an LDS broadcast.  Every window dword feeds both candidate
//      rows; the next row's operands are loaded while the current row's SADs issue;
//   4. packed 16-bit partial sums are flushed to 32 bits every <= 256 pixels (cannot overflow);
//   5. winner = min over (cost << 32 | raster index): the candidate that the reference's
//      strict-`<` raster scan keeps (reference: encoder/motion.cpp:738-771).
// The window pitch P is chosen with 2*P == (tasks per row pair) mod 32 so that the 32 lanes of a
// half-wave always read 32 different LDS banks.
#pragma once

#include "me_search.h"

namespace xa {

#ifdef XA_PROFILE
static __device__ unsigned long long g_full_prof[8];     // developer aid: clocks {workgroup total, staging, job staging, scan}, items
#define XA_TICK() ((long long)__builtin_readcyclecounter())
#endif
constexpr int FULL_THREADS = 256;
constexpr int FULL_WAVES = FULL_THREADS / 64;
constexpr int FB_STRIDE_PLAN = 16;        // FullBest slots per job in a planned job list (max slabs)

struct FullItem
{
    int32_t first;        // index of the item's first entry in the item-job index array
    int32_t count;        // jobs in the item (all of one size and one reference plane)
    int32_t y_lo, y_hi;   // candidate-row range [y_lo, y_hi) relative to each job's mvmin_y
    int32_t slot;         // FullBest slot (slab number) the results go to
    int32_t win_x0, win_y0;   // picture coordinates of the staged window's first pixel
    int32_t win_rows;     // rows that must be staged for [y_lo, y_hi) (0: derive per tile)
    int32_t win_nd;       // dwords staged per row
    int32_t pitch;        // LDS row pitch in dwords
};

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        unsigned long long o = __shfl_xor(v, m);
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long lds_pair(const uint32_t *p)
{
    unsigned long long v;
    __builtin_memcpy(&v, p, 8);        // two consecutive dwords, 4-byte aligned (ds_read2_b32)
    return v;
}

// Scan the candidates [y_lo, y_hi) x [0, nx) of one job.  `win` addresses the LDS dword of
// candidate (0, y_lo) pixel (0,0); tasks are dealt to (id, id + stride, ...).
// Window row r feeds candidate row A (= y0) with encoder row r and candidate row B (= y0+1) with
// encoder row r-1.  The 64-bit qsad operand for byte offset 4k is read from LDS as the dword pair
// (k, k+1) directly: every window dword is fetched twice, but no register shuffling is needed.
template <int W4>
__device__ __forceinline__ unsigned long long scan_job(const uint32_t *win, int P, const uint32_t *fenc,
                                                       const uint16_t *cx, const uint16_t *cy,
                                                       int h, int nx, int y_lo, int y_hi, int id, int stride)
{
    const int XG = (nx + 3) >> 2, XGe = XG + (XG & 1);
    const int npairs = (y_hi - y_lo + 1) >> 1;
    const int ntasks = npairs * XGe;
    const int rows_per_flush = 256 / (4 * W4);
    unsigned long long best = ~0ull;

    for (int task = id; task < ntasks; task += stride)
    {
        const int yp = task / XGe, xg = task - yp * XGe;
        if (xg >= XG)
            continue;
        const uint32_t *row = win + (2 * yp) * P + xg;
        unsigned long long pa = 0, pb = 0;
        uint32_t sa[4] = { 0, 0, 0, 0 }, sb[4] = { 0, 0, 0, 0 };
        uint32_t fA[W4], fB[W4];
        unsigned long long d[W4];
        int since = 0;

#define XA_LOAD_WIN(R)  do { _Pragma("unroll") for (int k = 0; k < W4; k++) d[k] = lds_pair(row + (R) * P + k); } while (0)
#define XA_LOAD_ENC(F, R) do { _Pragma("unroll") for (int k = 0; k < W4; k++) F[k] = fenc[(R) * W4 + k]; } while (0)
#define XA_QSAD(ACC, F) do { _Pragma("unroll") for (int k = 0; k < W4; k++) ACC = __builtin_amdgcn_qsad_pk_u16_u8(d[k], F[k], ACC); } while (0)
#define XA_FLUSH()                                                     \
        do {                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) {            \
                sa[i] += (uint32_t)(pa >> (16 * i)) & 0xffff;          \
                sb[i] += (uint32_t)(pb >> (16 * i)) & 0xffff; }        \
            pa = pb = 0; since = 0;                                    \
        } while (0)

        // rows are taken two at a time so that the encoder rows ping-pong between fA and fB
        XA_LOAD_WIN(0);
        XA_LOAD_ENC(fA, 0);
        XA_QSAD(pa, fA);
        since = 1;
        for (int r = 1; r + 1 < h; r += 2)       // h is a multiple of 4: ends with r == h - 1 left over
        {
            XA_LOAD_WIN(r);
            XA_LOAD_ENC(fB, r);
            XA_QSAD(pb, fA);
            XA_QSAD(pa, fB);
            if (++since >= rows_per_flush) XA_FLUSH();
            XA_LOAD_WIN(r + 1);
            XA_LOAD_ENC(fA, r + 1);
            XA_QSAD(pb, fB);
            XA_QSAD(pa, fA);
            if (++since >= rows_per_flush) XA_FLUSH();
        }
        XA_LOAD_WIN(h - 1);
        XA_LOAD_ENC(fB, h - 1);
        XA_QSAD(pb, fA);
        XA_QSAD(pa, fB);
        if (++since >= rows_per_flush) XA_FLUSH();
        XA_LOAD_WIN(h);
        XA_QSAD(pb, fB);
        XA_FLUSH();
#undef XA_LOAD_WIN
#undef XA_LOAD_ENC
#undef XA_QSAD
#undef XA_FLUSH

        const int iy = y_lo + 2 * yp;
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
        {
            const int y = iy + rr;
            if (y < y_hi)
            {
                const int cyv = cy[y];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int x = 4 * xg + i;
                    if (x < nx)
                    {
                        uint32_t c = (rr ? sb[i] : sa[i]) + (uint32_t)((cx[x] + cyv) & 0xffff);
                        unsigned long long key = ((unsigned long long)c << 32) | (uint32_t)(y * nx + x);
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// Tiny PUs (at most 16 dwords in 4 or 8 rows: 8x8, 8x4, 4x8, 4x4, 12x4, 16x4; a 4x16 PU takes the general scan).  The whole encoder block sits in scalar
// registers, a lane task is 4 x-candidates x 4 candidate rows (every window dword pair feeds four
// candidate rows), the row loop is fully unrolled, and no flush is needed (<= 64 pixels per
// candidate).  The winner is tracked as ONE 32-bit key (cost << 15 | raster index): the cost of a
// tiny PU is below 2^17 (64 x 255 + 65535) and the caller guarantees nx * ny <= 2^15.
// Candidates beyond nx / y_hi are switched off by OR-ing all-ones into their key.
constexpr int TINY_R = 4;
constexpr int TINY_KEY_SHIFT = 15;

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
    {
        uint32_t o = (uint32_t)__shfl_xor((int)v, m);
        v = o < v ? o : v;
    }
    return v;
}

template <int W4, int H>
__device__ __forceinline__ uint32_t scan_tiny(const uint32_t *win, int P, const uint32_t *fe /* H*W4 uniform dwords */,
                                              const uint16_t *cx, const uint16_t *cy,
                                              int nx, int y_lo, int y_hi, int id, int stride)
{
    constexpr int R = TINY_R;
    typedef unsigned short ushort2v __attribute__((ext_vector_type(2)));
    const int XG = (nx + 3) >> 2, XGe = XG + (XG & 1);
    const int nquads = (y_hi - y_lo + R - 1) / R;
    const int ntasks = nquads * XGe;
    uint32_t best = ~0u;
    for (int task = id; task < ntasks; task += stride)
    {
        const int yq = task / XGe, xg = task - yq * XGe;
        if (xg >= XG)
            continue;
        const uint32_t *row = win + (R * yq) * P + xg;
        unsigned long long p[R];
#pragma unroll
        for (int i = 0; i < R; i++)
            p[i] = 0;
#pragma unroll
        for (int r = 0; r < H + R - 1; r++)
        {
            unsigned long long d[W4];
#pragma unroll
            for (int k = 0; k < W4; k++)
                d[k] = lds_pair(row + r * P + k);
#pragma unroll
            for (int i = 0; i < R; i++)
            {
                const int e = r - i;                    // encoder row that window row r meets in candidate row i
                if (e >= 0 && e < H)
                {
#pragma unroll
                    for (int k = 0; k < W4; k++)
                        p[i] = __builtin_amdgcn_qsad_pk_u16_u8(d[k], fe[e * W4 + k], p[i]);
                }
            }
        }
        // MV costs of the 4 x-candidates as two packed pairs; per-candidate kill masks OR-ed with x
        const uint32_t *cxp = (const uint32_t*)(cx + 4 * xg);
        const uint32_t cx01 = cxp[0], cx23 = cxp[1];
        const int nvalid = nx - 4 * xg;
        uint32_t kx[4];
#pragma unroll
        for (int x = 0; x < 4; x++)
            kx[x] = (x < nvalid ? 0u : ~0u) | (uint32_t)x;
        const int y0 = y_lo + R * yq;
#pragma unroll
        for (int i = 0; i < R; i++)
        {
            const int y = y0 + i;
            const uint32_t cyv = cy[y < y_hi ? y : y_lo];
            ushort2v a01, a23, c2 = { (unsigned short)cyv, (unsigned short)cyv };
            __builtin_memcpy(&a01, &cx01, 4);
            __builtin_memcpy(&a23, &cx23, 4);
            a01 += c2;                                  // uint16 wrap-around = the reference's uint16_t mvcost sum
            a23 += c2;
            const uint32_t plo = (uint32_t)p[i], phi = (uint32_t)(p[i] >> 32);
            const uint32_t c0 = (plo & 0xffff) + a01.x, c1 = (plo >> 16) + a01.y;
            const uint32_t c2v = (phi & 0xffff) + a23.x, c3 = (phi >> 16) + a23.y;
            uint32_t m = (c0 << 2) | kx[0];
            uint32_t t1 = (c1 << 2) | kx[1], t2 = (c2v << 2) | kx[2], t3 = (c3 << 2) | kx[3];
            m = t1 < m ? t1 : m;
            t2 = t3 < t2 ? t3 : t2;
            m = t2 < m ? t2 : m;
            // (cost << 2 | x)  ->  (cost << 15 | y * nx + 4 xg + x)
            uint32_t key = ((m >> 2) << TINY_KEY_SHIFT) + (uint32_t)(y * nx + 4 * xg) + (m & 3);
            key = y < y_hi ? key : ~0u;
            best = key < best ? key : best;
        }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------
// 16-bit pixels: a dword holds 2 pixels, there is no byte-shifted packed SAD, so the lane keeps
// each window dword d[i] and its one-pixel-shifted neighbour o[i] = (d[i+1]:d[i]) >> 16 and
// issues v_sad_u16 (2 pixels per lane-op): candidate x uses d[k], o[k], d[k+1], o[k+1] against
// encoder dword k.  Same lane task (4 x-candidates x 2 candidate rows), same item / window /
// winner-key scheme as the 8-bit scan; the row is processed in chunks of 8 dwords to bound
// registers.  `win` / `P` / `fenc` are in dwords; W4 = PU width / 4 (2 * W4 dwords per row).
template <int W4>
__device__ __forceinline__ unsigned long long scan_job16(const uint32_t *win, int P, const uint32_t *fenc,
                                                         const uint16_t *cx, const uint16_t *cy,
                                                         int h, int nx, int y_lo, int y_hi, int id, int stride)
{
    constexpr int W2 = 2 * W4;                    // dwords per encoder row
    constexpr int CH = W2 % 8 == 0 ? 8 : W2 % 6 == 0 ? 6 : W2 % 4 == 0 ? 4 : 2;     // dwords per chunk (divides W2)
    const int XG = (nx + 3) >> 2, XGe = XG + (XG & 1);
    const int npairs = (y_hi - y_lo + 1) >> 1;
    const int ntasks = npairs * XGe;
    unsigned long long best = ~0ull;

    for (int task = id; task < ntasks; task += stride)
    {
        const int yp = task / XGe, xg = task - yp * XGe;
        if (xg >= XG)
            continue;
        const uint32_t *row = win + (2 * yp) * P + 2 * xg;
        uint32_t sa[4] = { 0, 0, 0, 0 }, sb[4] = { 0, 0, 0, 0 };
        // window row r meets encoder row r in candidate row A and encoder row r - 1 in candidate row B
        for (int r = 0; r <= h; r++)
        {
            const uint32_t *wr = row + r * P;
            const uint32_t *fa = fenc + (r < h ? r : 0) * W2, *fb = fenc + (r > 0 ? r - 1 : 0) * W2;
            const bool doA = r < h, doB = r > 0;
#pragma unroll
            for (int c0 = 0; c0 < W2; c0 += CH)
            {
                uint32_t d[CH + 2], o[CH + 1];
#pragma unroll
                for (int i = 0; i < CH + 2; i += 2)
                {
                    const unsigned long long v = lds_pair(wr + c0 + i);
                    d[i] = (uint32_t)v;
                    d[i + 1] = (uint32_t)(v >> 32);
                }
#pragma unroll
                for (int i = 0; i <= CH; i++)
                    o[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], 2);
                if (doA)
                {
#pragma unroll
                    for (int i = 0; i < CH; i++)
                    {
                        const uint32_t f = fa[c0 + i];
                        sa[0] = __builtin_amdgcn_sad_u16(d[i], f, sa[0]);
                        sa[1] = __builtin_amdgcn_sad_u16(o[i], f, sa[1]);
                        sa[2] = __builtin_amdgcn_sad_u16(d[i + 1], f, sa[2]);
                        sa[3] = __builtin_amdgcn_sad_u16(o[i + 1], f, sa[3]);
                    }
                }
                if (doB)
                {
#pragma unroll
                    for (int i = 0; i < CH; i++)
                    {
                        const uint32_t f = fb[c0 + i];
                        sb[0] = __builtin_amdgcn_sad_u16(d[i], f, sb[0]);
                        sb[1] = __builtin_amdgcn_sad_u16(o[i], f, sb[1]);
                        sb[2] = __builtin_amdgcn_sad_u16(d[i + 1], f, sb[2]);
                        sb[3] = __builtin_amdgcn_sad_u16(o[i + 1], f, sb[3]);
                    }
                }
            }
        }
        const int iy = y_lo + 2 * yp;
#pragma unroll
        for (int rr = 0; rr < 2; rr++)
        {
            const int y = iy + rr;
            if (y < y_hi)
            {
                const int cyv = cy[y];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int x = 4 * xg + i;
                    if (x < nx)
                    {
                        uint32_t c = (rr ? sb[i] : sa[i]) + (uint32_t)((cx[x] + cyv) & 0xffff);
                        unsigned long long key = ((unsigned long long)c << 32) | (uint32_t)(y * nx + x);
                        best = key < best ? key : best;
                    }
                }
            }
        }
    }
    return best;
}

__device__ __forceinline__ void write_best(FullBest *out, unsigned long long best, const x265amd_me_job &j, int nx)
{
    FullBest f;
    if (best == ~0ull)
    {
        f.cost = 0x7fffffff;
        f.mvx = f.mvy = 0;
    }
    else
    {
        int idx = (int)(best & 0xffffffffu);
        f.cost = (int)(best >> 32);
        f.mvx = (int16_t)(j.mvmin_x + idx % nx);
        f.mvy = (int16_t)(j.mvmin_y + idx / nx);
    }
    *out = f;
}

// stage one job's encoder block and MV-cost slices into `scratch`
// (layout: fenc[h * W4 * sizeof(PIX)] dwords | cx[nx] | cy[ny] as u16)
template <int W4, typename PIX>
__device__ __forceinline__ void stage_job(const PlaneDesc &enc, const uint16_t *cost, const x265amd_me_job &j,
                                          uint32_t *scratch, int nx, int ny, int id, int stride)
{
    constexpr int RD = W4 * (int)sizeof(PIX);         // dwords per encoder row
    const int h = j.h;
    const PIX *e = (const PIX*)enc.origin + (int64_t)j.pu_y * enc.stride + j.pu_x;
    for (int i = id; i < h * RD; i += stride)
        scratch[i] = ld_u32((const uint8_t*)(e + (int64_t)(i / RD) * enc.stride) + 4 * (i % RD));
    uint16_t *cx = (uint16_t*)(scratch + h * RD), *cy = cx + nx;
    for (int i = id; i < nx; i += stride)
        cx[i] = cost[(int16_t)((int16_t)((j.mvmin_x + i) << 2) - j.qmvp_x)];
    for (int i = id; i < ny; i += stride)
        cy[i] = cost[(int16_t)((int16_t)((j.mvmin_y + i) << 2) - j.qmvp_y)];
}

template <int W4, typename PIX>
__device__ void full_item_body(const BatchParams &P, const x265amd_me_job *jobs, const int *item_jobs,
                               const FullItem &it, FullBest *fb, int fb_stride, uint32_t *lds, int lds_dwords)
{
    constexpr int PB = (int)sizeof(PIX);          // bytes per pixel = dwords per 4 pixels
    constexpr int PPD = 4 / PB;                   // pixels per dword
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const PlaneDesc &enc = P.planes[P.enc_plane];
    const x265amd_me_job &j0 = jobs[item_jobs[it.first]];
    const PlaneDesc &ref = P.planes[P.ref_planes[j0.ref_idx]];
    const int h = j0.h;
    const int pitch = it.pitch, ND = it.win_nd;
    const bool wave_mode = it.count >= FULL_WAVES;

    // scratch: per wave (wave mode) or one for the block
    const int scratch_dw = h * W4 * PB + ((2 * FULL_MAX_CAND + 1) >> 1) + 1;
    const int n_scratch = wave_mode ? FULL_WAVES : 1;
    uint32_t *s_scratch = lds;
    uint32_t *s_win = lds + n_scratch * scratch_dw;
    const int rows_fit = (lds_dwords - n_scratch * scratch_dw) / pitch;
    int TH = (rows_fit - h) & ~1;                 // candidate rows per tile
    if (TH < 2)
        return;                                   // cannot happen for planned items
    const uint8_t *r0 = (const uint8_t*)ref.origin;
    const int64_t lo = ref.lo, hi = ref.hi - PPD;     // element offsets a staged dword may start at

    unsigned long long blk_best = ~0ull;          // block mode accumulates over tiles
#ifdef XA_PROFILE
    const long long t_begin = XA_TICK();
    const long long r_begin = (long long)__builtin_amdgcn_s_memrealtime();      // constant 100 MHz counter
    long long t_stage = 0, t_job = 0, t_scan = 0;
#endif
    for (int ty0 = it.y_lo; ty0 < it.y_hi; ty0 += TH)
    {
#ifdef XA_PROFILE
        const long long t_a = XA_TICK();
#endif
        const int ty1 = (it.y_hi - ty0 < TH) ? it.y_hi : ty0 + TH;
        const int rows = it.win_rows ? it.win_rows : (ty1 - ty0) + h - 1 + ((ty1 - ty0) & 1);
        __syncthreads();
        for (int r = wv; r < rows; r += FULL_WAVES)
        {
            int64_t o = (int64_t)(it.win_y0 + (it.win_rows ? 0 : ty0 - it.y_lo) + r) * ref.stride + it.win_x0;
            for (int c = lane; c < ND; c += 64)
            {
                int64_t a = o + PPD * c;
                a = a < lo ? lo : (a > hi ? hi : a);
                s_win[r * pitch + c] = ld_u32(r0 + a * PB);
            }
        }
        __syncthreads();
#ifdef XA_PROFILE
        t_stage += XA_TICK() - t_a;
#endif

        if (wave_mode)
        {
            uint32_t *scr = s_scratch + wv * scratch_dw;
            for (int m = wv; m < it.count; m += FULL_WAVES)
            {
                const int jid = item_jobs[it.first + m];
                const x265amd_me_job &j = jobs[jid];
                const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
                stage_job<W4, PIX>(enc, P.cost[j.qp], j, scr, nx, ny, lane, 64);
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_wave_barrier();
                const int y_hi = ty1 < ny ? ty1 : ny;
                const int wx = ((j.pu_x + j.mvmin_x - it.win_x0) >> 2) * PB;
                const int wy = j.pu_y + j.mvmin_y + ty0 - (it.win_y0 + (it.win_rows ? 0 : ty0 - it.y_lo));
                const uint16_t *cx = (const uint16_t*)(scr + h * W4 * PB);
                unsigned long long b;
                if (PB == 1 && W4 <= 4 && (h == 4 || h == 8) && h * W4 <= 16 && it.win_rows && nx * ny <= (1 << TINY_KEY_SHIFT))
                {
                    // tiny PU: encoder block into scalar registers (dword i from lane i)
                    constexpr int TW4 = W4 <= 4 ? W4 : 1;
                    const uint32_t mine = scr[lane < h * W4 ? lane : 0];
                    uint32_t fe[16];
#pragma unroll
                    for (int i = 0; i < 16; i++)
                        fe[i] = (uint32_t)__builtin_amdgcn_readlane((int)mine, i);
                    uint32_t b32 = h == 8 ? scan_tiny<TW4, (TW4 <= 2 ? 8 : 4)>(s_win + wy * pitch + wx, pitch, fe, cx, cx + nx, nx, ty0, y_hi, lane, 64)
                                          : scan_tiny<TW4, 4>(s_win + wy * pitch + wx, pitch, fe, cx, cx + nx, nx, ty0, y_hi, lane, 64);
                    b32 = wave_min_u32(b32);
                    b = b32 == ~0u ? ~0ull
                                   : ((unsigned long long)(b32 >> TINY_KEY_SHIFT) << 32) | (b32 & ((1u << TINY_KEY_SHIFT) - 1));
                }
                else
                {
                    b = PB == 1 ? scan_job<W4>(s_win + wy * pitch + wx, pitch, scr, cx, cx + nx, h, nx, ty0, y_hi, lane, 64)
                                : scan_job16<W4>(s_win + wy * pitch + wx, pitch, scr, cx, cx + nx, h, nx, ty0, y_hi, lane, 64);
                    b = wave_min_u64(b);
                }
                // a planned wave-mode item covers all candidate rows in one tile
                if (lane == 0)
                    write_best(fb + (size_t)jid * fb_stride + it.slot, b, j, nx);
                __builtin_amdgcn_wave_barrier();
            }
        }
        else
        {
            for (int m = 0; m < it.count; m++)
            {
                const int jid = item_jobs[it.first + m];
                const x265amd_me_job &j = jobs[jid];
                const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
                __syncthreads();
#ifdef XA_PROFILE
                const long long t_b = XA_TICK();
#endif
                stage_job<W4, PIX>(enc, P.cost[j.qp], j, s_scratch, nx, ny, tid, FULL_THREADS);
                __syncthreads();
#ifdef XA_PROFILE
                const long long t_c = XA_TICK();
                t_job += t_c - t_b;
#endif
                const int y_hi = ty1 < ny ? ty1 : ny;
                const int wx = ((j.pu_x + j.mvmin_x - it.win_x0) >> 2) * PB;
                const int wy = j.pu_y + j.mvmin_y + ty0 - (it.win_y0 + (it.win_rows ? 0 : ty0 - it.y_lo));
                const uint16_t *cx = (const uint16_t*)(s_scratch + h * W4 * PB);
                unsigned long long b = PB == 1
                    ? scan_job<W4>(s_win + wy * pitch + wx, pitch, s_scratch, cx, cx + nx, h, nx, ty0, y_hi, tid, FULL_THREADS)
                    : scan_job16<W4>(s_win + wy * pitch + wx, pitch, s_scratch, cx, cx + nx, h, nx, ty0, y_hi, tid, FULL_THREADS);
#ifdef XA_PROFILE
                t_scan += XA_TICK() - t_c;
#endif
                b = wave_min_u64(b);
                if (it.count == 1)
                    blk_best = b < blk_best ? b : blk_best;      // reduce across waves after the last tile
                else
                {
                    // several block-mode jobs share the item (always a single tile): reduce now
                    __syncthreads();
                    unsigned long long *red = (unsigned long long*)s_scratch;
                    if (lane == 0) red[wv] = b;
                    __syncthreads();
                    if (tid == 0)
                    {
                        for (int i = 1; i < FULL_WAVES; i++) b = red[i] < b ? red[i] : b;
                        write_best(fb + (size_t)jid * fb_stride + it.slot, b, j, nx);
                    }
                }
            }
        }
    }
#ifdef XA_PROFILE
    if (tid == 0 && (blockIdx.x & 15) == 0)
    {
        atomicAdd(&g_full_prof[0], (unsigned long long)(XA_TICK() - t_begin));
        atomicAdd(&g_full_prof[1], (unsigned long long)t_stage);
        atomicAdd(&g_full_prof[2], (unsigned long long)t_job);
        atomicAdd(&g_full_prof[3], (unsigned long long)t_scan);
        atomicAdd(&g_full_prof[4], 1ull);
        atomicAdd(&g_full_prof[7], (unsigned long long)((long long)__builtin_amdgcn_s_memrealtime() - r_begin));
    }

#endif
    if (!wave_mode && it.count == 1)
    {
        const int jid = item_jobs[it.first];
        const x265amd_me_job &j = jobs[jid];
        __syncthreads();
        unsigned long long *red = (unsigned long long*)s_scratch;
        if (lane == 0) red[wv] = blk_best;
        __syncthreads();
        if (tid == 0)
        {
            unsigned long long b = red[0];
            for (int i = 1; i < FULL_WAVES; i++) b = red[i] < b ? red[i] : b;
            write_best(fb + (size_t)jid * fb_stride + it.slot, b, j, j.mvmax_x - j.mvmin_x + 1);
        }
    }
}

// One instantiation per PU width so that narrow PUs keep a small register footprint (occupancy).
// W4 = 0 selects by the item's job width at run time (unplanned path).
template <int W4, typename PIX>
__global__ void __launch_bounds__(FULL_THREADS) k_full_items(BatchParams P, const x265amd_me_job *jobs,
                                                             const int *item_jobs, const FullItem *items, int n_items,
                                                             FullBest *fb, int fb_stride, int lds_dwords)
{
    extern __shared__ __align__(16) uint32_t lds[];
    if (W4 != 0)
    {
        const int id = blockIdx.x;
        if (id >= n_items)
            return;
        const FullItem &it = items[id];
        if (it.count <= 0)
            return;
        full_item_body<W4 ? W4 : 1, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords);
        return;
    }
    // (one item per job, most of them possibly empty: a bounded number of workgroups takes a stretch of the list each, so that
    // an empty item costs a lane of one load and not a workgroup with the full LDS allocation; every wavefront looks at the
    // stretch's items 256 at a time - four per lane - and all of them walk the non-empty ones together)
    const int per = (n_items + (int)gridDim.x - 1) / (int)gridDim.x;
    const int end = min(n_items, ((int)blockIdx.x + 1) * per), lane = threadIdx.x & 63;
    for (int c0 = (int)blockIdx.x * per; c0 < end; c0 += 256)
    {
        unsigned long long mask[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int id = c0 + 64 * r + lane;
            mask[r] = __ballot(id < end && items[id].count > 0);
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
            while (mask[r])
            {
                const int id = c0 + 64 * r + __builtin_ctzll(mask[r]);
                mask[r] &= mask[r] - 1;
                const FullItem &it = items[id];
                const x265amd_me_job &j0 = jobs[item_jobs[it.first]];
                switch (j0.w >> 2)
                {
                case 1:  full_item_body<1, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 2:  full_item_body<2, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 3:  full_item_body<3, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 4:  full_item_body<4, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 6:  full_item_body<6, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 8:  full_item_body<8, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 12: full_item_body<12, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                case 16: full_item_body<16, PIX>(P, jobs, item_jobs, it, fb, fb_stride, lds, lds_dwords); break;
                default: break;
                }
                __syncthreads();        // (the next item reuses the LDS)
            }
    }
}

// One item per job, built on the device: used when the job array only exists in device memory
// and no plan was made for it.  Ineligible jobs get an empty item.
// `skip` (optional): jobs whose scan the hierarchical kernel has run for bundles formed on the device (me_plan.h); their
// slot of `keys` holds the winner as cost << 32 | (mvy + 32768) << 16 | (mvx + 32768) and becomes a FullBest here.
__global__ void k_make_items(BatchParams P, const x265amd_me_job *jobs, int n, int *item_jobs, FullItem *items, const unsigned char *skip,
                             unsigned long long *keys)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n)
        return;
    if (skip && skip[id])
    {
        const unsigned long long k = keys[id];
        FullBest f;
        f.cost = k == ~0ull ? 0x7fffffff : (int)(k >> 32);
        f.mvx = k == ~0ull ? 0 : (int16_t)((int)(k & 0xffff) - 32768);
        f.mvy = k == ~0ull ? 0 : (int16_t)((int)((k >> 16) & 0xffff) - 32768);
        ((FullBest*)keys)[id] = f;
    }
    const x265amd_me_job &j = jobs[id];
    FullItem it;
    it.first = id;
    item_jobs[id] = id;
    it.count = job_fields_valid(j, P.n_refs) && full_kernel_eligible(j, P.planes[P.enc_plane].bpp) && !(skip && skip[id]) ? 1 : 0;
    const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    const int XG = (nx + 3) >> 2, XGe = XG + (XG & 1), W4 = j.w >> 2;
    const int ND = (XGe + W4) * P.planes[P.enc_plane].bpp;      // dwords per window row
    it.y_lo = 0;
    it.y_hi = ny;
    it.slot = 0;
    it.win_x0 = j.pu_x + j.mvmin_x;
    it.win_y0 = j.pu_y + j.mvmin_y;
    it.win_rows = 0;                               // staged per tile
    it.win_nd = ND;
    it.pitch = P.planes[P.enc_plane].bpp == 1 ? ND + ((((XGe >> 1) - ND) % 16) + 16) % 16 : ND + 2;
    items[id] = it;
}

} // namespace xa
