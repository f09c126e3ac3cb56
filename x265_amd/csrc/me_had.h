// Exhaustive (FULL_SEARCH) integer stage with the SA8D metric for all square PUs of a 64x64 picture cell
// from one pass over the search window (8- and 16-bit pixels up to 10 bits deep), and the same pass with
// the SAD metric: 16-bit planes up to 12 bits deep (v_sad_u16 on the same packed pairs, two candidates per
// lane), and 8-bit planes (v_qsad_pk_u16_u8, four candidates per lane) for what the quad-SAD kernel of
// me_hier.h does not take: rectangular partitions and cells with a single PU.
//
// sa8d_inter of a square PU is built from 8x8 Hadamard sums S8 (common/pixel.cpp:276-362, 756-826):
//   8x8           (S8 + 2) >> 2
//   16x16         (S8a + S8b + S8c + S8d + 2) >> 2
//   32x32, 64x64  the sum of their 16x16 values
// so, like SAD, every square of the quadtree follows from the 8x8 cells, measured once per candidate.
// Bundles, slabs and the partial-winner hand-over are those of me_hier.h; the work differs:
//
// A lane owns ONE candidate at a time and computes the whole 8x8 transform of a cell in its registers
// on packed 16-bit pairs: the cell's 8 reference rows (two horizontally adjacent pixels per register)
// minus the encoder rows (scalar operands, widened to 16 bits by k_pack_fenc16), three vertical
// butterfly stages and two horizontal ones as v_pk_add_i16 / v_pk_sub_i16 (|values| <= 32 x 1023, inside
// int16 for depths up to 10 bits), and the last horizontal stage folded into the sum of magnitudes by
// |a + b| + |a - b| = 2 max(|a|, |b|).  No cross-lane traffic until a PU is complete; then its key
// ((cost << 11) | candidate index inside the slab for 8x8 / 16x16, 64-bit above) is min-reduced across the wavefront.
//
// PUs with 4- or 12-pixel sides (8x4, 4x8 and the asymmetric partitions of a 16x16 CU: 16x4, 16x12, 4x16, 12x16) are
// priced by sa8d_inter with the SATD recipe: sums of 4x4 Hadamard tiles, each tile (or 8x4 pair of tiles) halved
// (common/pixel.cpp:187-274, table :756-826), which is exactly additive over 4x4 tiles because a tile's sum of
// magnitudes is even.  The 8x8 transform factors as H8 = H2 (x) H4: the 4x4 transforms of a cell's four quadrants
// followed by one more butterfly stage per direction.  Bundles with such PUs (HierBundle::quads) run the transform in
// that order and take the four quadrant sums on the way; halves of a cell and strips of a 16x16 block follow from them.
#pragma once

#include "me_hier.h"

#include <type_traits>

namespace xa {

constexpr int HAD_THREADS = 128;          // smallest workgroup of a launch (tuning aid); the default is 256 threads, or
constexpr int HAD_THREADS_BIG = 512;      // HAD_THREADS_BIG when a window exceeds 48 x 48 candidates
constexpr int HAD_WAVES = HAD_THREADS_BIG / 64;   // slot rows in LDS
constexpr int HAD_RIDX_BITS = 11;         // 32-bit keys: slab-local candidate index < 2^11 (a slab has at most 2^11 candidates), costs of 16x16 and below < 2^21
static_assert(HAD_THREADS >= HIER_MAX_RECT, "one thread per rectangular PU writes its partial winner");
constexpr int HAD_FENC_DW = 2048;         // packed encoder block: 64 cells x 8 rows x 4 dwords (8 pixels as 16-bit pairs)

// Re-pack the 64x64 encoder block of every bundle as 16-bit pixels: cell z (Z-order) -> 32 consecutive
// dwords, row r at [4 r .. 4 r + 3].
template <typename PIX>
__global__ void __launch_bounds__(256) k_pack_fenc16(BatchParams P, const HierBundle *bundles, int n, uint32_t *packed)
{
    const int b = blockIdx.x;
    if (b >= n)
        return;
    const PlaneDesc &enc = P.planes[P.enc_plane];
    const HierBundle &B = bundles[b];
    const PIX *org = (const PIX*)enc.origin;
    if (sizeof(PIX) == 1 && B.metric == X265AMD_SAD)
    {
        // 8-bit SAD bundles compare bytes: cell z -> 16 consecutive dwords, row r, half k at [2 r + k] (as k_pack_fenc)
        for (int i = threadIdx.x; i < HIER_FENC_DW; i += 256)
        {
            const int z = i >> 4, r = (i >> 1) & 7, k = i & 1;
            int cx, cy;
            hier_cell_xy(z, cx, cy);
            int64_t o = (int64_t)(B.ctu_y + cy * 8 + r) * enc.stride + B.ctu_x + cx * 8 + 4 * k;
            o = o < enc.lo ? enc.lo : (o > enc.hi - 4 ? enc.hi - 4 : o);
            packed[(size_t)b * HAD_FENC_DW + i] = ld_u32((const uint8_t*)org + o);
        }
        return;
    }
    for (int i = threadIdx.x; i < HAD_FENC_DW; i += 256)
    {
        const int z = i >> 5, r = (i >> 2) & 7, k = i & 3;
        int cx, cy;
        hier_cell_xy(z, cx, cy);
        int64_t o = (int64_t)(B.ctu_y + cy * 8 + r) * enc.stride + B.ctu_x + cx * 8 + 2 * k;
        o = o < enc.lo ? enc.lo : (o > enc.hi - 2 ? enc.hi - 2 : o);
        packed[(size_t)b * HAD_FENC_DW + i] = (uint32_t)org[o] | ((uint32_t)org[o + 1] << 16);
    }
}

typedef short had_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short had_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ had_s2 had_as_s2(uint32_t v) { had_s2 r; __builtin_memcpy(&r, &v, 4); return r; }

// lo = max(a.lo, a.hi), hi = max(a.hi, a.lo)  (unsigned 16-bit)
__device__ __forceinline__ uint32_t had_max_halves(uint32_t a)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a));
    return r;
}

// Sum of |8x8 Hadamard coefficients| / 2 of (ref - fenc): rows[r][c] = reference pixels (2 c, 2 c + 1) of
// row r as 16-bit pairs, f = the cell's 32 encoder dwords in the same layout
__device__ __forceinline__ uint32_t had8x8_half(const uint32_t (*rows)[4], const uint32_t *f)
{
    had_s2 a[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            a[r][c] = had_as_s2(rows[r][c]) - had_as_s2(f[4 * r + c]);
    // vertical: three butterfly stages over the rows, two columns per instruction
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
#pragma unroll
        for (int d = 4; d >= 1; d >>= 1)
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (!(r & d))
                {
                    const had_s2 s = a[r][c] + a[r + d][c], t = a[r][c] - a[r + d][c];
                    a[r][c] = s;
                    a[r + d][c] = t;
                }
    }
    // horizontal: distances 4 and 2 pair whole registers; distance 1 lives inside a register and is folded
    // into the magnitudes: |x + y| + |x - y| = 2 max(|x|, |y|)
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        had_s2 p0 = a[r][0] + a[r][2], p2 = a[r][0] - a[r][2], p1 = a[r][1] + a[r][3], p3 = a[r][1] - a[r][3];
        had_s2 q[4] = { p0 + p1, p0 - p1, p2 + p3, p2 - p3 };
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
            const had_s2 m = __builtin_elementwise_max(q[c], -q[c]);
            uint32_t mu;
            __builtin_memcpy(&mu, &m, 4);
            const uint32_t mx = had_max_halves(mu);
            had_u2 v, one = { 1, 0 };
            __builtin_memcpy(&v, &mx, 4);
            sum = __builtin_amdgcn_udot2(v, one, sum, false);
        }
    }
    return sum;
}

// sum over both halves of max(|lo|, |hi|) of a packed pair: half the sum of magnitudes after the folded last stage
__device__ __forceinline__ uint32_t had_fold_add(had_s2 q, uint32_t sum)
{
    const had_s2 m = __builtin_elementwise_max(q, -q);
    uint32_t mu;
    __builtin_memcpy(&mu, &m, 4);
    const uint32_t mx = had_max_halves(mu);
    had_u2 v, one = { 1, 0 };
    __builtin_memcpy(&v, &mx, 4);
    return __builtin_amdgcn_udot2(v, one, sum, false);
}

// had8x8_half with the butterfly stages reordered so that the 4x4 transforms of the cell's four quadrants exist
// on the way: t[q] = sum of |4x4 Hadamard coefficients| / 2 of quadrant q (0 top left, 1 top right, 2 bottom left,
// 3 bottom right) = satd_4x4 of that quadrant (common/pixel.cpp:187-213).  Returns the 8x8 sum / 2 as had8x8_half.
__device__ __forceinline__ uint32_t had8x8_half_quads(const uint32_t (*rows)[4], const uint32_t *f, uint32_t *t)
{
    had_s2 a[8][4];
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            a[r][c] = had_as_s2(rows[r][c]) - had_as_s2(f[4 * r + c]);
    // inside the quadrants: vertical distances 2 and 1, horizontal distance 2 pixels = registers (0,1), (2,3)
#pragma unroll
    for (int c = 0; c < 4; c++)
    {
#pragma unroll
        for (int d = 2; d >= 1; d >>= 1)
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (!(r & d))
                {
                    const had_s2 s = a[r][c] + a[r + d][c], u = a[r][c] - a[r + d][c];
                    a[r][c] = s;
                    a[r + d][c] = u;
                }
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 4; c += 2)
        {
            const had_s2 s = a[r][c] + a[r][c + 1], u = a[r][c] - a[r][c + 1];
            a[r][c] = s;
            a[r][c + 1] = u;
        }
    // (horizontal distance 1 lives inside a register and is folded into the magnitudes)
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        uint32_t sum = 0;
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int c = 0; c < 2; c++)
                sum = had_fold_add(a[4 * (q >> 1) + r][2 * (q & 1) + c], sum);
        t[q] = sum;
    }
    // across the quadrants: vertical distance 4, horizontal distance 4 pixels = registers (0,2), (1,3)
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const had_s2 s = a[r][c] + a[r + 4][c], u = a[r][c] - a[r + 4][c];
            a[r][c] = s;
            a[r + 4][c] = u;
        }
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 2; c++)
        {
            sum = had_fold_add(a[r][c] + a[r][c + 2], sum);
            sum = had_fold_add(a[r][c] - a[r][c + 2], sum);
        }
    return sum;
}

// 8 reference pixels starting at element `e` of LDS row `row` (dwords), as four 16-bit pairs
template <typename PIX>
__device__ __forceinline__ void had_load_row(const uint32_t *row, int e, uint32_t *out)
{
    if (sizeof(PIX) == 2)
    {
        const uint32_t *d = row + (e >> 1);
        const uint32_t sh = (uint32_t)(e & 1) * 2;
        uint32_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            w[k] = d[k];
#pragma unroll
        for (int k = 0; k < 4; k++)
            out[k] = __builtin_amdgcn_alignbyte(w[k + 1], w[k], sh);
    }
    else
    {
        const uint32_t *d = row + (e >> 2);
        const uint32_t sh = (uint32_t)(e & 3);
        const uint32_t w0 = d[0], w1 = d[1], w2 = d[2];
        const uint32_t b0 = __builtin_amdgcn_alignbyte(w1, w0, sh), b1 = __builtin_amdgcn_alignbyte(w2, w1, sh);
        out[0] = __builtin_amdgcn_perm(0, b0, 0x0c010c00);
        out[1] = __builtin_amdgcn_perm(0, b0, 0x0c030c02);
        out[2] = __builtin_amdgcn_perm(0, b1, 0x0c010c00);
        out[3] = __builtin_amdgcn_perm(0, b1, 0x0c030c02);
    }
}

// SAD variant (16-bit planes): a lane owns TWO horizontally adjacent candidates (x even, x + 1), which share the
// five dwords of every window row: the even one reads them as they are, the odd one shifted by one pixel.
// v_sad_u16 adds both halves' differences.
__device__ __forceinline__ void sad8x8_pair_u16(const uint32_t *wbase, int pitch, int e, const uint32_t *f, uint32_t *h)
{
    uint32_t s0 = 0, s1 = 0;
    const uint32_t *d = wbase + (e >> 1);
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        uint32_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            w[k] = d[r * pitch + k];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            s0 = __builtin_amdgcn_sad_u16(w[k], f[4 * r + k], s0);
            s1 = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbyte(w[k + 1], w[k], 2), f[4 * r + k], s1);
        }
    }
    h[0] = s0;
    h[1] = s1;
}

// SAD variant (8-bit planes): a lane owns FOUR horizontally adjacent candidates (x a multiple of 4): v_qsad_pk_u16_u8
// prices the four byte-shifted positions of a 4-pixel group at once; f = the cell's 16 encoder dwords (bytes).
__device__ __forceinline__ void sad8x8_quad_u8(const uint32_t *wbase, int pitch, int e, const uint32_t *f, uint32_t *h)
{
    unsigned long long acc = 0;
    const uint32_t *d = wbase + (e >> 2);
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        acc = __builtin_amdgcn_qsad_pk_u16_u8(lds_pair(d + r * pitch), f[2 * r], acc);
        acc = __builtin_amdgcn_qsad_pk_u16_u8(lds_pair(d + r * pitch + 1), f[2 * r + 1], acc);
    }
    h[0] = (uint32_t)acc & 0xffffu;
    h[1] = (uint32_t)acc >> 16;
    h[2] = (uint32_t)(acc >> 32) & 0xffffu;
    h[3] = (uint32_t)(acc >> 48);
}

// The SAD variants with one sum per quadrant of the cell: hq[q][c], q as in had8x8_half_quads, c = candidate
__device__ __forceinline__ void sad8x8_pair_u16_quads(const uint32_t *wbase, int pitch, int e, const uint32_t *f, uint32_t (*hq)[2])
{
    const uint32_t *d = wbase + (e >> 1);
#pragma unroll
    for (int q = 0; q < 4; q++)
        hq[q][0] = hq[q][1] = 0;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        uint32_t w[5];
#pragma unroll
        for (int k = 0; k < 5; k++)
            w[k] = d[r * pitch + k];
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int q = 2 * (r >> 2) + (k >> 1);
            hq[q][0] = __builtin_amdgcn_sad_u16(w[k], f[4 * r + k], hq[q][0]);
            hq[q][1] = __builtin_amdgcn_sad_u16(__builtin_amdgcn_alignbyte(w[k + 1], w[k], 2), f[4 * r + k], hq[q][1]);
        }
    }
}
__device__ __forceinline__ void sad8x8_quad_u8_quads(const uint32_t *wbase, int pitch, int e, const uint32_t *f, uint32_t (*hq)[4])
{
    unsigned long long acc[4] = { 0, 0, 0, 0 };
    const uint32_t *d = wbase + (e >> 2);
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        const int q = 2 * (r >> 2);
        acc[q] = __builtin_amdgcn_qsad_pk_u16_u8(lds_pair(d + r * pitch), f[2 * r], acc[q]);
        acc[q + 1] = __builtin_amdgcn_qsad_pk_u16_u8(lds_pair(d + r * pitch + 1), f[2 * r + 1], acc[q + 1]);
    }
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        hq[q][0] = (uint32_t)acc[q] & 0xffffu;
        hq[q][1] = (uint32_t)acc[q] >> 16;
        hq[q][2] = (uint32_t)(acc[q] >> 32) & 0xffffu;
        hq[q][3] = (uint32_t)(acc[q] >> 48);
    }
}

// candidates per lane task: SAD: 2 (16-bit pixels) or 4 (8-bit); SA8D: 1
__device__ __host__ constexpr int had_nc(bool sad, int bytes_per_px) { return !sad ? 1 : bytes_per_px == 2 ? 2 : 4; }

// The rectangular partitions of HEVC inside a 64x64 cell (HierBundle::rect, `kind` below) follow from a few running
// sums a lane keeps in registers:
//   16x8 / 8x16 in a 16x16 block           pairs of its four cells;
//   32x8, 32x24, 8x32, 24x32 (asymmetric)  row / column strips of cells of the 32x32 block (rounded per 8x8 for SA8D,
//                                          as sa8d_inter does for sizes that are not multiples of 16);
//   32x16 / 16x32                          pairs of the block's four 16x16 values;
//   64x16, 64x48, 16x64, 48x64, 64x32, 32x64   row / column strips of the 16x16 values of the 64x64 cell.
// kind: 0 = first strip, 1 = strips 1..3, 2 = strips 0..2, 3 = last strip (rows), 4..7 the same for columns,
// 8 = upper half, 9 = lower half, 10 = left half, 11 = right half (of the four quarters in Z-order); 12 = strips 0..1,
// 13 = strips 2..3 (rows), 14 / 15 the same for columns (the halves of the 64x64 cell).
template <int NC>
__device__ __forceinline__ void had_rect_value(int kind, const uint32_t (*S)[NC], const uint32_t (*T)[NC],
                                               const uint32_t *Q0, const uint32_t *Q1, const uint32_t *Q2, const uint32_t *Q3, uint32_t *v)
{
#pragma unroll
    for (int c = 0; c < NC; c++)
    {
        uint32_t r;
        switch (kind)
        {
        case 0:  r = S[0][c]; break;
        case 1:  r = S[1][c] + S[2][c] + S[3][c]; break;
        case 2:  r = S[0][c] + S[1][c] + S[2][c]; break;
        case 3:  r = S[3][c]; break;
        case 4:  r = T[0][c]; break;
        case 5:  r = T[1][c] + T[2][c] + T[3][c]; break;
        case 6:  r = T[0][c] + T[1][c] + T[2][c]; break;
        case 7:  r = T[3][c]; break;
        case 8:  r = Q0[c] + Q1[c]; break;
        case 9:  r = Q2[c] + Q3[c]; break;
        case 10: r = Q0[c] + Q2[c]; break;
        case 11: r = Q1[c] + Q3[c]; break;
        case 12: r = S[0][c] + S[1][c]; break;
        case 13: r = S[2][c] + S[3][c]; break;
        case 14: r = T[0][c] + T[1][c]; break;
        default: r = T[2][c] + T[3][c]; break;
        }
        v[c] = r;
    }
}

template <typename PIX, bool MASK, bool SAD, bool RECTS, bool QUADS>
__device__ __forceinline__ void had_item_body(const BatchParams &P, const x265amd_me_job *jobs, const HierBundle &B, const HierItem &it,
                                              const XA_CONST uint32_t *pf, const HierQuads *QJ, FullBest *fb, int fb_stride, uint32_t *lds)
{
    constexpr int PB = (int)sizeof(PIX), PPD = 4 / PB, NC = had_nc(SAD, PB);
    constexpr bool SAD8 = SAD && PB == 1;               // byte SADs: 16 encoder dwords per cell
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const PlaneDesc &ref = P.planes[P.ref_planes[B.ref_idx]];
    const int nxw = B.nxw, nyw = B.nyw, pitch = it.pitch;
    const int XP = (nxw + NC - 1) / NC;                 // lane tasks per candidate row
    const int y_lo = it.t_lo / XP;                      // first candidate row of this slab = first staged window row
    const int x_lo = NC * (it.t_lo - y_lo * XP);        // slab-local candidate index = (y - y_lo) * nxw + x - x_lo

    // LDS: window | cx[nxw] cy[nyw] (u16) | 32-bit slots [waves][80] | 64-bit slots [waves][5]
    // (the window area is rounded up to an even dword count, so the 64-bit slots behind it are 8-byte aligned)
    uint32_t *s_win = lds;
    // (bundles with per-job ranges or predictors: cx[4][lxp] cy[4][lyp])
    const int lxp = (B.lx + 3) & ~3, lyp = (B.ly + 3) & ~3;
    uint16_t *s_cx = (uint16_t*)(s_win + ((it.win_rows * pitch + 1) & ~1));
    uint16_t *s_cy = s_cx + (MASK ? 4 * lxp : (nxw + 3) & ~3);
    uint32_t *s_slot = (uint32_t*)(s_cy + (MASK ? 4 * lyp : (nyw + 3) & ~3));
    unsigned long long *s_big = (unsigned long long*)(s_slot + HAD_WAVES * HIER_SMALL);
    // bundles with rectangular PUs: 64-bit slots [waves][HIER_MAX_RECT]; with PUs of 4- or 12-pixel sides: 32-bit
    // slots [waves][HIER_QUAD_PUS] behind them (cell z, half k at 4 z + k; block q, partition k at 256 + 8 q + k)
    const int n_rect = RECTS ? B.rect_end[20] : 0;
    const bool quads = QUADS && B.quads != 0;
    unsigned long long *s_rk = s_big + HAD_WAVES * 5;
    uint32_t *s_qslot = (uint32_t*)(s_rk + (n_rect ? nwaves * HIER_MAX_RECT : 0));
    {
        if (RECTS && n_rect)
            for (int i = tid; i < nwaves * HIER_MAX_RECT; i += nthreads)
                s_rk[i] = ~0ull;
        if (quads)
            for (int i = tid; i < nwaves * HIER_QUAD_PUS; i += nthreads)
                s_qslot[i] = ~0u;
        const uint8_t *r0 = (const uint8_t*)ref.origin;
        const int64_t lo = ref.lo, hi = ref.hi - PPD;
        const int wx0 = B.ctu_x + B.mvx0, wy0 = B.ctu_y + B.mvy0 + y_lo;
        const int nd = it.win_nd;
        const int chunks = (nd + 3) >> 2;               // 16-byte pieces of a window row
        const int64_t first = (int64_t)wy0 * ref.stride + wx0;
        const int64_t last16 = (int64_t)(wy0 + it.win_rows - 1) * ref.stride + wx0 + 4 * PPD * (chunks - 1);   // the last 16-byte piece
        if (first >= ref.lo && last16 + 4 * PPD <= ref.hi && chunks <= 64)
        {
            // the whole window (with the bytes a row's last 16-byte piece reads beyond it) lies inside the plane allocation:
            // plain 32-bit offsets, 16 bytes per lane, 64 / chunks window rows per load instruction
            const uint8_t *w0 = r0 + first * PB;
            const int stride_b = ref.stride * PB;
            const int rpi = 64 / chunks, rr = lane / chunks, ch = lane - rr * chunks;
            struct __attribute__((packed)) U4 { uint32_t v[4]; };
            if (rr < rpi)
                for (int r = wv * rpi + rr; r < it.win_rows; r += nwaves * rpi)
                {
                    const XA_GLOBAL U4 *src = (const XA_GLOBAL U4*)(uintptr_t)(w0 + r * stride_b + 16 * ch);
                    const uint32_t d0 = src->v[0], d1 = src->v[1], d2 = src->v[2], d3 = src->v[3];
                    uint32_t *dst = s_win + r * pitch + 4 * ch;
                    dst[0] = d0;
                    if (4 * ch + 1 < nd) dst[1] = d1;
                    if (4 * ch + 2 < nd) dst[2] = d2;
                    if (4 * ch + 3 < nd) dst[3] = d3;
                }
        }
        else
            for (int r = wv; r < it.win_rows; r += nwaves)
                for (int c = lane; c < nd; c += 64)
                {
                    int64_t a = (int64_t)(wy0 + r) * ref.stride + wx0 + PPD * c;
                    a = a < lo ? lo : (a > hi ? hi : a);
                    s_win[r * pitch + c] = ld_u32(r0 + a * PB);
                }
        const uint16_t *cost = P.cost[B.qp];
        if (MASK)
        {
            // per-job ranges and predictors: four phase tables per axis over d = mv - (qmvp >> 2) (HierBundle::pred)
            for (int i = tid; i < 4 * lxp; i += nthreads)
            {
                const int ph = i / lxp;
                s_cx[i] = cost[(int16_t)(4 * (B.dx0 + i - ph * lxp) - ph)];
            }
            for (int i = tid; i < 4 * lyp; i += nthreads)
            {
                const int ph = i / lyp;
                s_cy[i] = cost[(int16_t)(4 * (B.dy0 + i - ph * lyp) - ph)];
            }
        }
        else
        {
            for (int i = tid; i < nxw; i += nthreads)
                s_cx[i] = cost[(int16_t)((int16_t)((B.mvx0 + i) << 2) - B.qmvp_x)];
            for (int i = tid; i < nyw; i += nthreads)
                s_cy[i] = cost[(int16_t)((int16_t)((B.mvy0 + i) << 2) - B.qmvp_y)];
        }
        for (int i = tid; i < HAD_WAVES * HIER_SMALL; i += nthreads)
            s_slot[i] = ~0u;
        if (tid < HAD_WAVES * 5)
            s_big[tid] = ~0ull;
    }
    __syncthreads();

    const uint64_t cells = B.cells;
    for (int base = it.t_lo; base < it.t_hi; base += nthreads)
    {
        const int task = base + tid;
        const bool active = task < it.t_hi;
        const int tk = active ? task : it.t_lo;
        const int y = tk / XP, x = NC * (tk - y * XP);                  // the task's first candidate (x, y) of the bundle window
        uint32_t mv[NC];
        bool valid[NC];
#pragma unroll
        for (int c = 0; c < NC; c++)
        {
            valid[c] = active && x + c < nxw;
            if (!MASK)
                mv[c] = (s_cx[x + c < nxw ? x + c : nxw - 1] + s_cy[y]) & 0xffffu;
        }
        // the MV costs of the lane's candidates for job J (its own predictor)
        auto job_mv = [&](const x265amd_me_job &J, uint32_t *m) {
            const uint16_t *tx = s_cx + (J.qmvp_x & 3) * lxp + (B.mvx0 - (J.qmvp_x >> 2) - B.dx0) + x;
            const uint32_t cyv = s_cy[(J.qmvp_y & 3) * lyp + (B.mvy0 - (J.qmvp_y >> 2) - B.dy0) + y];
#pragma unroll
            for (int c = 0; c < NC; c++)
                m[c] = (tx[c] + cyv) & 0xffffu;
        };
        const uint32_t lidx = (uint32_t)((y - y_lo) * nxw + x - x_lo);  // slab-local index of candidate 0
        const uint32_t ridx = (uint32_t)(y * nxw + x);                  // its raster index in the window
        const uint32_t *wbase = s_win + (y - y_lo) * pitch;

        // raw sums of the running 16x16 (SAD, or half Hadamard sums); values of the running 32x32 / 64x64
        uint32_t q16[NC], a32[NC], a64[NC];
#pragma unroll
        for (int c = 0; c < NC; c++)
            q16[c] = a32[c] = a64[c] = 0;
        // QUADS: strips of the running 16x16 block in SATD / SAD units: rows 0-3, rows 12-15, columns 0-3, columns 12-15, all
        [[maybe_unused]] uint32_t bq[5][QUADS ? NC : 1];
        if constexpr (QUADS)
#pragma unroll
            for (int i = 0; i < 5; i++)
#pragma unroll
                for (int c = 0; c < NC; c++)
                    bq[i][c] = 0;
        // is candidate c outside job jid's own range?
        auto kill = [&](int jid, int c) -> bool {
            if (!MASK)
                return !valid[c];
            const x265amd_me_job &J = jobs[jid];
            return !valid[c] || x + c < J.mvmin_x - B.mvx0 || x + c > J.mvmax_x - B.mvx0 || y < J.mvmin_y - B.mvy0 || y > J.mvmax_y - B.mvy0;
        };
        auto key32 = [&](int jid, const uint32_t *v) -> uint32_t {
            uint32_t k = ~0u;
            uint32_t mj[NC];
            if (MASK)
                job_mv(jobs[jid], mj);
#pragma unroll
            for (int c = 0; c < NC; c++)
            {
                const uint32_t kc = kill(jid, c) ? ~0u : ((v[c] + (MASK ? mj[c] : mv[c])) << HAD_RIDX_BITS) | (lidx + c);
                k = kc < k ? kc : k;
            }
            return k;
        };
        auto key64 = [&](int jid, const uint32_t *v) -> unsigned long long {
            unsigned long long k = ~0ull;
            uint32_t mj[NC];
            if (MASK)
                job_mv(jobs[jid], mj);
#pragma unroll
            for (int c = 0; c < NC; c++)
            {
                const unsigned long long kc = kill(jid, c) ? ~0ull : ((unsigned long long)(v[c] + (MASK ? mj[c] : mv[c])) << 32) | (ridx + c);
                k = kc < k ? kc : k;
            }
            return k;
        };
        // running sums for the rectangular PUs (had_rect_value): the last three cells / 16x16 values, the strips of cells of
        // the running 32x32 block, the strips of 16x16 values of the 64x64 cell
        uint32_t cprev[3][NC], qprev[3][NC], S[4][NC], T[4][NC], RQ[4][NC], CQ[4][NC];
        if (RECTS)
        {
#pragma unroll
            for (int c = 0; c < NC; c++)
            {
#pragma unroll
                for (int i = 0; i < 3; i++)
                    cprev[i][c] = qprev[i][c] = 0;
#pragma unroll
                for (int i = 0; i < 4; i++)
                    S[i][c] = T[i][c] = RQ[i][c] = CQ[i][c] = 0;
            }
        }
        // rectangular PUs [k0, k1) of the bundle.  kind bit 31: the PU's cost may exceed the 21 bits of a 32-bit key (set
        // by the planner from the PU's area).
        auto rects = [&](int k0, int k1, const uint32_t (*St)[NC], const uint32_t (*Tt)[NC],
                         const uint32_t *Q0, const uint32_t *Q1, const uint32_t *Q2, const uint32_t *Q3) {
#pragma nounroll
            for (int k = k0; k < k1; k++)
            {
                const int jid = B.rect[k].job;
                const uint32_t kword = B.rect[k].kind;
                uint32_t v[NC];
                had_rect_value<NC>((int)(kword & 0xffu), St, Tt, Q0, Q1, Q2, Q3, v);
                unsigned long long *sl = s_rk + wv * HIER_MAX_RECT + k;
                if (kword >> 31)
                {
                    const unsigned long long m = wave_min_u64(key64(jid, v));
                    if (lane == 0)
                        atomicMin(sl, m);                              // (ds_min_u64)
                }
                else
                {
                    const uint32_t m = hier_wave_min_last(key32(jid, v));
                    if (lane == 63 && m != ~0u)
                    {
                        const unsigned long long m64 = ((unsigned long long)(m >> HAD_RIDX_BITS) << 32) |
                                                       (uint32_t)(y_lo * nxw + x_lo + (int)(m & ((1u << HAD_RIDX_BITS) - 1)));
                        atomicMin(sl, m64);
                    }
                }
            }
        };
        uint64_t rem = cells;
        int z = rem ? __builtin_ctzll(rem) : 64;
#pragma nounroll
        while (z < 64)
        {
            rem &= rem - 1;
            const int nz = rem ? __builtin_ctzll(rem) : 64;
            int cx, cy;
            hier_cell_xy(z, cx, cy);
            constexpr int FDW = SAD8 ? 16 : 32;
            const XA_CONST uint32_t *fe = pf + z * FDW;
            uint32_t f[FDW];
#pragma unroll
            for (int e = 0; e < FDW; e++)
                f[e] = fe[e];
            uint32_t h[NC];                                              // the cell's SAD, or S8 / 2
            [[maybe_unused]] uint32_t hq[4][NC];                         // QUADS: the SADs / 4x4 SATDs of its four quadrants
            if constexpr (QUADS)
            {
                if constexpr (SAD8)
                    sad8x8_quad_u8_quads(wbase + cy * 8 * pitch, pitch, x + cx * 8, f, hq);
                else if constexpr (SAD)
                    sad8x8_pair_u16_quads(wbase + cy * 8 * pitch, pitch, x + cx * 8, f, hq);
                else
                {
                    uint32_t rows[8][4], t[4];
#pragma unroll
                    for (int r = 0; r < 8; r++)
                        had_load_row<PIX>(wbase + (cy * 8 + r) * pitch, x + cx * 8, rows[r]);
                    h[0] = had8x8_half_quads(rows, f, t);
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        hq[q][0] = t[q];
                }
                if constexpr (SAD)
#pragma unroll
                    for (int c = 0; c < NC; c++)
                        h[c] = hq[0][c] + hq[1][c] + hq[2][c] + hq[3][c];
            }
            else if constexpr (SAD8)
                sad8x8_quad_u8(wbase + cy * 8 * pitch, pitch, x + cx * 8, f, h);
            else if constexpr (SAD)
                sad8x8_pair_u16(wbase + cy * 8 * pitch, pitch, x + cx * 8, f, h);
            else
            {
                uint32_t rows[8][4];
#pragma unroll
                for (int r = 0; r < 8; r++)
                    had_load_row<PIX>(wbase + (cy * 8 + r) * pitch, x + cx * 8, rows[r]);
                h[0] = had8x8_half(rows, f);
            }
            if constexpr (QUADS)
            {
                if (quads)
                {
                    // halves of the cell
                    const int32_t *cj = QJ->cell[z];
#pragma unroll
                    for (int k = 0; k < 4; k++)
                    {
                        const int jid = cj[k];
                        if (jid < 0)
                            continue;
                        const int qa = k == 0 ? 0 : k == 1 ? 2 : k == 2 ? 0 : 1, qb = k == 0 ? 1 : k == 1 ? 3 : k == 2 ? 2 : 3;
                        uint32_t v[NC];
#pragma unroll
                        for (int c = 0; c < NC; c++)
                            v[c] = hq[qa][c] + hq[qb][c];
                        const uint32_t m = hier_wave_min_last(key32(jid, v));
                        if (lane == 63)
                            atomicMin(s_qslot + wv * HIER_QUAD_PUS + 4 * z + k, m);
                    }
                    // strips of the 16x16 block: its first / last 4 rows and columns, and everything
                    const int zx = z & 1, zy = (z >> 1) & 1;
#pragma unroll
                    for (int c = 0; c < NC; c++)
                    {
                        bq[0][c] += zy == 0 ? hq[0][c] + hq[1][c] : 0;
                        bq[1][c] += zy == 1 ? hq[2][c] + hq[3][c] : 0;
                        bq[2][c] += zx == 0 ? hq[0][c] + hq[2][c] : 0;
                        bq[3][c] += zx == 1 ? hq[1][c] + hq[3][c] : 0;
                        bq[4][c] += hq[0][c] + hq[1][c] + hq[2][c] + hq[3][c];
                    }
                }
            }
            uint32_t v8[NC];
#pragma unroll
            for (int c = 0; c < NC; c++)
                v8[c] = SAD ? h[c] : (h[c] + 1) >> 1;                    // (S8 + 2) >> 2
            const int j8 = B.job[z];
            if (j8 >= 0)
            {
                const uint32_t m = hier_wave_min_last(key32(j8, v8));
                if (lane == 63)
                {
                    atomicMin(s_slot + wv * HIER_SMALL + z, m);        // (ds_min_u32: no read-back to wait for)
                }
            }
#pragma unroll
            for (int c = 0; c < NC; c++)
                q16[c] += h[c];
            if (RECTS)
            {
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    if ((cy & 3) == i)
#pragma unroll
                        for (int c = 0; c < NC; c++)
                            S[i][c] += v8[c];
                    if ((cx & 3) == i)
#pragma unroll
                        for (int c = 0; c < NC; c++)
                            T[i][c] += v8[c];
                }
                // (a 16x16 block with such PUs has all four cells covered, so they arrive one after the other)
                if ((z & 3) == 3 && B.rect_end[z >> 2] != ((z >> 2) ? B.rect_end[(z >> 2) - 1] : 0))
                    rects((z >> 2) ? B.rect_end[(z >> 2) - 1] : 0, B.rect_end[z >> 2], S, T, cprev[2], cprev[1], cprev[0], v8);
#pragma unroll
                for (int c = 0; c < NC; c++)
                {
                    cprev[2][c] = cprev[1][c];
                    cprev[1][c] = cprev[0][c];
                    cprev[0][c] = v8[c];
                }
            }
            if ((nz >> 2) != (z >> 2))      // the 16x16 block is complete
            {
                const int q = z >> 2;
                if constexpr (QUADS)
                {
                    if (quads)
                    {
                        // (a block with such PUs has all four cells covered)
                        const int32_t *bj = QJ->blk[q];
#pragma unroll
                        for (int k = 0; k < 8; k++)
                        {
                            const int jid = bj[k];
                            if (jid < 0)
                                continue;
                            const int strip = k == 0 || k == 1 ? 0 : k == 2 || k == 3 ? 1 : k == 4 || k == 5 ? 2 : 3;
                            const bool rest = k == 1 || k == 2 || k == 5 || k == 6;     // the 12-pixel part: everything but the strip
                            uint32_t v[NC];
#pragma unroll
                            for (int c = 0; c < NC; c++)
                                v[c] = rest ? bq[4][c] - bq[strip][c] : bq[strip][c];
                            const uint32_t m = hier_wave_min_last(key32(jid, v));
                            if (lane == 63)
                                atomicMin(s_qslot + wv * HIER_QUAD_PUS + 256 + 8 * q + k, m);
                        }
#pragma unroll
                        for (int i = 0; i < 5; i++)
#pragma unroll
                            for (int c = 0; c < NC; c++)
                                bq[i][c] = 0;
                    }
                }
                uint32_t v16[NC];
#pragma unroll
                for (int c = 0; c < NC; c++)
                {
                    v16[c] = SAD ? q16[c] : (q16[c] + 1) >> 1;           // (S8a + S8b + S8c + S8d + 2) >> 2
                    q16[c] = 0;
                }
                if (RECTS)
                {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        if ((cy >> 1) == i)
#pragma unroll
                            for (int c = 0; c < NC; c++)
                                RQ[i][c] += v16[c];
                        if ((cx >> 1) == i)
#pragma unroll
                            for (int c = 0; c < NC; c++)
                                CQ[i][c] += v16[c];
                    }
                }
                const int j16 = B.job[64 + q];
                if (j16 >= 0)
                {
                    const uint32_t m = hier_wave_min_last(key32(j16, v16));
                    if (lane == 63)
                    {
                        atomicMin(s_slot + wv * HIER_SMALL + 64 + q, m);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; c++)
                    a32[c] += v16[c];
                if ((nz >> 4) != (z >> 4))  // the 32x32 block is complete
                {
                    const int q32 = z >> 4;
                    // (a 32x32 block with such PUs is covered completely)
                    if (RECTS && B.rect_end[16 + q32] != B.rect_end[15 + q32])
                        rects(B.rect_end[15 + q32], B.rect_end[16 + q32], S, T, qprev[2], qprev[1], qprev[0], v16);
                    const int j32 = B.job[80 + q32];
                    if (j32 >= 0)
                    {
                        const unsigned long long m = wave_min_u64(key64(j32, a32));
                        if (lane == 0)
                            atomicMin(s_big + wv * 5 + q32, m);
                    }
#pragma unroll
                    for (int c = 0; c < NC; c++)
                    {
                        a64[c] += a32[c];
                        a32[c] = 0;
                    }
                    if (RECTS)
                    {
#pragma unroll
                        for (int i = 0; i < 4; i++)
#pragma unroll
                            for (int c = 0; c < NC; c++)
                                S[i][c] = T[i][c] = 0;
                    }
                }
                if (RECTS)
                {
#pragma unroll
                    for (int c = 0; c < NC; c++)
                    {
                        qprev[2][c] = qprev[1][c];
                        qprev[1][c] = qprev[0][c];
                        qprev[0][c] = v16[c];
                    }
                }
            }
            z = nz;
        }
        // (PUs made of 16x16 values of the whole cell: the cell is covered completely)
        if (RECTS && n_rect != B.rect_end[19])
            rects(B.rect_end[19], n_rect, RQ, CQ, RQ[0], RQ[0], RQ[0], RQ[0]);
        const int j64 = B.job[84];
        if (j64 >= 0)
        {
            const unsigned long long m = wave_min_u64(key64(j64, a64));
            if (lane == 0)
                atomicMin(s_big + wv * 5 + 4, m);
        }
    }
    __syncthreads();

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
                for (int w = 0; w < nwaves; w++)
                {
                    const uint32_t v = s_slot[w * HIER_SMALL + tid];
                    m = v < m ? v : m;
                }
                if (m != ~0u)
                {
                    cost = (int)(m >> HAD_RIDX_BITS);
                    ridx = y_lo * nxw + x_lo + (int)(m & ((1u << HAD_RIDX_BITS) - 1));
                }
            }
            else
            {
                unsigned long long m = ~0ull;
                for (int w = 0; w < nwaves; w++)
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
            fb[(size_t)jid * fb_stride + it.slot] = f;
        }
    }
    if (tid < n_rect)           // (every launch has at least HAD_THREADS >= HIER_MAX_RECT threads)
    {
        unsigned long long m = ~0ull;
        for (int w = 0; w < nwaves; w++)
        {
            const unsigned long long v = s_rk[w * HIER_MAX_RECT + tid];
            m = v < m ? v : m;
        }
        FullBest f;
        const int ridx = m != ~0ull ? (int)(m & 0xffffffffu) : -1;
        f.cost = m != ~0ull ? (int)(m >> 32) : 0x7fffffff;
        f.mvx = ridx < 0 ? 0 : (int16_t)(B.mvx0 + ridx % nxw);
        f.mvy = ridx < 0 ? 0 : (int16_t)(B.mvy0 + ridx / nxw);
        fb[(size_t)B.rect[tid].job * fb_stride + it.slot] = f;
    }
    if constexpr (QUADS)
    {
        if (quads)
            for (int k = tid; k < HIER_QUAD_PUS; k += nthreads)
            {
                const int jid = k < 256 ? QJ->cell[k >> 2][k & 3] : QJ->blk[(k - 256) >> 3][(k - 256) & 7];
                if (jid < 0)
                    continue;
                uint32_t m = ~0u;
                for (int w = 0; w < nwaves; w++)
                {
                    const uint32_t v = s_qslot[w * HIER_QUAD_PUS + k];
                    m = v < m ? v : m;
                }
                FullBest f;
                const int ridx = m != ~0u ? y_lo * nxw + x_lo + (int)(m & ((1u << HAD_RIDX_BITS) - 1)) : -1;
                f.cost = m != ~0u ? (int)(m >> HAD_RIDX_BITS) : 0x7fffffff;
                f.mvx = ridx < 0 ? 0 : (int16_t)(B.mvx0 + ridx % nxw);
                f.mvy = ridx < 0 ? 0 : (int16_t)(B.mvy0 + ridx / nxw);
                fb[(size_t)jid * fb_stride + it.slot] = f;
            }
    }
}

// (16-bit pixels: 4 wavefronts per SIMD hide the LDS latency of the row reads; the register allocator is held to that)
template <typename PIX, bool RECTS, bool QUADS>
__global__ void __launch_bounds__(HAD_THREADS_BIG) __attribute__((amdgpu_waves_per_eu(sizeof(PIX) == 2 ? 4 : 1))) k_full_had(BatchParams P, const x265amd_me_job *jobs, const HierBundle *bundles,
                                                              const HierItem *items, int n_items, const uint32_t *packed,
                                                              const HierQuads *quads, FullBest *fb, int fb_stride)
{
    extern __shared__ __align__(16) uint32_t lds[];
    const int id = blockIdx.x;
    if (id >= n_items)
        return;
    const HierItem &it = items[id];
    const HierBundle &B = bundles[it.bundle];
    const XA_CONST uint32_t *pf = (const XA_CONST uint32_t*)(uintptr_t)(packed + (size_t)it.bundle * HAD_FENC_DW);
    const HierQuads *qj = QUADS ? quads + it.bundle : nullptr;
    if (B.metric == X265AMD_SAD)
    {
        if (B.uniform)
            had_item_body<PIX, false, true, RECTS, QUADS>(P, jobs, B, it, pf, qj, fb, fb_stride, lds);
        else
            had_item_body<PIX, true, true, RECTS, QUADS>(P, jobs, B, it, pf, qj, fb, fb_stride, lds);
    }
    else if (B.uniform)
        had_item_body<PIX, false, false, RECTS, QUADS>(P, jobs, B, it, pf, qj, fb, fb_stride, lds);
    else
        had_item_body<PIX, true, false, RECTS, QUADS>(P, jobs, B, it, pf, qj, fb, fb_stride, lds);
}

} // namespace xa
