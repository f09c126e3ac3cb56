// Quad-per-job motion search for the smallest PUs (gfx950): FOUR LANES own one job, one 4x4 tile each.
//
// The general kernel (me_kernels.h, me_device.h) gives every job a wavefront and spreads the PU's 4-pixel
// groups over its lanes: an 8x4 PU then uses 8 of 64 lanes and an 8x8 PU 16, while the scalar search logic,
// the reductions and the calls cost the same as for a PU that fills the wavefront.  Here a PU of at most 64
// pixels (8x8, 8x4, 4x8, 16x4, 4x16, ...) is at most four 4x4 tiles, and lane t of a quad of lanes takes tile t:
// 16 jobs per wavefront.  The search itself (Search<> of me_search.h, the reference's encoder/motion.cpp:277-866)
// runs as ordinary SIMT code, identically in the four lanes of a quad (they hold the same search state), and
// differently from quad to quad.  A block cost is the lane's tile cost summed over the quad by two DPP
// quad-permute adds:
//   * sad                 v_sad_u8 / v_sad_u16 on the tile's four rows (common/pixel.cpp:78-95); the candidates of one
//                         pattern step (sad_x3 / sad_x4 and the 6- and 8-point steps) are priced by ONE call whose
//                         loads are all in flight together - a search is a chain of memory round trips, not of ALU work;
//   * satd                4x4 Hadamard of the tile in the lane's registers, halved after the sum (exactly additive over
//                         tiles: a tile's sum of magnitudes is even; common/pixel.cpp:187-274, table :652-722);
//   * sa8d_inter          8x8: H8 = H2 (x) H4, so the quad's four 4x4 transforms get one more butterfly stage across
//                         the lanes (DPP) per direction, rounded (S + 2) >> 2; every other shape of this class has a 4-
//                         or 12-pixel side and takes the SATD recipe (common/pixel.cpp:276-362, table :756-826);
//   * sub-pel candidates  the lane interpolates its tile with the same four 8-tap primitives as the general kernel
//                         (Dev<>::h_pp / v_pp / h_ps / v_sp, common/ipfilter.cpp).
// Every 4- or 12-pixel read is clamped into the plane allocation exactly like the general kernel's reads
// (Dev<>::Src), so the two kernels agree bit for bit on every job, in-margin or not.
//
// The same code with SIXTEEN lanes per job (LPJ = 16: one 16x16 block of tiles per step, four jobs per wavefront) takes
// the PUs of up to 32x32 pixels whose exhaustive scan ran in its own kernel: what is left of such a job is a handful of
// evaluations, for which a wavefront per job spends most of its instructions on set-up.
//
// Neighbouring quads hold neighbouring PUs (the planner orders this class by shape, search kind and picture
// position), so a wavefront's row reads fall into a few cache lines and its quads mostly run the same loops.
#pragma once

#include "me_kernels.h"

namespace xa {

struct LaneCfg { static constexpr bool LR = false, CP = false; };

// The search's small tables as 4-bit immediates: column c of a table of up to 16 rows in one 64-bit constant, entries
// biased by +4 (all lie in [-4, 8]); a lookup is a shift and a mask instead of a per-lane memory read.
template <int N, int M>
constexpr uint64_t nib_pack(const int (&t)[N][M], int c)
{
    uint64_t v = 0;
    for (int i = 0; i < N; i++)
        v |= (uint64_t)((t[i][c] + 4) & 15) << (4 * i);
    return v;
}
constexpr int k_square1[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { -1, 1 }, { 1, -1 }, { 1, 1 } };
constexpr int k_hex2[8][2] = XA_T_HEX2;
constexpr int k_hex4[16][2] = XA_T_HEX4;
constexpr int k_star[16][2] = XA_T_STAR;
constexpr int k_workload[8][5] = XA_T_WORKLOAD;
constexpr int k_refine_h[9][2] = XA_T_REFINE_H;
constexpr int k_refine_q[9][2] = XA_T_REFINE_Q;
constexpr int k_range_mul[4][4] = XA_T_RANGE_MUL;
template <int TAB> constexpr uint64_t nib_table(int c)
{
    return TAB == T_SQUARE1 ? nib_pack(k_square1, c) : TAB == T_HEX2 ? nib_pack(k_hex2, c) : TAB == T_HEX4 ? nib_pack(k_hex4, c) :
           TAB == T_STAR ? nib_pack(k_star, c) : TAB == T_WORKLOAD ? nib_pack(k_workload, c) : TAB == T_REFINE_H ? nib_pack(k_refine_h, c) :
           TAB == T_REFINE_Q ? nib_pack(k_refine_q, c) : 0;
}
// (range_mul is indexed [i][c] with both variable: its 16 entries in one constant)
constexpr uint64_t nib_range_mul()
{
    uint64_t v = 0;
    for (int i = 0; i < 4; i++)
        for (int c = 0; c < 4; c++)
            v |= (uint64_t)((k_range_mul[i][c] + 4) & 15) << (4 * (4 * i + c));
    return v;
}
typedef int P8 __attribute__((ext_vector_type(8)));     // eight values in registers (a struct would travel through the stack)

template <typename PIX, int LPJ = 4>      // LPJ: lanes per job, 4 or 16
struct QuadWave
{
    typedef Dev<JobLds<PIX, 1, 64>> D;
    static constexpr int DW = D::DW;
    static_assert(LPJ == 4 || LPJ == 16, "a job is a quad or a 16-lane row of lanes");

    // the job, identical in the four lanes of its quad
    const PIX *enc;                 // the PU's first pixel in the encoder plane
    const PIX *ref;                 // the reference pixel under it (encoder/motion.cpp:288)
    const uint16_t *cost;           // centre of the per-QP MV cost table
    int estride, stride, gLo, gHi;  // [gLo, gHi): element offsets relative to `ref` of the whole padded allocation
    int depth, w, h, mvpx, mvpy, ikind;
    unsigned evals;
    static constexpr bool lowres = false;

    __device__ __forceinline__ void stamp(int) {}

    template <int TAB> static __device__ __forceinline__ int tab(int i, int c)
    {
        if (TAB == T_RANGE_MUL)
            return (int)((nib_range_mul() >> (4 * (4 * i + c))) & 15) - 4;
        constexpr uint64_t c0 = nib_table<TAB>(0), c1 = nib_table<TAB>(1), c2 = TAB == T_WORKLOAD ? nib_table<TAB>(2) : 0,
                           c3 = TAB == T_WORKLOAD ? nib_table<TAB>(3) : 0, c4 = TAB == T_WORKLOAD ? nib_table<TAB>(4) : 0;
        const uint64_t t = c == 0 ? c0 : c == 1 ? c1 : c == 2 ? c2 : c == 3 ? c3 : c4;
        return (int)((t >> (4 * i)) & 15) - 4;
    }

    __device__ __forceinline__ void bind(const PlaneDesc &e, const PlaneDesc &r, int pu_x, int pu_y, int w_, int h_, const uint16_t *cost_)
    {
        enc = (const PIX*)e.origin + (int64_t)pu_y * e.stride + pu_x;
        estride = e.stride;
        const int64_t blockOffset = (int64_t)pu_y * r.stride + pu_x;
        ref = (const PIX*)r.origin + blockOffset;
        stride = r.stride;
        gLo = (int32_t)(r.lo - blockOffset);
        gHi = (int32_t)(r.hi - blockOffset);
        depth = DW == 1 ? 8 : r.depth;
        w = w_; h = h_;
        cost = cost_;
        mvpx = mvpy = 0;
        ikind = X265AMD_SAD;
        evals = 0;
    }
    __device__ __forceinline__ void set_mvp(int x, int y) { mvpx = x; mvpy = y; }

    // ---------------------------------------------------------------- helpers
    static __device__ __forceinline__ int quad_sum(int v)
    {
        v += dpp<0xB1>(v);          // quad_perm [1,0,3,2]
        v += dpp<0x4E>(v);          // quad_perm [2,3,0,1]
        return v;
    }
    // sum over the job's lanes, result in all of them
    static __device__ __forceinline__ int job_sum(int v)
    {
        v = quad_sum(v);
        if (LPJ == 16)
        {
            v += xor4(v);
            v += xor8(v);
        }
        return v;
    }
    static __device__ __forceinline__ int clamp_off(int o, int len, int gLo, int gHi)
    {
        o = o < gLo ? gLo : o;
        return o > gHi - len ? gHi - len : o;
    }
    static __device__ __forceinline__ void rd4(const PIX *p, uint32_t *r)
    {
        if (DW == 1)
            r[0] = ld_u32(p);
        else
        {
            const uint64_t v = ld_u64(p);
            r[0] = (uint32_t)v;
            r[1] = (uint32_t)(v >> 32);
        }
    }
    static __device__ __forceinline__ int mv_cost(const uint16_t *cost, int mvp, int label, int cflags)
    {
        const int dx = (int16_t)((int16_t)label - (int16_t)mvp), dy = (int16_t)((label >> 16) - (mvp >> 16));
        const XA_GLOBAL uint16_t *tab = (const XA_GLOBAL uint16_t*)(uintptr_t)cost;
        const int a = tab[dx], b = tab[dy];
        return (cflags & CF_NOTRUNC) ? a + b : ((a + b) & 0xffff);
    }

    // The PU's 4x4 tiles are dealt to the job's lanes in `tile_steps` steps of LPJ tiles.  PUs whose sides are multiples
    // of 8: a quad of lanes takes one 8x8 block per step, lane q of the quad its quadrant q (so the 8x8 transform's last
    // stage stays inside the quad); sides multiples of 16: the four blocks of a 16x16 block are consecutive (sa8d_inter
    // rounds per 16x16 there, common/pixel.cpp:347-362: four steps of a quad, or one step of a 16-lane row).  Any other
    // PU: tiles in raster order.  `on` = the lane has a tile in this step (one without repeats tile 0 and adds nothing).
    static __device__ __forceinline__ int tile_steps(int w, int h)
    {
        return ((w | h) & 7) ? (((w * h) >> 4) + LPJ - 1) / LPJ : (((w * h) >> 6) + LPJ / 4 - 1) / (LPJ / 4);
    }
    static __device__ __forceinline__ void tile_at(int i, int w, int h, int &tx, int &ty, bool &on)
    {
        const int t = (int)(threadIdx.x & (LPJ - 1));
        if (((w | h) & 7) == 0)
        {
            const int b0 = i * (LPJ / 4) + (t >> 2);
            on = b0 < ((w * h) >> 6);
            const int b = on ? b0 : 0;
            int bx, by;
            if (((w | h) & 15) == 0)
            {
                const int g = b >> 2, q = b & 3, ngx = w >> 4, gy = g / ngx, gx = g - gy * ngx;
                bx = gx * 2 + (q & 1);
                by = gy * 2 + (q >> 1);
            }
            else
            {
                const int nbx = w >> 3;
                by = b / nbx;
                bx = b - by * nbx;
            }
            tx = bx * 8 + (t & 1) * 4;
            ty = by * 8 + ((t >> 1) & 1) * 4;
        }
        else
        {
            const int ntx = w >> 2, tt = LPJ * i + t;
            on = tt < ((w * h) >> 4);
            const int u = on ? tt : 0;
            ty = u / ntx;
            tx = (u - ty * ntx) * 4;
            ty *= 4;
        }
    }

    // ---------------------------------------------------------------- full-pel SADs of up to NB candidates in one call
    // pos[k] = (y << 16) | (x & 0xffff): full-pel position measured; lbl[k]: q-pel label priced.  Entries >= n repeat
    // entry 0 (their loads hit the same lines; their costs are not used).  All loads are issued before the first use.
    template <int NB>
    static __device__ __noinline__ P8 sad_multi(const PIX *enc, int estride, const PIX *ref, int stride, int gLo, int gHi, int wh,
                                                const uint16_t *cost, int mvp, int cflags, P8 pos, P8 lbl)
    {
        const int w = wh & 0xff, h = wh >> 8, steps = tile_steps(w, h);
        int mv[NB], off[NB];
        uint32_t s[NB];
#pragma unroll
        for (int k = 0; k < NB; k++)
        {
            off[k] = (pos[k] >> 16) * stride + (int)(int16_t)pos[k];
            mv[k] = (cflags & CF_NOMV) ? 0 : mv_cost(cost, mvp, lbl[k], cflags);
            s[k] = 0;
        }
        for (int i = 0; i < steps; i++)
        {
            int tx, ty;
            bool on;
            tile_at(i, w, h, tx, ty, on);
            const PIX *et = enc + ty * estride + tx;
            const int toff = ty * stride + tx;
            uint32_t e[4][2], r[NB][4][2];
#pragma unroll
            for (int y = 0; y < 4; y++)
                rd4(et + y * estride, e[y]);
#pragma unroll
            for (int k = 0; k < NB; k++)
#pragma unroll
                for (int y = 0; y < 4; y++)
                    rd4(ref + clamp_off(off[k] + toff + y * stride, 4, gLo, gHi), r[k][y]);
#pragma unroll
            for (int k = 0; k < NB; k++)
            {
                uint32_t a = 0;
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    if (DW == 1)
                        a = __builtin_amdgcn_sad_u8(e[y][0], r[k][y][0], a);
                    else
                        a = __builtin_amdgcn_sad_u16(e[y][1], r[k][y][1], __builtin_amdgcn_sad_u16(e[y][0], r[k][y][0], a));
                }
                s[k] += on ? a : 0;
            }
        }
        P8 out = { 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
        for (int k = 0; k < NB; k++)
            out[k] = job_sum((int)s[k]) + mv[k];
        return out;
    }

    // ---------------------------------------------------------------- any kind, any q-pel position, one candidate
    // 4x4 Hadamard of d in place (rows, then columns), no normalisation
    static __device__ __forceinline__ void had4(int (*d)[4])
    {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int a0 = d[r][0] + d[r][1], a1 = d[r][0] - d[r][1], a2 = d[r][2] + d[r][3], a3 = d[r][2] - d[r][3];
            d[r][0] = a0 + a2; d[r][1] = a1 + a3; d[r][2] = a0 - a2; d[r][3] = a1 - a3;
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
        {
            const int a0 = d[0][c] + d[1][c], a1 = d[0][c] - d[1][c], a2 = d[2][c] + d[3][c], a3 = d[2][c] - d[3][c];
            d[0][c] = a0 + a2; d[1][c] = a1 + a3; d[2][c] = a0 - a2; d[3][c] = a1 - a3;
        }
    }

    // Cost of kind `kind` of the PU against the candidate at q-pel (qx, qy) (subpelCompare, encoder/motion.cpp:1140-1165;
    // at a full-pel position the plain primitive) plus, unless CF_NOMV, the MV cost of the q-pel label.
    static __device__ __noinline__ int cost_one(const PIX *enc, int estride, const PIX *ref, int stride, int gLo, int gHi,
                                                int dwh, const uint16_t *cost, int mvp, int qx, int qy, int label, int kf)
    {
        const int depth = dwh & 0xff, w = (dwh >> 8) & 0xff, h = dwh >> 16;
        const int kind = kf & 3, cflags = kf >> 2;
        const int xf = qx & 3, yf = qy & 3;
        const int off0 = (qy >> 2) * stride + (qx >> 2);
        const int mv = (cflags & CF_NOMV) ? 0 : mv_cost(cost, mvp, label, cflags);
        const int am = D::accum_mode(kind, w, h), steps = tile_steps(w, h);
        // acc: the lane's magnitudes not yet rounded; tot: rounded sums, kept by ONE lane of the rounding unit (a quad for
        // sa8d8's 8x8 blocks, the quad over four steps or the 16-lane row for sa8d16's 16x16 blocks)
        const int t = (int)(threadIdx.x & (LPJ - 1));
        int acc = 0, tot = 0;
        for (int i = 0; i < steps; i++)
        {
            int tx, ty;
            bool on;
            tile_at(i, w, h, tx, ty, on);
            const PIX *et = enc + ty * estride + tx;
            const int off = off0 + ty * stride + tx;
            uint32_t e[4][2];
#pragma unroll
            for (int y = 0; y < 4; y++)
                rd4(et + y * estride, e[y]);
            // the lane's 4x4 tile of the prediction block
            int p[4][4];
            if (!xf && !yf)
            {
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    uint32_t d[2];
                    rd4(ref + clamp_off(off + y * stride, 4, gLo, gHi), d);
                    const P4 v = D::unpack4(d);
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        p[y][c] = v.v[c];
                }
            }
            else if (!yf)           // filterHorizontal_p_p
            {
                uint32_t px[4][6];
#pragma unroll
                for (int y = 0; y < 4; y++)
                {
                    const PIX *q = ref + clamp_off(off + y * stride - 3, 12, gLo, gHi);
#pragma unroll
                    for (int k = 0; k < 3 * DW; k++)
                        px[y][k] = ld_u32((const uint8_t*)q + 4 * k);
                }
#pragma unroll
                for (int y = 0; y < 4; y++)
                    D::h_pp(px[y], xf, depth, p[y]);
            }
            else if (!xf)           // filterVertical_p_p
            {
                uint32_t rw[11][2];
#pragma unroll
                for (int k = 0; k < 11; k++)
                {
                    rw[k][1] = 0;
                    rd4(ref + clamp_off(off + (k - 3) * stride, 4, gLo, gHi), rw[k]);
                }
#pragma unroll
                for (int y = 0; y < 4; y++)
                    D::v_pp(rw + y, yf, depth, p[y]);
            }
            else                    // filterHorizontal_p_s over rows -3 .. 7, then filterVertical_s_p
            {
                uint32_t px[11][6], im[11][2];
#pragma unroll
                for (int k = 0; k < 11; k++)
                {
                    const PIX *q = ref + clamp_off(off + (k - 3) * stride - 3, 12, gLo, gHi);
#pragma unroll
                    for (int c = 0; c < 3 * DW; c++)
                        px[k][c] = ld_u32((const uint8_t*)q + 4 * c);
                }
#pragma unroll
                for (int k = 0; k < 11; k++)
                {
                    int sm[4];
                    D::h_ps(px[k], xf, depth, sm);
                    im[k][0] = __builtin_amdgcn_perm((uint32_t)sm[1], (uint32_t)sm[0], 0x05040100);
                    im[k][1] = __builtin_amdgcn_perm((uint32_t)sm[3], (uint32_t)sm[2], 0x05040100);
                }
#pragma unroll
                for (int y = 0; y < 4; y++)
                    D::v_sp(im + y, yf, depth, p[y]);
            }
            int d[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
            {
                const P4 f = D::unpack4(e[y]);
#pragma unroll
                for (int c = 0; c < 4; c++)
                    d[y][c] = on ? f.v[c] - p[y][c] : 0;
            }
            if (am >= 1)
                had4(d);
            if (am >= 2)
            {
                // the 8x8 transform = the four tiles' 4x4 transforms and one butterfly across the quad per direction
                const bool hx = (threadIdx.x & 1) != 0, hy = (threadIdx.x & 2) != 0;
#pragma unroll
                for (int y = 0; y < 4; y++)
#pragma unroll
                    for (int c = 0; c < 4; c++)
                    {
                        int v = d[y][c];
                        const int px = dpp<0xB1>(v);
                        v = hx ? px - v : v + px;
                        const int py = dpp<0x4E>(v);
                        d[y][c] = hy ? py - v : v + py;
                    }
            }
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int c = 0; c < 4; c++)
                    acc += d[y][c] < 0 ? -d[y][c] : d[y][c];
            if (am == 2)
            {
                const int r = (quad_sum(acc) + 2) >> 2;     // sa8d8: one rounding per 8x8 block
                tot += (t & 3) == 0 && on ? r : 0;
                acc = 0;
            }
            else if (am == 3 && (LPJ == 16 || (i & 3) == 3))
            {
                const int r = (job_sum(acc) + 2) >> 2;      // sa8d16: one rounding per 16x16 block
                tot += t == 0 ? r : 0;
                acc = 0;
            }
        }
        int res = job_sum(am <= 1 ? acc : tot);
        if (am == 1)
            res >>= 1;                                      // every 4x4 sum of magnitudes is even
        return res + mv;
    }

    __device__ __forceinline__ int cost_at(int qx, int qy, int lx, int ly, int kind, int cflags) const
    {
        return cost_one(enc, estride, ref, stride, gLo, gHi, depth | (w << 8) | (h << 16), cost,
                        (mvpx & 0xffff) | (mvpy << 16), qx, qy, (lx & 0xffff) | (ly << 16), kind | (cflags << 2));
    }

    // ---------------------------------------------------------------- the interface Search<> drives (as Wave<>)
    template <int N>
    __device__ __forceinline__ void eval_points(const int *sx, const int *sy, const int *lx, const int *ly, int *out,
                                                int pflags = 0, int shift3 = -1)
    {
        evals += (unsigned)N;
        if (ikind == X265AMD_SAD)
        {
            constexpr int NB = N == 1 ? 1 : N <= 4 ? 4 : 8;
            P8 pos, lbl;
#pragma unroll
            for (int k = 0; k < 8; k++)
            {
                const int s = k < N ? k : 0, sh = (s == shift3) ? 3 : 2;
                pos[k] = (sy[s] << 16) | (sx[s] & 0xffff);
                lbl[k] = ((int)(int16_t)((int16_t)ly[s] << sh) << 16) | ((int)(int16_t)((int16_t)lx[s] << sh) & 0xffff);
            }
            const P8 c = sad_multi<NB>(enc, estride, ref, stride, gLo, gHi, w | (h << 8), cost, (mvpx & 0xffff) | (mvpy << 16), pflags, pos, lbl);
#pragma unroll
            for (int k = 0; k < N; k++)
                out[k] = c[k];
            return;
        }
#pragma unroll
        for (int k = 0; k < N; k++)
        {
            const int sh = (k == shift3) ? 3 : 2;
            out[k] = cost_at(sx[k] * 4, sy[k] * 4, (int16_t)((int16_t)lx[k] << sh), (int16_t)((int16_t)ly[k] << sh), ikind, pflags);
        }
    }
    __device__ __forceinline__ int cost_fpel(int mx, int my)
    {
        int c;
        eval_points<1>(&mx, &my, &mx, &my, &c);
        return c;
    }
    __device__ __forceinline__ int subpel_cost(int qx, int qy, int kind, bool with_mv)
    {
        evals++;
        return cost_at(qx, qy, qx, qy, kind, with_mv ? 0 : CF_NOMV);
    }
    __device__ __forceinline__ void subpel_cost8(int n, const int *qx, const int *qy, int kind, int *out, int cflags = 0)
    {
        evals += (unsigned)n;
#pragma unroll
        for (int k = 0; k < 8; k++)
            out[k] = k < n ? cost_at(qx[k], qy[k], qx[k], qy[k], kind, (k == 0 && (cflags & CF_NOMV0)) ? CF_NOMV : 0) : 0;
    }
    // one refinement round: the first n of the 8 neighbours of (bmx, bmy) at distance `step`, each subpelCompare + mvcost
    __device__ __forceinline__ void round_costs(int bmx, int bmy, int step, int n, int kind, int *c)
    {
        evals += (unsigned)n;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            c[k] = 0x7fffffff;
            if (k < n)
            {
                const int qx = (int)(int16_t)(bmx + tab<T_SQUARE1>(k + 1, 0) * step), qy = (int)(int16_t)(bmy + tab<T_SQUARE1>(k + 1, 1) * step);
                c[k] = cost_at(qx, qy, qx, qy, kind, 0);
            }
        }
    }
    // winner over a job's slab partials: lowest cost, then lowest raster index (see combine_slabs).  Lane t of the job
    // reads slabs t, t + LPJ, ... (all loads in flight together); the minimum goes to all of the job's lanes.
    __device__ __forceinline__ SlabBest slab_best(const FullBest *fb, int n, int nxr, int minx, int miny) const
    {
        const int t = (int)(threadIdx.x & (LPJ - 1));
        constexpr int PER = 16 / LPJ;
        FullBest f[PER];
#pragma unroll
        for (int k = 0; k < PER; k++)
        {
            const int s = t + LPJ * k;
            f[k] = fb[s < n ? s : 0];
            if (s >= n)
                f[k].cost = 0x7fffffff;
        }
        // (cost, raster index) as two words; a slab without candidates has cost INT32_MAX and loses to every other
        uint32_t bc = 0x7fffffffu, bi = ~0u;
        int bm = 0;
#pragma unroll
        for (int k = 0; k < PER; k++)
        {
            const uint32_t c = (uint32_t)f[k].cost, i = (uint32_t)((f[k].mvy - miny) * nxr + (f[k].mvx - minx));
            const bool better = f[k].cost != 0x7fffffff && (c < bc || (c == bc && i < bi));
            bc = better ? c : bc;
            bi = better ? i : bi;
            bm = better ? (((int)f[k].mvy << 16) | ((int)f[k].mvx & 0xffff)) : bm;
        }
#pragma unroll
        for (int step = 0; step < (LPJ == 16 ? 4 : 2); step++)
        {
            auto xch = [&](int v) { return step == 0 ? dpp<0xB1>(v) : step == 1 ? dpp<0x4E>(v) : step == 2 ? xor4(v) : xor8(v); };
            const uint32_t oc = (uint32_t)xch((int)bc), oi = (uint32_t)xch((int)bi);
            const int om = xch(bm);
            const bool better = oc < bc || (oc == bc && oi < bi);
            bc = better ? oc : bc;
            bi = better ? oi : bi;
            bm = better ? om : bm;
        }
        SlabBest r;
        r.cost = (int)bc;
        r.mv = bc == 0x7fffffffu ? 0 : bm;
        return r;
    }
};

// 64 / LPJ jobs per wavefront: lane group q of workgroup b takes job idx[(64 / LPJ) b + q] (or that job of the array
// without an index list, skipping the jobs of the other classes).  LPJ = 4: the quad-per-job class (LANE_CLASS);
// LPJ = 16: the row-per-job class (ROW_CLASS).
template <typename PIX, int LPJ>
__global__ void __launch_bounds__(64) k_lane(BatchParams P, const x265amd_me_job *jobs, const int *idx, int n,
                                             const FullBest *fb, int fb_stride, x265amd_me_result *out, unsigned long long *evals)
{
    const int slot = (int)(blockIdx.x * (64 / LPJ) + threadIdx.x / LPJ);
    unsigned n_evals = 0;
    if (slot < n)
    {
        const int id = idx ? idx[slot] : slot;
        JobRegs j;
        j.load(jobs + id);
        const bool valid = idx || job_fields_valid(j, P.n_refs);
        const bool covered = fb && (P.fb_jobs ? P.fb_jobs[id] != 0 : full_kernel_eligible(j, (int)sizeof(PIX)));
        if (valid && (idx || group_job_class(j, covered, P.lane_on) == (LPJ == 4 ? LANE_CLASS : ROW_CLASS)))
        {
            const PlaneDesc &enc = P.planes[P.enc_plane];
            const PlaneDesc &ref = P.planes[P.ref_planes[j.ref_idx]];
            QuadWave<PIX, LPJ> W;
            W.bind(enc, ref, j.pu_x, j.pu_y, j.w, j.h, P.cost[j.qp]);
            Search<LaneCfg, QuadWave<PIX, LPJ>> S(W, j);
            int ox, oy;
            const FullBest *f = covered ? fb + (size_t)id * fb_stride : nullptr;
            const int cost = S.run(f, fb_stride, ox, oy);
            if ((threadIdx.x & (LPJ - 1)) == 0)
            {
                x265amd_me_result r;
                r.mvx = (int16_t)ox;
                r.mvy = (int16_t)oy;
                r.cost = cost;
                out[id] = r;
                n_evals = W.evals;
                if (P.hist)
                    P.hist[id] = n_evals;
            }
        }
    }
    const int total = wave_sum((int)n_evals);
    if (threadIdx.x == 0 && total)
        atomicAdd(evals + (blockIdx.x % EVAL_SLOTS) * 16, (unsigned long long)(unsigned)total);
}

} // namespace xa
