// Bundles of the hierarchical exhaustive kernel (me_hier.h) formed ON THE DEVICE, for job arrays the host never sees
// (x265amd_me_search_batch_device and its allocation-free form).
//
// The host planner (plan_hier in x265_amd.hip) walks the job array; here the same decisions are taken by four small
// kernels over a table of the picture's 64x64 cells (per reference plane):
//   k_plan_clear    resets the table, the bundle flags and the per-job result keys;
//   k_plan_slots    per job: the smallest index of the FULL-search SAD jobs on each square of a cell's quadtree (a second
//                   job on the same square stays outside);
//   k_plan_bundles  per cell: the jobs with the QP of the cell's first job (a bundle shares one MV-cost table), union
//                   window, per-job predictors / ranges (HierBundle::pred), the same profit and size tests as the host, the
//                   slabs as work items, and the flags of the jobs taken.
//   k_plan_scan     the running count of work items over the cells.
// Bundle c lives at cell c of the table; the exhaustive kernel is launched over cells x PLAN_SLABS workgroups, each finds its
// work item from the running counts (in the XCD-interleaved order of the planned lists) and those beyond the last one, all at
// the end of the launch, leave at once.  The slabs of a job meet in ONE 64-bit key per job
// (cost << 32 | row << 16 | column, raster order among equal costs) by atomic minimum, which k_make_items turns into
// the FullBest the search kernels read - no per-slab slots, so the scratch stays one entry per job.
// Unlike the host planner, jobs of a cell whose windows lie far apart are not split into several bundles: the cell
// then fails the profit test and its jobs are scanned one by one (k_full_items), as all of them were before.
#pragma once

#include "me_hier.h"

namespace xa {


struct DevPlan
{
    uint32_t *slots;            // [cells][HIER_PUS]
    HierBundle *bundles;        // [cells]
    int32_t *offsets;           // [cells + 1]: work items of the cells before cell c; [cells]: all
    unsigned char *in_bundle;   // [n]
    unsigned long long *keys;   // [n]: the FullBest array of the call, as keys while the exhaustive kernel runs
    int32_t cells_x, cells_y, n_cells;      // n_cells = cells_x * cells_y * reference planes
};

__device__ __forceinline__ int plan_zidx(int cx, int cy)
{
    return (cx & 1) | ((cy & 1) << 1) | ((cx & 2) << 1) | ((cy & 2) << 2) | ((cx & 4) << 2) | ((cy & 4) << 3);
}

// cell and square slot of a job the hierarchical kernel could take, or -1
__device__ __forceinline__ int plan_job_cell(const BatchParams &P, const DevPlan &D, const x265amd_me_job &j, int &slot)
{
    if (!job_fields_valid(j, P.n_refs) || !full_kernel_eligible(j, 1) || j.method != X265AMD_FULL || j.w != j.h ||
        !(j.w == 8 || j.w == 16 || j.w == 32 || j.w == 64) || j.pu_x < 0 || j.pu_y < 0)
        return -1;
    const int gx = j.pu_x >> 6, gy = j.pu_y >> 6;
    if (gx >= D.cells_x || gy >= D.cells_y || ((j.pu_x | j.pu_y) & (j.w - 1)))
        return -1;
    const int z = plan_zidx((j.pu_x & 63) >> 3, (j.pu_y & 63) >> 3);
    slot = j.w == 8 ? z : j.w == 16 ? 64 + (z >> 2) : j.w == 32 ? 80 + (z >> 4) : 84;
    return (j.ref_idx * D.cells_y + gy) * D.cells_x + gx;
}

__global__ void k_plan_clear(DevPlan D, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, step = gridDim.x * blockDim.x;
    for (int k = i; k < D.n_cells * HIER_PUS; k += step)
        D.slots[k] = ~0u;
    for (int k = i; k < n; k += step)
    {
        D.in_bundle[k] = 0;
        D.keys[k] = ~0ull;
    }
}

__global__ void k_plan_slots(BatchParams P, DevPlan D, const x265amd_me_job *jobs, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    int slot;
    const int cell = plan_job_cell(P, D, jobs[i], slot);
    if (cell >= 0)
        atomicMin(D.slots + (size_t)cell * HIER_PUS + slot, (uint32_t)i);
}

template <class T, class F>
__device__ __forceinline__ T plan_wave_reduce(T v, F f)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1)
        v = f(v, (T)__shfl_xor(v, m));
    return v;
}

// one wavefront per cell: lane l looks at the squares l and l + 64 of the cell's quadtree
__global__ void __launch_bounds__(64) k_plan_bundles(BatchParams P, DevPlan D, const x265amd_me_job *jobs, int lds_limit)
{
    const int cell = blockIdx.x, lane = threadIdx.x;
    if (cell >= D.n_cells)
        return;
    HierBundle &B = D.bundles[cell];
    if (lane == 0)
    {
        B.nxw = 0;                              // "no bundle" until every test has passed
        B.n_slabs = 0;
    }
    const uint32_t *slots = D.slots + (size_t)cell * HIER_PUS;
    auto imin = [](int a, int b) { return a < b ? a : b; };
    auto imax = [](int a, int b) { return a > b ? a : b; };
    int taken = 0, first = INT32_MAX;
    int x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
    int qx0 = INT32_MAX, qx1 = INT32_MIN, qy0 = INT32_MAX, qy1 = INT32_MIN;
    int ulo_x = INT32_MAX, uhi_x = INT32_MIN, ulo_y = INT32_MAX, uhi_y = INT32_MIN;
    double old_work = 0;
    unsigned long long cells = 0;
    uint32_t mine[2] = { ~0u, ~0u };
    for (int r = 0; r < 2; r++)
    {
        const int k = lane + 64 * r;
        if (k < HIER_PUS && slots[k] != ~0u)
        {
            mine[r] = slots[k];
            first = imin(first, (int)mine[r]);
        }
    }
    first = plan_wave_reduce(first, imin);
    if (first == INT32_MAX)
        return;
    const int cell_qp = jobs[first].qp;
    for (int r = 0; r < 2; r++)
    {
        if (mine[r] != ~0u && jobs[mine[r]].qp != cell_qp)
            mine[r] = ~0u;                      // (scanned on its own)
        if (mine[r] == ~0u)
            continue;
        const x265amd_me_job &j = jobs[mine[r]];
        taken++;
        x0 = imin(x0, (int)j.mvmin_x); y0 = imin(y0, (int)j.mvmin_y);
        x1 = imax(x1, (int)j.mvmax_x); y1 = imax(y1, (int)j.mvmax_y);
        const int qx = j.qmvp_x >> 2, qy = j.qmvp_y >> 2;
        qx0 = imin(qx0, qx); qx1 = imax(qx1, qx); qy0 = imin(qy0, qy); qy1 = imax(qy1, qy);
        ulo_x = imin(ulo_x, j.mvmin_x - qx); uhi_x = imax(uhi_x, j.mvmax_x - qx);
        ulo_y = imin(ulo_y, j.mvmin_y - qy); uhi_y = imax(uhi_y, j.mvmax_y - qy);
        old_work += (double)j.w * j.h * (j.mvmax_x - j.mvmin_x + 1) * (j.mvmax_y - j.mvmin_y + 1);
        const int cx = (j.pu_x & 63) >> 3, cy = (j.pu_y & 63) >> 3;
        for (int yy = 0; yy < (j.h >> 3); yy++)
            for (int xx = 0; xx < (j.w >> 3); xx++)
                cells |= 1ull << plan_zidx(cx + xx, cy + yy);
    }
    taken = plan_wave_reduce(taken, [](int a, int b) { return a + b; });
    if (taken < 2)
        return;
    x0 = plan_wave_reduce(x0, imin); y0 = plan_wave_reduce(y0, imin);
    x1 = plan_wave_reduce(x1, imax); y1 = plan_wave_reduce(y1, imax);
    qx0 = plan_wave_reduce(qx0, imin); qx1 = plan_wave_reduce(qx1, imax);
    qy0 = plan_wave_reduce(qy0, imin); qy1 = plan_wave_reduce(qy1, imax);
    ulo_x = plan_wave_reduce(ulo_x, imin); uhi_x = plan_wave_reduce(uhi_x, imax);
    ulo_y = plan_wave_reduce(ulo_y, imin); uhi_y = plan_wave_reduce(uhi_y, imax);
    old_work = plan_wave_reduce(old_work, [](double a, double b) { return a + b; });
    cells = plan_wave_reduce(cells, [](unsigned long long a, unsigned long long b) { return a | b; });
    const int nxw = x1 - x0 + 1, nyw = y1 - y0 + 1;
    if ((int64_t)nxw * nyw > (1 << HIER_RIDX_BITS))
        return;
    if ((double)__popcll(cells) * 64.0 * nxw * nyw * 1.25 > old_work)
        return;
    const x265amd_me_job &j0 = jobs[first];
    int pred = 0, ragged = 0, own = 0;         // different predictors / ranges that differ from the window / from [ulo, uhi] around the predictor
    for (int r = 0; r < 2; r++)
    {
        if (mine[r] == ~0u)
            continue;
        const x265amd_me_job &j = jobs[mine[r]];
        const int qx = j.qmvp_x >> 2, qy = j.qmvp_y >> 2;
        pred |= j.qmvp_x != j0.qmvp_x || j.qmvp_y != j0.qmvp_y;
        ragged |= j.mvmin_x != x0 || j.mvmax_x != x1 || j.mvmin_y != y0 || j.mvmax_y != y1;
        own |= j.mvmin_x - qx != ulo_x || j.mvmax_x - qx != uhi_x || j.mvmin_y - qy != ulo_y || j.mvmax_y - qy != uhi_y;
    }
    auto ior = [](int a, int b) { return a | b; };
    pred = plan_wave_reduce(pred, ior); ragged = plan_wave_reduce(ragged, ior); own = plan_wave_reduce(own, ior);
    if (pred && own)
        pred = 2;
    const int lx = nxw + 4 + qx1 - qx0, ly = nyw + 4 + qy1 - qy0;
    const int lxp = (lx + 3) & ~3, lyp = (ly + 3) & ~3;
    if (pred && (int64_t)nxw * nyw + lx > (1 << HIER_RIDX_BITS))
        return;
    // slabs of one lane task per thread (more only when that would exceed the work items of a bundle)
    const int XG = (nxw + 3) >> 2, nd = XG + 16;
    const int pitch = hier_pitch(nd);
    const int nquads = (nyw + HIER_R - 1) / HIER_R, ntasks = nquads * XG;
    int per = HIER_THREADS;
    while ((ntasks + per - 1) / per > PLAN_SLABS)
        per += HIER_THREADS;
    const int fixed = (pred ? 16 * (lxp + lyp) + 16 * (HIER_PUS + 1) + 32 : 2 * (((nxw + 3) & ~3) + ((nyw + 3) & ~3))) +
                      hier_slot_bytes();
    int lds_need = 0;
    for (int t = 0; t < ntasks; t += per)
    {
        const int t_hi = imin(ntasks, t + per);
        const int win_rows = ((t_hi - 1) / XG - t / XG + 1) * HIER_R + 63;
        lds_need = imax(lds_need, 4 * win_rows * pitch + fixed);
    }
    if (lds_need > lds_limit)
        return;
    for (int r = 0; r < 2; r++)
    {
        const int k = lane + 64 * r;
        if (k >= HIER_PUS)
            continue;
        B.job[k] = mine[r] == ~0u ? -1 : (int)mine[r];
        if (mine[r] != ~0u)
            D.in_bundle[mine[r]] = 1;
    }
    if (lane < 21)
        B.rect_end[lane] = 0;
    if (lane == 0)
    {
        const int gx = cell % D.cells_x, gy = (cell / D.cells_x) % D.cells_y;
        B.ctu_x = gx * 64; B.ctu_y = gy * 64;
        B.mvx0 = x0; B.mvy0 = y0; B.nyw = nyw;
        B.qmvp_x = j0.qmvp_x; B.qmvp_y = j0.qmvp_y; B.qp = j0.qp; B.ref_idx = j0.ref_idx;
        B.uniform = !ragged;
        B.metric = X265AMD_SAD;
        B.pred = pred;
        B.dx0 = x0 - qx1; B.dy0 = y0 - qy1; B.lx = lx; B.ly = ly;
        B.ulo_x = ulo_x; B.uhi_x = uhi_x; B.ulo_y = ulo_y; B.uhi_y = uhi_y;
        B.cells = cells;
        B.quads = 0;
        B.n_slabs = (ntasks + per - 1) / per;
        B.per = per;
        B.pitch = pitch;
        B.lds_need = lds_need;
        B.nxw = nxw;
    }
}

// offsets[c] = work items of the bundles in the cells before c (one workgroup; the cells in chunks of its size)
__global__ void __launch_bounds__(256) k_plan_scan(DevPlan D)
{
    __shared__ int part[256];
    __shared__ int run;
    const int tid = threadIdx.x;
    if (tid == 0)
        run = 0;
    __syncthreads();
    for (int base = 0; base < D.n_cells; base += 256)
    {
        const int c = base + tid;
        const int v = c < D.n_cells ? D.bundles[c].n_slabs : 0;
        part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1)       // inclusive scan of the chunk
        {
            const int add = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (c < D.n_cells)
            D.offsets[c] = run + part[tid] - v;
        __syncthreads();
        if (tid == 255)
            run += part[255];
        __syncthreads();
    }
    if (tid == 0)
        D.offsets[D.n_cells] = run;
}

} // namespace xa
