"""Motion-estimation job descriptors and builders.

``JOB_DTYPE`` / ``RESULT_DTYPE`` are the numpy images of ``x265amd_me_job`` /
``x265amd_me_result`` (include/x265_amd.h).  A job describes one
``MotionEstimate::motionEstimate`` call of the reference completely
(reference: source/encoder/motion.h:72-95).
"""
from __future__ import annotations

import numpy as np

MAX_MVC = 4

DIA, HEX, UMH, STAR, FULL = 0, 1, 2, 3, 4
BIPRED = 5      # the search of TEncSearch::xMotionEstimation (exhaustive SAD scan + two 9-point SATD refinements)
SAD, SATD, SA8D = 0, 1, 2

JOB_DTYPE = np.dtype([
    ("pu_x", "<i4"), ("pu_y", "<i4"),
    ("w", "<i2"), ("h", "<i2"),
    ("mvmin_x", "<i2"), ("mvmin_y", "<i2"), ("mvmax_x", "<i2"), ("mvmax_y", "<i2"),
    ("qmvp_x", "<i2"), ("qmvp_y", "<i2"),
    ("merange", "<i2"),
    ("method", "u1"), ("subme", "u1"), ("qp", "u1"), ("numc", "u1"),
    ("ref_idx", "u1"), ("icost", "u1"),
    ("mvc", "<i2", (MAX_MVC, 2)),
], align=True)
assert JOB_DTYPE.itemsize == 48

RESULT_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("cost", "<i4")], align=True)
assert RESULT_DTYPE.itemsize == 8

DIMS = (4, 8, 12, 16, 24, 32, 48, 64)


def partition_from_sizes(w: int, h: int) -> int:
    """Partitions enum value (reference: source/common/primitives.cpp:40-54)."""
    return (DIMS.index(w) << 3) + DIMS.index(h)


def search_range(pic_w: int, pic_h: int, cu_x, cu_y, qmvp_x, qmvp_y, merange: int, max_cu: int = 64):
    """mvmin/mvmax (full-pel, inclusive) as the encoder derives them for a CU at (cu_x, cu_y)
    and q-pel predictor (reference: TEncSearch::xSetSearchRange, TEncSearch.cpp:3007-3024, with
    TComDataCU::clipMv, TComDataCU.cpp:2724-2736; frame-parallel row clamp not applied).
    Vectorised over numpy arrays."""
    cu_x = np.asarray(cu_x, dtype=np.int64)
    cu_y = np.asarray(cu_y, dtype=np.int64)
    xmax = (pic_w + 8 - cu_x - 1) << 2
    xmin = (-max_cu - 8 - cu_x + 1) << 2
    ymax = (pic_h + 8 - cu_y - 1) << 2
    ymin = (-max_cu - 8 - cu_y + 1) << 2

    def clipx(v):
        return np.minimum(xmax, np.maximum(xmin, v))

    def clipy(v):
        return np.minimum(ymax, np.maximum(ymin, v))

    px = clipx(np.asarray(qmvp_x, dtype=np.int64))
    py = clipy(np.asarray(qmvp_y, dtype=np.int64))
    d = merange << 2
    return (clipx(px - d) >> 2, clipy(py - d) >> 2, clipx(px + d) >> 2, clipy(py + d) >> 2)


def square_pu_grid(pic_w: int, pic_h: int, size: int):
    """Top-left corners of all size x size PUs of the CTU quadtree that lie fully inside the picture."""
    xs = np.arange(0, pic_w - size + 1, size)
    ys = np.arange(0, pic_h - size + 1, size)
    gx, gy = np.meshgrid(xs, ys)
    return gx.ravel().astype(np.int32), gy.ravel().astype(np.int32)


def make_jobs(pic_w: int, pic_h: int, pu_x, pu_y, w, h, *, method: int, subme: int, merange: int, qp: int,
              qmvp=(0, 0), mvc=None, ref_idx=0, icost: int = SAD, cu_x=None, cu_y=None) -> np.ndarray:
    """Job array for PUs at (pu_x, pu_y) of size (w, h) (scalars or arrays), with the encoder's
    search-range rule centred on ``qmvp`` (scalar pair or arrays)."""
    pu_x = np.atleast_1d(np.asarray(pu_x, dtype=np.int32))
    n = pu_x.shape[0]
    jobs = np.zeros(n, dtype=JOB_DTYPE)
    jobs["pu_x"] = pu_x
    jobs["pu_y"] = pu_y
    jobs["w"] = w
    jobs["h"] = h
    jobs["qmvp_x"] = qmvp[0]
    jobs["qmvp_y"] = qmvp[1]
    cx = jobs["pu_x"] if cu_x is None else cu_x
    cy = jobs["pu_y"] if cu_y is None else cu_y
    x0, y0, x1, y1 = search_range(pic_w, pic_h, cx, cy, jobs["qmvp_x"], jobs["qmvp_y"], merange)
    jobs["mvmin_x"], jobs["mvmin_y"], jobs["mvmax_x"], jobs["mvmax_y"] = x0, y0, x1, y1
    jobs["merange"] = merange
    jobs["method"] = method
    jobs["subme"] = subme
    jobs["qp"] = qp
    jobs["ref_idx"] = ref_idx
    jobs["icost"] = icost
    if mvc is not None:
        mvc = np.asarray(mvc, dtype=np.int16)
        if mvc.ndim == 2:
            mvc = np.broadcast_to(mvc, (n,) + mvc.shape)
        k = mvc.shape[1]
        jobs["numc"] = k
        jobs["mvc"][:, :k, :] = mvc
    return jobs


def frame_jobs(pic_w: int, pic_h: int, sizes=(64, 32, 16, 8), **kw) -> np.ndarray:
    """All square PUs of the given sizes, largest first (the whole-frame job list of the
    full-search and pattern-search benchmarks)."""
    parts = []
    for s in sizes:
        x, y = square_pu_grid(pic_w, pic_h, s)
        parts.append(make_jobs(pic_w, pic_h, x, y, s, s, **kw))
    return np.concatenate(parts)


def full_search_evals(jobs: np.ndarray) -> int:
    """Number of raster candidates of FULL-search jobs (excludes the <=2 predictor evaluations)."""
    nx = jobs["mvmax_x"].astype(np.int64) - jobs["mvmin_x"] + 1
    ny = jobs["mvmax_y"].astype(np.int64) - jobs["mvmin_y"] + 1
    return int((np.maximum(nx, 0) * np.maximum(ny, 0)).sum())


def cu_partitions(n2: int):
    """(ox, oy, w, h) of every PU an n2 x n2 CU can be cut into (reference: the PartSize loop of
    TEncCu::xCompressCU, Lib/TLibEncoder/TEncCu.cpp; SIZE_2Nx2N, 2NxN, Nx2N, and for CUs of 16 pixels and
    more the asymmetric 2NxnU / 2NxnD / nLx2N / nRx2N): 13 PUs, 5 for an 8x8 CU."""
    h2, q = n2 // 2, n2 // 4
    parts = [(0, 0, n2, n2), (0, 0, n2, h2), (0, h2, n2, h2), (0, 0, h2, n2), (h2, 0, h2, n2)]
    if n2 >= 16:
        parts += [(0, 0, n2, q), (0, q, n2, n2 - q), (0, 0, n2, n2 - q), (0, n2 - q, n2, q),
                  (0, 0, q, n2), (q, 0, n2 - q, n2), (0, 0, n2 - q, n2), (n2 - q, 0, q, n2)]
    return parts


def all_partition_jobs(pic_w: int, pic_h: int, cu_sizes=(64, 32, 16, 8), **kw) -> np.ndarray:
    """Every PU of every CU of the quadtree that lies fully inside the picture: 593 PUs per 64x64 CTU
    (13 + 4 * 13 + 16 * 13 + 64 * 5) in 24 shapes from 64x64 down to 8x4 / 4x8."""
    parts = []
    for n2 in cu_sizes:
        x, y = square_pu_grid(pic_w, pic_h, n2)
        for (ox, oy, w, h) in cu_partitions(n2):
            parts.append(make_jobs(pic_w, pic_h, x + ox, y + oy, w, h, **kw))
    return np.concatenate(parts)
