"""GPU parity tests: the HIP path, called through the C ABI, against
(a) the golden vectors generated from the compiled reference and
(b) the CPU restatement (oracle) on seeded inputs.
Bit-exact: integer costs and MVs must be identical."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import golden_cases as G
from x265_amd import jobs as J
from x265_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from x265_amd.api import Engine
    e = Engine(0)
    yield e
    e.close()


def _load(name):
    return np.load(os.path.join(G.GOLDEN_DIR, name), allow_pickle=False)


class PrimAdapter:
    """Drives the per-call primitive table (reference signatures, host pointers)."""

    def __init__(self, bits):
        from x265_amd.api import setup_primitives
        self.p = setup_primitives(0, bits // 8)

    def cmp(self, kind, idx, fenc, fenc_off, fs, fref, fref_off, rs):
        a = fenc.ctypes.data + fenc_off * fenc.itemsize
        b = fref.ctypes.data + fref_off * fref.itemsize
        tab = (self.p.sad, self.p.satd, self.p.sa8d_inter, self.p.sa8d)[kind]
        return tab[idx](a, fs, b, rs)

    def sad_xn(self, n, part, fenc, fenc_off, fref, fref_offs, rs):
        res = (C.c_int * 16)()
        a = fenc.ctypes.data + fenc_off * fenc.itemsize
        ptrs = [fref.ctypes.data + o * fref.itemsize for o in fref_offs]
        if n == 3:
            self.p.sad_x3[part](a, *ptrs, rs, C.addressof(res))
        else:
            self.p.sad_x4[part](a, *ptrs, rs, C.addressof(res))
        return np.array(res[:n], dtype=np.int32)


@pytest.mark.parametrize("bits", [8, 16])
def test_primitive_table_matches_reference_vectors(bits):
    """The reference's TestBench recipe (test/pixelharness.cpp) replayed through the table."""
    gold = _load("prim_u%d.npz" % bits)
    got = G.run_prim_suite(PrimAdapter(bits), bits)
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), k


def test_mvcost_tables_match_reference_vectors(eng):
    gold = _load("mvcost.npz")
    for qp in G.MV_QPS:
        assert np.array_equal(eng.mvcost_table(qp)[1:], gold["qp%d" % qp][1:]), qp
    for p, m, (c, b) in zip(gold["bc_mvp"].tolist(), gold["bc_mv"].tolist(), gold["bc_out"].tolist()):
        assert eng.bitcost(p, m) == b


@pytest.mark.parametrize("bits", [8, 16])
def test_search_matches_reference_vectors(eng, bits):
    gold = _load("search_u%d.npz" % bits)
    jobs = G.search_jobs(bits)
    assert zlib.crc32(jobs.tobytes()) == int(gold["jobs_crc"][0])
    for kind in ("motion", "noise"):
        fr = G.search_frames(bits, kind)
        depth = 8 if bits == 8 else 10
        ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
        got = eng.search(ids[0], [ids[1], ids[2]], jobs)
        for i in ids:
            eng.release_plane(i)
        exp = gold["res_" + kind]
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (kind, len(bad), jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])


def test_full_search_field_matches_reference_vectors(eng):
    gold = _load("field_full.npz")
    jobs = G.field_jobs()
    fr = G.field_frames()
    ids = [eng.upload_plane(f) for f in fr]
    got = eng.search(ids[0], [ids[1]], jobs)
    for i in ids:
        eng.release_plane(i)
    bad = np.nonzero(got != gold["res"])[0]
    assert bad.size == 0, (len(bad), jobs[bad[:3]], got[bad[:3]], gold["res"][bad[:3]])


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 10), (16, 8)])
def test_random_jobs_match_oracle(eng, oracle, bits, depth):
    """Randomised jobs (all methods, sub-pel levels, partitions, extension cost kinds)."""
    W, H = 352, 288
    rng = np.random.default_rng(555 + bits + depth)
    for kind in ("motion", "noise"):
        fr = synth.make_sequence(W, H, 2, seed=91 + bits, bit_depth=depth if bits == 16 else 8, kind=kind)
        parts = []
        for _ in range(400):
            w, h = int(rng.choice(J.DIMS)), int(rng.choice(J.DIMS))
            method = int(rng.integers(0, 5))
            if method == J.UMH and w == 4 and h == 4:
                continue
            merange = int(rng.integers(2, 20)) if method == J.FULL else int(rng.integers(2, 70))
            px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
            py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
            k = int(rng.integers(0, 5))
            mvc = rng.integers(-300, 300, (k, 2)) if k else None
            icost = int(rng.choice([0, 0, 0, 1, 2]))
            parts.append(J.make_jobs(W, H, [px], [py], w, h, method=method, subme=int(rng.integers(0, 8)),
                                     merange=merange, qp=int(rng.integers(0, 52)),
                                     qmvp=(int(rng.integers(-250, 250)), int(rng.integers(-250, 250))), mvc=mvc,
                                     icost=icost))
        jobs = np.concatenate(parts)
        exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=depth)
        ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
        eng.eval_count(reset=True)
        got = eng.search(ids[0], [ids[1]], jobs)
        gpu_evals = eng.eval_count()
        for i in ids:
            eng.release_plane(i)
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (kind, len(bad), jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
        assert gpu_evals == evals, (gpu_evals, evals)


@pytest.mark.parametrize("bits", [8, 16])
def test_cost_batch_matches_oracle(eng, oracle, bits):
    """Batched sad / satd / sa8d_inter on plane-resident blocks, incl. blocks in the padded border."""
    W, H = 320, 192
    depth = 8 if bits == 8 else 10
    fr = synth.make_sequence(W, H, 2, seed=5 + bits, bit_depth=depth, kind="noise")
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    rng = np.random.default_rng(bits)
    M = synth.MARGIN
    stride = fr[0].shape[1]
    from x265_amd.api import PAIR_DTYPE
    for kind in (0, 1, 2):
        for (w, h) in ((4, 4), (8, 8), (16, 16), (64, 64), (12, 16), (24, 32), (48, 64), (64, 16), (8, 4), (32, 24), (16, 8)):
            n = 40
            pairs = np.zeros(n, dtype=PAIR_DTYPE)
            pairs["enc_plane"] = ids[0]
            pairs["ref_plane"] = ids[1]
            pairs["enc_x"] = rng.integers(-M, W + M - w + 1, n)
            pairs["enc_y"] = rng.integers(-M, H + M - h + 1, n)
            pairs["ref_x"] = rng.integers(-M, W + M - w + 1, n)
            pairs["ref_y"] = rng.integers(-M, H + M - h + 1, n)
            pairs[0] = (ids[0], -M, -M, ids[1], W + M - w, H + M - h)     # extreme corners
            got = eng.cost_batch(kind, w, h, pairs)
            part = J.partition_from_sizes(w, h)
            for i in range(n):
                eo = (int(pairs["enc_y"][i]) + M) * stride + int(pairs["enc_x"][i]) + M
                ro = (int(pairs["ref_y"][i]) + M) * stride + int(pairs["ref_x"][i]) + M
                exp = oracle.cmp(kind, part, fr[0].reshape(-1), eo, stride, fr[1].reshape(-1), ro, stride)
                assert got[i] == exp, (kind, w, h, i, got[i], exp)
    for i in ids:
        eng.release_plane(i)


def test_full_search_properties_at_full_size(eng, oracle):
    """1080p, FULL +-64: size-independent checks -- (1) a planted pure translation is found with
    SAD 0, (2) the winner's cost is reproduced by the batched SAD primitive + mvcost, (3) a sample
    of PUs equals the oracle."""
    W, H = 1920, 1080
    base = synth.base_frame(W, H, seed=99)
    shift = (7, -5)
    ref_img = np.roll(base, (-shift[1], -shift[0]), axis=(0, 1))   # ref(x,y) = enc(x - sx, y - sy)... planted MV = shift
    enc = synth.pad_plane(base)
    ref = synth.pad_plane(ref_img)
    ids = [eng.upload_plane(enc), eng.upload_plane(ref)]
    jobs = J.frame_jobs(W, H, sizes=(64, 16), method=J.FULL, subme=0, merange=64, qp=32)
    # interior PUs only (np.roll wraps at the borders)
    keep = (jobs["pu_x"] >= 128) & (jobs["pu_x"] + jobs["w"] <= W - 128) & (jobs["pu_y"] >= 128) & (jobs["pu_y"] + jobs["h"] <= H - 128)
    jobs = jobs[keep][::7]
    got = eng.search(ids[0], [ids[1]], jobs)
    # enc(x,y) == ref(x - sx, y - sy) -> best MV = (-sx, -sy) full-pel, SATD 0, cost = mvcost only
    tab = eng.mvcost_table(32).astype(np.int64)
    expect_cost = tab[32768 + (-shift[0] * 4)] + tab[32768 + (-shift[1] * 4)]
    assert np.all(got["mvx"] == -shift[0] * 4) and np.all(got["mvy"] == -shift[1] * 4)
    assert np.all(got["cost"] == expect_cost)
    sample = jobs[:: max(1, len(jobs) // 12)]
    exp, _ = oracle.search(enc, [ref], sample, threads=8)
    got_s = eng.search(ids[0], [ids[1]], sample)
    assert np.array_equal(exp, got_s)
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits", [8, 16])
def test_exhaustive_ties_and_wide_ranges(eng, oracle, bits):
    """FULL search where every candidate has the same SAD (flat and periodic pictures: the winner is
    decided by the MV cost and, on equal cost, by raster order), with ranges wide enough
    (201 x 201 candidates) to leave the small-PU fast path of the 8-bit exhaustive kernel, and on a
    moving textured picture at +-64; 8-bit and 10-bit-in-16-bit planes."""
    W, H = 640, 480
    depth = 8 if bits == 8 else 10
    dt = np.uint8 if bits == 8 else np.uint16
    scale = 1 if bits == 8 else 4
    flat = np.full((H, W), 77 * scale, dtype=dt)
    yy, xx = np.mgrid[0:H, 0:W]
    periodic = ((((xx // 4) + (yy // 4)) % 2 * 90 + 40) * scale).astype(dt)      # 8-pixel period: many exact ties
    seq = synth.make_sequence(W, H, 2, seed=17, bit_depth=depth)
    cases = [(synth.pad_plane(flat, depth), synth.pad_plane(flat, depth)),
             (synth.pad_plane(periodic, depth), synth.pad_plane(periodic, depth)),
             (seq[1], seq[0])]
    for enc, ref in cases:
        ids = [eng.upload_plane(enc, bit_depth=depth), eng.upload_plane(ref, bit_depth=depth)]
        parts = []
        for (w, h, mr, qmvp) in ((8, 8, 64, (0, 0)), (8, 8, 100, (9, -14)), (4, 8, 100, (0, 0)), (16, 16, 64, (-21, 6)),
                                 (16, 8, 100, (3, 3)), (32, 32, 64, (0, 0)), (64, 64, 64, (13, 2)), (24, 32, 64, (0, 0)),
                                 (48, 64, 33, (-7, 7)), (12, 16, 20, (0, 0))):
            x, y = J.square_pu_grid(W, H, max(w, h))
            sel = slice(None, None, max(1, len(x) // 24))
            parts.append(J.make_jobs(W, H, x[sel], y[sel], w, h, method=J.FULL, subme=0, merange=mr, qp=30, qmvp=qmvp))
        jobs = np.concatenate(parts)
        exp, _ = oracle.search(enc, [ref], jobs, threads=8, bit_depth=depth)
        got = eng.search(ids[0], [ids[1]], jobs)
        assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:5]
        jl = eng.make_joblist(jobs, bytes_per_px=bits // 8)
        got2 = jl.run(ids[0], [ids[1]])
        jl.close()
        assert np.array_equal(got2, exp), np.nonzero(got2 != exp)[0][:5]
        for i in ids:
            eng.release_plane(i)


@pytest.mark.parametrize("kind", ["motion", "noise", "flat"])
def test_whole_cell_exhaustive_search_matches_oracle(eng, oracle, kind):
    """FULL-search jobs on every square of the 64x64 quadtree (8/16/32/64) share one pass over the window
    (8x8 SADs summed up the tree): whole frames with partial cells at the right / bottom edges (clipped,
    per-PU different ranges), a non-zero predictor, a subset of the levels, a range of 257 candidates that
    does not qualify, and flat pictures where every candidate ties and the raster order decides."""
    W, H = 200, 152
    if kind == "flat":
        a = np.full((H, W), 77, np.int32)
        fr = [synth.pad_plane(a), synth.pad_plane(a)]
    else:
        fr = synth.make_sequence(W, H, 2, seed=515, kind=kind)
    ids = [eng.upload_plane(f) for f in fr]
    parts = [J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32),
             J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=1, merange=20, qp=24, qmvp=(-37, 22)),
             J.frame_jobs(W, H, sizes=(32, 8), method=J.FULL, subme=0, merange=13, qp=40, qmvp=(6, 6)),
             J.frame_jobs(W, H, sizes=(64, 16), method=J.BIPRED, subme=0, merange=4, qp=30, qmvp=(3, -5))]
    wide = J.frame_jobs(W, H, sizes=(64, 8), method=J.FULL, subme=0, merange=8, qp=30)[:40:3]
    wide["mvmin_x"], wide["mvmax_x"] = -70, 70               # 141 x 257 candidates: beyond the shared-window kernel's index
    wide["mvmin_y"], wide["mvmax_y"] = -8, 8
    parts.append(wide)
    jobs = np.concatenate(parts)
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8)
    eng.eval_count(reset=True)
    got = eng.search(ids[0], [ids[1]], jobs)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad[:5], jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("kind", ["motion", "flat"])
@pytest.mark.parametrize("spread", [3, 9, 40])
def test_whole_cell_exhaustive_search_with_a_predictor_per_pu(eng, oracle, kind, spread):
    """The PUs of one 64x64 cell each have their OWN predictor (and with it their own window and MV-cost slice), as an
    encoder's AMVP gives them: predictors scattered `spread` quarter-pels around a per-cell vector share the pass over
    the cell's window (spread 3 and 9: all sub-pel phases, windows a few candidates apart, clipped differently at the
    picture edges); spread 40 pulls some windows too far apart to share, and those PUs are scanned on their own."""
    W, H = 200, 152
    if kind == "flat":
        a = np.full((H, W), 91, np.int32)
        fr = [synth.pad_plane(a), synth.pad_plane(a)]
    else:
        fr = synth.make_sequence(W, H, 2, seed=626, kind=kind)
    ids = [eng.upload_plane(f) for f in fr]
    rng = np.random.default_rng(1000 + spread)
    parts = []
    for sizes, merange, qp, subme in (((64, 32, 16, 8), 24, 30, 0), ((16, 8), 64, 37, 1), ((32, 16), 9, 22, 0)):
        for s in sizes:
            x, y = J.square_pu_grid(W, H, s)
            cell = (y // 64) * 8 + x // 64
            base = rng.integers(-60, 61, size=(64, 2))[cell]
            q = base + rng.integers(-spread, spread + 1, size=(x.size, 2))
            parts.append(J.make_jobs(W, H, x, y, s, s, method=J.FULL, subme=subme, merange=merange, qp=qp,
                                     qmvp=(q[:, 0].astype(np.int16), q[:, 1].astype(np.int16))))
    jobs = np.concatenate(parts)
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8)
    eng.eval_count(reset=True)
    got = eng.search(ids[0], [ids[1]], jobs)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad[:5], jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 10), (16, 12)])
@pytest.mark.parametrize("kind", ["motion", "flat"])
def test_whole_cell_exhaustive_hadamard_search_matches_oracle(eng, oracle, bits, depth, kind):
    """FULL search with the SA8D metric on every square of the 64x64 quadtree: one pass over the window
    prices the 8x8 Hadamard cells and sums them up the tree (8- and 10-bit planes; 12-bit planes take the
    general kernel).  Whole frames with partial cells at the edges (per-PU different clipped ranges), +-16 and
    a non-zero predictor, a sub-pel round behind the scan, single-level lists, +-64 on one row of cells
    (several slabs per window), rectangles mixed in, and flat pictures where only the MV cost and the
    raster order decide."""
    W, H = 200, 152
    if kind == "flat":
        a = np.full((H, W), 77 * (1 << (depth - 8)), np.int32)
        fr = [synth.pad_plane(a, depth), synth.pad_plane(a, depth)]
    else:
        fr = synth.make_sequence(W, H, 2, seed=77, bit_depth=depth)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    kw = dict(method=J.FULL, icost=J.SA8D)
    parts = [J.frame_jobs(W, H, sizes=(64, 32, 16, 8), subme=0, merange=16, qp=27, **kw),
             J.frame_jobs(W, H, sizes=(64, 32, 16, 8), subme=2, merange=9, qp=24, qmvp=(-37, 22), **kw),
             J.frame_jobs(W, H, sizes=(8,), subme=0, merange=12, qp=40, qmvp=(6, 6), **kw),
             J.frame_jobs(W, H, sizes=(32, 16), subme=0, merange=64, qp=30, **kw)[:12]]
    x, y = J.square_pu_grid(W, H, 32)
    parts.append(J.make_jobs(W, H, x[::3], y[::3], 32, 16, subme=0, merange=10, qp=27, **kw))
    # every rectangular / asymmetric partition of the CUs of 16, 32 and 64 pixels (those made of 8x8 cells share the pass;
    # the 4- and 12-pixel ones of the 16-pixel CUs are priced with SATD by the reference and take the general kernel)
    rx, ry, rw, rh = [], [], [], []
    for n2 in (16, 32, 64):
        q = n2 // 4
        for cy in range(0, H - n2 + 1, n2):
            for cx in range(0, W - n2 + 1, n2):
                for (ox, oy, w, h) in ((0, 0, n2, n2 // 2), (0, n2 // 2, n2, n2 // 2), (0, 0, n2 // 2, n2), (n2 // 2, 0, n2 // 2, n2),
                                       (0, 0, n2, q), (0, q, n2, n2 - q), (0, 0, n2, n2 - q), (0, n2 - q, n2, q),
                                       (0, 0, q, n2), (q, 0, n2 - q, n2), (0, 0, n2 - q, n2), (n2 - q, 0, q, n2)):
                    rx.append(cx + ox); ry.append(cy + oy); rw.append(w); rh.append(h)
    rx, ry, rw, rh = (np.array(a) for a in (rx, ry, rw, rh))
    for (w, h) in sorted(set(zip(rw.tolist(), rh.tolist()))):
        sel = (rw == w) & (rh == h)
        parts.append(J.make_jobs(W, H, rx[sel], ry[sel], w, h, subme=0, merange=7, qp=27, **kw))
        parts.append(J.make_jobs(W, H, rx[sel][::2], ry[sel][::2], w, h, method=J.FULL, subme=0, merange=11, qp=27, qmvp=(-3, 2)))
    # the SAD metric through the same pass (16-bit planes): all levels, single small / large levels, bidirectional refinement
    parts += [J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=24, qp=33, qmvp=(5, -9)),
              J.frame_jobs(W, H, sizes=(8,), method=J.FULL, subme=1, merange=16, qp=22),
              J.frame_jobs(W, H, sizes=(32,), method=J.FULL, subme=0, merange=16, qp=22, qmvp=(1, 1)),
              J.frame_jobs(W, H, sizes=(32, 16), method=J.BIPRED, subme=0, merange=4, qp=30, qmvp=(3, -5))]
    jobs = np.concatenate(parts)
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=depth)
    eng.eval_count(reset=True)
    got = eng.search(ids[0], [ids[1]], jobs)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad[:5], jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 10)])
@pytest.mark.parametrize("icost", [J.SA8D, J.SAD])
def test_every_partition_of_the_ctus_through_the_shared_pass(eng, oracle, bits, depth, icost):
    """All 593 PUs of every CTU (24 shapes, 64x64 down to 8x4 / 4x8, the asymmetric ones included) in one list, exhaustive
    search: squares and partitions made of 8x8 cells with sa8d_inter's 8x8 / 16x16 recipe, those with 4- or 12-pixel
    sides with its SATD recipe (4x4 tiles), all from one pass over each cell's window; the same with the SAD metric.
    A picture with partial CTUs at the right and bottom edges (per-PU clipped ranges), a centred +-9 window, and a
    +-5 window around a non-zero predictor with a sub-pel round behind it."""
    W, H = 152, 136
    fr = synth.make_sequence(W, H, 2, seed=123, bit_depth=depth)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    parts = [J.all_partition_jobs(W, H, method=J.FULL, subme=0, merange=9, qp=27, icost=icost),
             J.all_partition_jobs(W, H, cu_sizes=(16, 8), method=J.FULL, subme=1, merange=5, qp=33, icost=icost, qmvp=(-9, 14))]
    jobs = np.concatenate(parts)
    assert len(J.all_partition_jobs(64, 64, method=J.FULL, subme=0, merange=1, qp=27)) == 593
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=depth)
    eng.eval_count(reset=True)
    jl = eng.make_joblist(jobs, bytes_per_px=bits // 8)
    got = jl.run(ids[0], [ids[1]])
    jl.close()
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (len(bad), bad[:5], jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 10)])
@pytest.mark.parametrize("icost", [J.SA8D, J.SAD])
def test_every_partition_with_a_predictor_per_pu(eng, oracle, bits, depth, icost):
    """The 593 PUs of every CTU, each with its own predictor a few quarter-pels around a per-CTU vector (so its own
    window and MV-cost slice), still share the pass over the CTU's window."""
    W, H = 152, 136
    fr = synth.make_sequence(W, H, 2, seed=321, bit_depth=depth)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    jobs = np.concatenate([J.all_partition_jobs(W, H, method=J.FULL, subme=0, merange=9, qp=27, icost=icost),
                           J.all_partition_jobs(W, H, cu_sizes=(16, 8), method=J.FULL, subme=1, merange=5, qp=33, icost=icost)])
    rng = np.random.default_rng(77)
    ctu = (jobs["pu_y"] // 64) * 4 + jobs["pu_x"] // 64
    q = rng.integers(-30, 31, size=(16, 2))[ctu] + rng.integers(-6, 7, size=(jobs.size, 2))
    for k, (mr, sel) in enumerate(((9, jobs["subme"] == 0), (5, jobs["subme"] == 1))):
        jobs["qmvp_x"][sel], jobs["qmvp_y"][sel] = q[sel, 0], q[sel, 1]
        r = J.search_range(W, H, jobs["pu_x"][sel], jobs["pu_y"][sel], q[sel, 0], q[sel, 1], mr)
        jobs["mvmin_x"][sel], jobs["mvmin_y"][sel], jobs["mvmax_x"][sel], jobs["mvmax_y"][sel] = r
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=depth)
    eng.eval_count(reset=True)
    jl = eng.make_joblist(jobs, bytes_per_px=bits // 8)
    got = jl.run(ids[0], [ids[1]])
    jl.close()
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (len(bad), bad[:5], jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("depth", [12, 14])
def test_shared_pass_limits_fall_back_to_the_other_kernels(eng, oracle, depth):
    """What the candidate-per-lane exhaustive kernel does not take must come out right through the other kernels:
    16-bit planes deeper than its packed 16-bit arithmetic allows (SA8D: 10 bits, SAD: 12 bits), more rectangular PUs in
    one 64x64 cell than a bundle holds (the same partition listed many times), rectangles with a +-40 window (small
    workgroups although the window is large), PUs straddling two cells, and 4- / 12-pixel sides."""
    W, H = 136, 136
    rng = np.random.default_rng(depth)
    a = rng.integers(0, 1 << depth, (H, W)).astype(np.int32)
    b = np.roll(a, (2, -3), (0, 1))
    b[::7, ::5] = (1 << depth) - 1
    fr = [synth.pad_plane(a, depth), synth.pad_plane(b, depth)]
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    parts = []
    for icost in (J.SAD, J.SA8D):
        kw = dict(method=J.FULL, subme=0, qp=30, icost=icost)
        parts += [J.frame_jobs(W, H, sizes=(64, 32, 16, 8), merange=6, **kw),
                  J.make_jobs(W, H, [0] * 150, [0] * 150, 16, 8, merange=5, **kw),          # one partition 150 times
                  J.make_jobs(W, H, [0, 16, 32, 64], [8, 0, 32, 64], 32, 8, merange=40, **kw),
                  J.make_jobs(W, H, [48, 56, 40], [48, 8, 56], 32, 32, merange=6, **kw),     # straddle cell borders
                  J.make_jobs(W, H, [0, 16, 64], [0, 32, 64], 16, 12, merange=6, **kw),
                  J.make_jobs(W, H, [0, 16, 64], [0, 32, 64], 4, 16, merange=6, **kw)]
    jobs = np.concatenate(parts)
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=depth)
    eng.eval_count(reset=True)
    got = eng.search(ids[0], [ids[1]], jobs)
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (bad[:5], jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


def test_device_resident_and_planned_paths_agree(eng, oracle):
    """The three ways in (host batch, raw device job array, planned job list) give identical,
    oracle-exact results; the job list mixes FULL jobs of every size class with pattern searches."""
    import torch
    W, H = 416, 240
    fr = synth.make_sequence(W, H, 3, seed=321)
    ids = [eng.upload_plane(f) for f in fr]
    rng = np.random.default_rng(8)
    parts = [J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=2, merange=40, qp=30, qmvp=(5, -9)),
             J.frame_jobs(W, H, sizes=(16,), method=J.FULL, subme=0, merange=64, qp=22),
             J.frame_jobs(W, H, sizes=(32, 8), method=J.HEX, subme=5, merange=32, qp=37, ref_idx=1)]
    for (w, h) in ((4, 4), (8, 4), (12, 16), (24, 32), (48, 64), (64, 16), (16, 64), (4, 16)):
        x, y = J.square_pu_grid(W, H, max(w, h))
        parts.append(J.make_jobs(W, H, x[::2], y[::2], w, h, method=J.FULL, subme=1, merange=int(rng.integers(3, 30)),
                                 qp=28, qmvp=(int(rng.integers(-40, 40)), int(rng.integers(-40, 40))), ref_idx=int(rng.integers(0, 2))))
    jobs = np.concatenate(parts)
    exp, _ = oracle.search(fr[0], [fr[1], fr[2]], jobs, threads=8)
    a = eng.search(ids[0], [ids[1], ids[2]], jobs)
    assert np.array_equal(a, exp), np.nonzero(a != exp)[0][:5]
    jl = eng.make_joblist(jobs, n_refs=2)
    b = jl.run(ids[0], [ids[1], ids[2]])
    b2 = jl.run(ids[0], [ids[1], ids[2]])          # a plan is reusable
    jl.close()
    assert np.array_equal(b, exp) and np.array_equal(b2, exp)
    dev = torch.device("cuda", 0)
    dj = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
    do = torch.zeros(len(jobs) * 8, dtype=torch.uint8, device=dev)
    eng.search_device(ids[0], [ids[1], ids[2]], dj.data_ptr(), len(jobs), do.data_ptr())
    eng.sync()
    c = do.cpu().numpy().view(J.RESULT_DTYPE)
    assert np.array_equal(c, exp), np.nonzero(c != exp)[0][:5]
    for i in ids:
        eng.release_plane(i)


def test_device_resident_full_frame_is_bundled_on_the_device(eng, oracle):
    """A 1080p FULL +-64 frame (all four square PU levels, a predictor per cell and - second list - per PU) handed over as
    a raw device array: the bundles of the shared-window pass are formed on the device, the results equal the planned
    list's and a sample of the oracle's, and the call takes at most 1.2 x the planned list's time (1.3 x with a predictor
    per PU)."""
    import time
    import torch
    W, H = 1920, 1080
    fr = synth.make_sequence(W, H, 2, seed=77)
    ids = [eng.upload_plane(f) for f in fr]
    rng = np.random.default_rng(12)
    dev = torch.device("cuda", 0)
    for spread in (0, 5):
        parts = []
        cells = rng.integers(-16, 17, size=(32 * 17, 2))
        for s in (64, 32, 16, 8):
            x, y = J.square_pu_grid(W, H, s)
            q = cells[(y // 64) * 32 + x // 64] + (rng.integers(-spread, spread + 1, size=(x.size, 2)) if spread else 0)
            parts.append(J.make_jobs(W, H, x, y, s, s, method=J.FULL, subme=0, merange=64, qp=32,
                                     qmvp=(q[:, 0].astype(np.int16), q[:, 1].astype(np.int16))))
        jobs = np.concatenate(parts)
        jl = eng.make_joblist(jobs)
        b = jl.run(ids[0], [ids[1]])
        dj = torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(dev)
        do = torch.zeros(len(jobs) * 8, dtype=torch.uint8, device=dev)
        eng.search_device(ids[0], [ids[1]], dj.data_ptr(), len(jobs), do.data_ptr())
        eng.sync()
        c = do.cpu().numpy().view(J.RESULT_DTYPE)
        assert np.array_equal(c, b), (spread, np.nonzero(c != b)[0][:5])
        sel = np.arange(0, len(jobs), 97)
        exp, _ = oracle.search(fr[0], [fr[1]], jobs[sel], threads=8)
        assert np.array_equal(c[sel], exp), spread
        t_list, t_dev = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            jl.run(ids[0], [ids[1]], fetch=False)
            eng.sync()
            t_list.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            eng.search_device(ids[0], [ids[1]], dj.data_ptr(), len(jobs), do.data_ptr())
            eng.sync()
            t_dev.append(time.perf_counter() - t0)
        print("spread %d: planned list %.3f ms, device array %.3f ms" % (spread, 1e3 * min(t_list), 1e3 * min(t_dev)))
        # (a predictor per PU: the device planner keeps a cell's PUs in one bundle where the host planner may split them)
        assert min(t_dev) <= (1.2 if spread == 0 else 1.3) * min(t_list), (spread, min(t_dev), min(t_list))
        jl.close()
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 10)])
def test_large_pus_alone_and_beside_small_pus(eng, oracle, bits, depth):
    """PUs of more than 1024 pixels run on eight wavefronts when their launch has the GPU to itself and on four
    beside the launches of smaller PUs: both forms, every pattern method, with sub-pel refinement, against the oracle."""
    W, H = 448, 256
    fr = synth.make_sequence(W, H, 2, seed=77 + bits, bit_depth=depth if bits == 16 else 8)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    large, small = [], []
    for m, method in enumerate((J.DIA, J.HEX, J.UMH, J.STAR)):
        for (w, h) in ((64, 64), (64, 32), (32, 64), (48, 64), (64, 48), (64, 24)):
            x, y = J.square_pu_grid(W, H, 64)
            large.append(J.make_jobs(W, H, x, y, w, h, method=method, subme=(5, 2, 7, 3)[m], merange=32, qp=30,
                                     qmvp=(3 * m - 7, 11 - 5 * m), mvc=np.array([[9, -6], [-14, 4]])))
        small.append(J.frame_jobs(W, H, sizes=(32, 16, 8), method=method, subme=(5, 2, 7, 3)[m], merange=32, qp=30))
    for jobs in (np.concatenate(large), np.concatenate(large + small)):
        exp, _ = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=depth)
        jl = eng.make_joblist(jobs, bytes_per_px=bits // 8)
        got = jl.run(ids[0], [ids[1]])
        jl.close()
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (len(jobs), len(bad), jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    for i in ids:
        eng.release_plane(i)


class GpuIp:
    """the stand-alone 8-tap filter entry point (x265amd_ipfilter) behind golden_cases.run_ip_suite"""

    def __init__(self, eng, bits):
        self.eng, self.bpp = eng, bits // 8

    def ipfilter_run(self, kind, src, x0, y0, w, h, frac, depth):
        return self.eng.ipfilter(kind, np.ascontiguousarray(src), x0, y0, w, h, frac, depth, self.bpp)


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 8), (16, 10)])
def test_interpolation_matches_reference_vectors(eng, bits, depth):
    """ipfilter_pp[H_8] / ipfilter_pp[V_8] / ipfilter_ps[H_8] / ipfilter_sp[V_8] of the compiled reference
    (3 phases x 7 block sizes each, incl. the (w+1) x h and w x (h+1) blocks of the half-pel pairs), replayed
    through the device functions the search interpolates with.  Depth 10 includes the reference's
    cast-to-short-before-the-shift wrap-around (common/ipfilter.cpp:129,302)."""
    gold = _load("ipfilter_u%d_d%d.npz" % (bits, depth))
    got = G.run_ip_suite(GpuIp(eng, bits), bits, depth)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert gold[k].dtype == got[k].dtype and np.array_equal(gold[k], got[k]), k


def test_interpolation_hv_chain_matches_oracle(eng, oracle):
    """ps-H over rows -3 .. h+4 into a `short` intermediate, then sp-V: subpelCompare's diagonal case
    (encoder/motion.cpp:1158-1161), on the GPU entry point against the CPU restatement, 8- and 10-bit"""
    for bits, depth in ((8, 8), (16, 10)):
        src, _ = G.ip_source(bits)
        bpp = bits // 8
        for (w, h, xf, yf) in ((16, 16, 1, 3), (64, 32, 2, 2), (8, 8, 3, 1), (12, 16, 1, 1)):
            tmp = eng.ipfilter(2, src, 20, 20 - 3, w, h + 7, xf, depth, bpp)
            got = eng.ipfilter(3, tmp, 0, 3, w, h, yf, depth, bpp)
            etmp = np.zeros((h + 7, w), np.int16)
            oracle.ipfilter(2, src, (20 - 3) * src.shape[1] + 20, src.shape[1], etmp, 0, w, w, h + 7, xf, depth, bpp)
            exp = np.zeros((h, w), src.dtype)
            oracle.ipfilter(3, etmp, 3 * w, w, exp, 0, w, w, h, yf, depth, bpp)
            assert np.array_equal(tmp, etmp) and np.array_equal(got, exp), (bits, w, h, xf, yf)


def test_joblist_reference_count_and_ranges_are_checked(eng):
    """a list made for two references does not run with one; a list whose windows leave the planes' margins is refused"""
    from x265_amd.api import X265AmdError
    fr = synth.make_sequence(128, 128, 3, seed=5)
    ids = [eng.upload_plane(f) for f in fr]
    jobs = J.make_jobs(128, 128, [16, 32], [16, 32], 16, 16, method=J.HEX, subme=1, merange=8, qp=30)
    jobs["ref_idx"] = [0, 1]
    jl = eng.make_joblist(jobs, n_refs=2)
    jl.run(ids[0], [ids[1], ids[2]])
    with pytest.raises(X265AmdError, match="-2"):
        jl.run(ids[0], [ids[1]])
    jl.close()
    bad = jobs.copy()
    bad["ref_idx"] = 0
    bad["mvmin_x"] = -100
    jl = eng.make_joblist(bad)
    with pytest.raises(X265AmdError, match="-5"):
        jl.run(ids[0], [ids[1]])
    jl.close()
    for i in ids:
        eng.release_plane(i)


def test_plane_geometry_query(eng):
    fr = synth.make_sequence(96, 64, 1, seed=2)
    pid = eng.upload_plane(fr[0])
    assert eng.plane_geometry(pid) == (96, 64, synth.MARGIN, synth.MARGIN, 96 + 2 * synth.MARGIN, 1, 8)
    eng.release_plane(pid)


def test_errors_are_reported(eng):
    fr = synth.make_sequence(64, 64, 2, seed=1)
    ids = [eng.upload_plane(f) for f in fr]
    from x265_amd.api import X265AmdError
    bad = J.make_jobs(64, 64, [0], [0], 16, 16, method=J.HEX, subme=1, merange=8, qp=30)
    bad["w"] = 20
    with pytest.raises(X265AmdError):
        eng.search(ids[0], [ids[1]], bad)
    ok = J.make_jobs(64, 64, [0], [0], 16, 16, method=J.HEX, subme=1, merange=8, qp=30)
    with pytest.raises(X265AmdError):
        eng.search(ids[0], [99], ok)
    empty = eng.search(ids[0], [ids[1]], ok[:0])
    assert len(empty) == 0
    for i in ids:
        eng.release_plane(i)


def test_search_window_outside_margins_is_rejected(eng):
    """the host batch call refuses jobs whose bounds (plus overshoot and filter apron) leave the plane"""
    from x265_amd.api import X265AmdError
    fr = synth.make_sequence(128, 128, 2, seed=12)
    ids = [eng.upload_plane(f) for f in fr]
    ok = J.make_jobs(128, 128, [0], [0], 16, 16, method=J.HEX, subme=5, merange=64, qp=30)
    eng.search(ids[0], [ids[1]], ok)                   # the encoder's own rule stays inside (71 + 3 + 4 <= 80)
    bad = ok.copy()
    bad["mvmin_x"] = -75
    with pytest.raises(X265AmdError, match="-5"):
        eng.search(ids[0], [ids[1]], bad)
    bad = ok.copy()
    bad["pu_x"] = 128 + 80 - 8
    with pytest.raises(X265AmdError, match="-5"):
        eng.search(ids[0], [ids[1]], bad)
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits", [8, 16])
def test_bidirectional_refinement_matches_reference_vectors(eng, oracle, bits):
    """X265AMD_BIPRED jobs (exhaustive SAD scan + two 9-point SATD refinements on a modified source
    block) against the vectors produced with the reference's primitives, through the host batch call
    and through a planned job list (exhaustive kernel + general kernel)"""
    import os
    import golden_cases as G
    gold = np.load(os.path.join(G.GOLDEN_DIR, "bipred_u%d.npz" % bits), allow_pickle=False)["res"]
    depth = 8 if bits == 8 else 10
    mod, rf = G.bipred_frames(bits)
    jobs = G.bipred_jobs(bits)
    ids = [eng.upload_plane(mod, bit_depth=depth), eng.upload_plane(rf, bit_depth=depth)]
    got = eng.search(ids[0], [ids[1]], jobs)
    bad = np.nonzero(got != gold)[0]
    assert bad.size == 0, (jobs[bad[:3]], got[bad[:3]], gold[bad[:3]])
    jl = eng.make_joblist(jobs, bytes_per_px=bits // 8)
    got2 = jl.run(ids[0], [ids[1]])
    jl.close()
    assert np.array_equal(got2, gold)
    # mixed with ordinary searches in one batch
    mixed = np.concatenate([jobs[:20], J.make_jobs(G.SEARCH_W, G.SEARCH_H, [64, 128], [32, 64], 16, 16, method=J.HEX, subme=3,
                                                   merange=16, qp=30)])
    got3 = eng.search(ids[0], [ids[1]], mixed)
    exp3, _ = oracle.search(mod, [rf], mixed, bit_depth=depth)
    assert np.array_equal(got3, exp3)
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("bits", [8, 16])
def test_bipred_source_blocks_made_on_the_device(eng, oracle, bits):
    """The block a bidirectional search runs on, ClipY(2 * org - otherPred) (TComYuv::removeHighFreq), made on the device from
    the original picture and the other list's reference at a q-pel MV (all four interpolation cases, MVs that reach into
    the margins), compared pixel for pixel with the same block built from the oracle's filter primitives; then
    X265AMD_BIPRED jobs on the device-made plane against the oracle searching the host-made plane."""
    depth = 8 if bits == 8 else 10
    dt = np.uint8 if bits == 8 else np.uint16
    W, H, M = 256, 160, synth.MARGIN
    fr = synth.make_sequence(W, H, 3, seed=91, bit_depth=depth)         # 0: original, 1: this list's reference, 2: the other list's
    rng = np.random.default_rng(4)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    dst = eng.create_plane(W, H, bytes_per_px=bits // 8, bit_depth=depth)
    host_dst = np.zeros_like(fr[0])
    all_jobs = []
    for (w, h) in ((64, 64), (32, 16), (16, 16), (8, 8), (8, 4), (4, 16)):
        xs = np.arange(0, W - w + 1, w)[:4]
        ys = np.arange(0, H - h + 1, h)[:2]
        gx, gy = (a.ravel() for a in np.meshgrid(xs, ys))
        pairs = np.zeros(len(gx), dtype=api_mc_dtype())
        pairs["enc_plane"], pairs["ref_plane"] = ids[0], ids[2]
        pairs["pu_x"], pairs["pu_y"] = gx, gy
        pairs["qmv_x"] = rng.integers(-40, 40, len(gx))
        pairs["qmv_y"] = rng.integers(-40, 40, len(gx))
        pairs["qmv_x"][0], pairs["qmv_y"][0] = -(gx[0] + 60) * 4 - 2, 8          # far into the left margin
        pairs["qmv_x"][1:5] = (4, 6, 8, 5)
        pairs["qmv_y"][1:5] = (8, 4, 7, 4)                                       # integer, H, V and HV phases for sure
        eng.bipred_source_batch(dst, w, h, pairs)
        for p in pairs:
            qx, qy, px, py = int(p["qmv_x"]), int(p["qmv_y"]), int(p["pu_x"]), int(p["pu_y"])
            ref, ss = fr[2], fr[2].shape[1]
            o = (M + py + (qy >> 2)) * ss + M + px + (qx >> 2)
            pred = np.zeros((h, w), dtype=dt)
            if not (qx & 3) and not (qy & 3):
                pred[:] = ref[M + py + (qy >> 2):M + py + (qy >> 2) + h, M + px + (qx >> 2):M + px + (qx >> 2) + w]
            elif not (qy & 3):
                oracle.ipfilter(0, ref, o, ss, pred, 0, w, w, h, qx & 3, depth, bits // 8)
            elif not (qx & 3):
                oracle.ipfilter(1, ref, o, ss, pred, 0, w, w, h, qy & 3, depth, bits // 8)
            else:
                tmp = np.zeros((h + 7, w), dtype=np.int16)
                oracle.ipfilter(2, ref, o - 3 * ss, ss, tmp, 0, w, w, h + 7, qx & 3, depth, bits // 8)
                oracle.ipfilter(3, tmp, 3 * w, w, pred, 0, w, w, h, qy & 3, depth, bits // 8)
            org = fr[0][M + py:M + py + h, M + px:M + px + w].astype(np.int32)
            host_dst[M + py:M + py + h, M + px:M + px + w] = np.clip(2 * org - pred.astype(np.int32), 0, (1 << depth) - 1).astype(dt)
        all_jobs.append(J.make_jobs(W, H, gx, gy, w, h, method=J.BIPRED, subme=0, merange=4, qp=30, qmvp=(3, -5)))
    got_plane = eng.download_plane(dst, fr[0].shape, dt)
    assert np.array_equal(got_plane[M:M + H, M:M + W], host_dst[M:M + H, M:M + W])
    jobs = np.concatenate(all_jobs)
    host_dst_padded = synth.pad_plane(host_dst[M:M + H, M:M + W].astype(np.int32), depth)
    # (the device plane's margins were never written: keep the searches' PUs inside the picture, as they are)
    got = eng.search(dst, [ids[1]], jobs)
    exp, _ = oracle.search(host_dst_padded, [fr[1]], jobs, threads=4, bit_depth=depth)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:5]
    for i in ids + [dst]:
        eng.release_plane(i)


def api_mc_dtype():
    from x265_amd.api import MC_PAIR_DTYPE
    return MC_PAIR_DTYPE


@pytest.mark.parametrize("bits", [8, 16])
def test_rowwise_published_plane_equals_whole_upload(eng, oracle, bits):
    """a reference plane filled CTU row by CTU row from an unpadded picture, borders extended on the
    device, equals the plane uploaded whole with host-made borders - bytes and search results"""
    depth = 8 if bits == 8 else 10
    W, H = 208, 152                                   # the last CTU row is partial
    fr = synth.make_sequence(W, H, 2, seed=66, bit_depth=depth)
    M = synth.MARGIN
    pic = np.ascontiguousarray(fr[1][M:M + H, M:M + W])
    whole = eng.upload_plane(fr[1], bit_depth=depth)
    rows = eng.create_plane(W, H, bytes_per_px=bits // 8, bit_depth=depth)
    jobs = J.frame_jobs(W, H, sizes=(16,), method=J.HEX, subme=2, merange=16, qp=30)
    enc = eng.upload_plane(fr[0], bit_depth=depth)
    for y0 in range(0, H, 64):
        n = min(64, H - y0)
        eng.update_plane_rows(rows, pic, y0, n)
        if y0 + n < H:
            # rows published so far: searches whose windows stay above the published edge already work
            jb = jobs[jobs["pu_y"] + 16 + 16 + 7 <= y0 + n].copy()
            jb["mvmax_y"] = np.minimum(jb["mvmax_y"], y0 + n - 7 - 3 - 16 - jb["pu_y"])
            jb = jb[jb["mvmax_y"] >= jb["mvmin_y"]]
            if len(jb):
                assert np.array_equal(eng.search(enc, [rows], jb), eng.search(enc, [whole], jb)), y0
    a = eng.download_plane(whole, fr[1].shape, fr[1].dtype)
    b = eng.download_plane(rows, fr[1].shape, fr[1].dtype)
    assert np.array_equal(a, fr[1]) and np.array_equal(b, fr[1])
    got = eng.search(enc, [rows], jobs)
    exp, _ = oracle.search(fr[0], [fr[1]], jobs, bit_depth=depth)
    assert np.array_equal(got, exp)
    for p in (whole, rows, enc):
        eng.release_plane(p)


def test_joblist_rebalance_keeps_results(eng, oracle):
    """reordering a planned list by the last run's job lengths changes the launch order only"""
    fr = synth.make_sequence(416, 240, 3, seed=21)
    ids = [eng.upload_plane(f) for f in fr]
    jobs = np.concatenate([J.frame_jobs(416, 240, sizes=(64, 32, 16, 8), method=m, subme=3, merange=32, qp=28)
                           for m in (J.HEX, J.STAR)])
    jl = eng.make_joblist(jobs)
    jl.rebalance()                                    # before any run: a no-op
    a = jl.run(ids[0], [ids[1]])
    jl.rebalance()
    b = jl.run(ids[0], [ids[1]])
    c = jl.run(ids[2], [ids[1]])                      # another picture with the order learnt from the first
    jl.rebalance()
    d = jl.run(ids[2], [ids[1]])
    jl.close()
    exp, _ = oracle.search(fr[0], [fr[1]], jobs, threads=4)
    exp2, _ = oracle.search(fr[2], [fr[1]], jobs, threads=4)
    assert np.array_equal(a, exp) and np.array_equal(b, exp)
    assert np.array_equal(c, exp2) and np.array_equal(d, exp2)
    for i in ids:
        eng.release_plane(i)
