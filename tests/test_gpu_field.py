"""GPU parity of the whole-frame field with the encoder's predictor chain (x265amd_field_me): every PU's
predictor, range and candidates are derived on the device from the neighbours' final MVs."""
import os

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


class GpuChain:
    def __init__(self, eng, bits):
        self.eng, self.depth = eng, 8 if bits == 8 else 10

    def field_run(self, enc, ref, pw, ph, mr, method, subme, qp, part_mode=0, col=None):
        ids = [self.eng.upload_plane(enc, bit_depth=self.depth), self.eng.upload_plane(ref, bit_depth=self.depth)]
        if part_mode or col is not None:
            out = self.eng.field_me_multi([ids[0]], [ids[1]], [pw], [ph], [part_mode], mr, method, subme, qp, col=col)[0]
        else:
            out = self.eng.field_me(ids[0], ids[1], pw, ph, mr, method, subme, qp)
        for i in ids:
            self.eng.release_plane(i)
        return out


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_match_reference_vectors(eng, bits):
    """the fixtures made by the compiled reference's MotionEstimate under oracle/ref_shim.cpp's driver
    (all five methods, PU sizes 8x8 .. 64x64 and 16x8, motion and noise content), replayed on the GPU"""
    gold = np.load(os.path.join(G.GOLDEN_DIR, "chain_u%d.npz" % bits), allow_pickle=False)
    got = G.run_chain_suite(GpuChain(eng, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_of_cu_parts_match_reference_vectors(eng, bits):
    """the parts of 2NxN / Nx2N CUs (x265amd_field_me_multi's part modes) against the fixtures made by the compiled
    reference under the same driver: 16x8 .. 64x32 and 4x8 parts, all five methods"""
    gold = np.load(os.path.join(G.GOLDEN_DIR, "chain_parts_u%d.npz" % bits), allow_pickle=False)
    got = G.run_chain_parts_suite(GpuChain(eng, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_of_asymmetric_parts_match_reference_vectors(eng, bits):
    """the parts of 2NxnU / 2NxnD / nLx2N / nRx2N CUs (part modes 4..7) against the fixtures made by the compiled reference under
    the same driver: all twelve asymmetric PU shapes, 16x4 .. 48x64, all five methods"""
    gold = np.load(os.path.join(G.GOLDEN_DIR, "chain_amp_u%d.npz" % bits), allow_pickle=False)
    got = G.run_chain_amp_suite(GpuChain(eng, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_with_temporal_candidate_match_reference_vectors(eng, bits):
    """x265amd_field_me_multi_tmvp: a co-located motion field per 16x16 block supplies the temporal AMVP candidate (below-right of
    the PU, else its centre) where fewer than two spatial candidates exist; against the fixtures made by the compiled reference
    under the same driver: 2Nx2N, 2NxN, Nx2N and asymmetric parts"""
    gold = np.load(os.path.join(G.GOLDEN_DIR, "chain_tmvp_u%d.npz" % bits), allow_pickle=False)
    got = G.run_chain_tmvp_suite(GpuChain(eng, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


def test_1080p_fields_with_a_co_located_field_each_in_one_call(eng, oracle):
    """one call, one 1080p frame, four fields (16x16 2Nx2N, 32x32 2NxnD, 8x8, 16x8 2NxN), each with a co-located motion field of
    its own - here the 16x16 field of the previous frame pair scaled to twice / half / the opposite distance - and one without
    motion anywhere (equal to no temporal candidate); every field and the evaluation total equal the oracle's"""
    W, H = 1920, 1080
    fr = synth.make_sequence(W, H, 3, seed=94)
    ids = [eng.upload_plane(f) for f in fr]
    prev, _ = eng.field_me(ids[1], ids[0], 16, 16, 32, J.HEX, 3, 30)         # the "coded" motion of the previous picture
    cw, ch = (W + 15) // 16, (H + 15) // 16
    base = np.zeros((ch, cw), dtype=J.RESULT_DTYPE)
    base["cost"] = -1
    base[:prev.shape[0], :prev.shape[1]] = prev
    base["cost"][::7, ::5] = -1                                               # some "intra" blocks

    def scaled(cur, cur_ref, col, col_ref):
        c = base.copy()
        for e in c.reshape(-1):
            if e["cost"] >= 0:
                e["mvx"], e["mvy"] = oracle.scale_col_mv(cur, cur_ref, col, col_ref, int(e["mvx"]), int(e["mvy"]))
        return c.reshape(-1)

    none = base.copy().reshape(-1)
    none["cost"] = -1
    shapes = [(16, 16, 0), (32, 32, 5), (8, 8, 0), (16, 8, 1), (16, 16, 0)]
    cols = [scaled(2, 1, 1, 0), scaled(3, 1, 1, 0), scaled(2, 1, 1, -1), scaled(2, 3, 1, 0), none]
    n = len(shapes)
    eng.eval_count(reset=True)
    got = eng.field_me_multi([ids[2]] * n, [ids[1]] * n, [s[0] for s in shapes], [s[1] for s in shapes], [s[2] for s in shapes],
                             32, J.HEX, 3, 30, col=np.stack(cols))
    total = 0
    for (pw, ph, pm), col, (res, mvp) in zip(shapes, cols, got):
        exp, exp_mvp, evals = oracle.field_me(fr[2], fr[1], pw, ph, 32, J.HEX, 3, 30, part_mode=pm, col=col)
        assert np.array_equal(res, exp), ((pw, ph, pm), np.argwhere(res != exp)[:5])
        assert np.array_equal(mvp, exp_mvp), (pw, ph, pm)
        total += evals
    assert eng.eval_count() == total
    plain, plain_mvp, _ = oracle.field_me(fr[2], fr[1], 16, 16, 32, J.HEX, 3, 30)
    assert np.array_equal(got[4][1], plain_mvp) and np.array_equal(got[4][0], plain)
    assert not np.array_equal(got[0][1], plain_mvp)                           # the candidate decides some predictors
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("waves", [1, 3, 4, 8])
def test_fields_with_every_number_of_wavefronts_per_row(eng, oracle, waves, monkeypatch):
    """the candidate-parallel classes (PUs of at most 1024 pixels) deal a call's candidates to the wavefronts of the row's workgroup;
    the library picks eight or four by the size of the call - here every count is forced (X265AMD_CP_WAVES, read per call): six
    fields of both classes and all kinds of parts with sub-pel rounds of eight candidates, each equal to the oracle's"""
    monkeypatch.setenv("X265AMD_CP_WAVES", str(waves))
    W, H = 416, 240
    fr = synth.make_sequence(W, H, 2, seed=95)
    ids = [eng.upload_plane(f) for f in fr]
    shapes = [(8, 8, 0), (16, 16, 0), (32, 32, 0), (16, 8, 1), (32, 32, 6), (64, 64, 4)]
    n = len(shapes)
    eng.eval_count(reset=True)
    got = eng.field_me_multi([ids[1]] * n, [ids[0]] * n, [s[0] for s in shapes], [s[1] for s in shapes], [s[2] for s in shapes],
                             24, J.UMH, 6, 29)
    total = 0
    for (pw, ph, pm), (res, mvp) in zip(shapes, got):
        exp, exp_mvp, evals = oracle.field_me(fr[1], fr[0], pw, ph, 24, J.UMH, 6, 29, part_mode=pm)
        assert np.array_equal(res, exp), ((pw, ph, pm), np.argwhere(res != exp)[:5])
        assert np.array_equal(mvp, exp_mvp), (pw, ph, pm)
        total += evals
    assert eng.eval_count() == total
    for i in ids:
        eng.release_plane(i)


def test_1080p_asymmetric_fields_beside_the_square_ones_in_one_call(eng, oracle):
    """one call, one 1080p frame: the four asymmetric fields of the 32x32 CUs, two of the 16x16 and two of the 64x64 CUs beside
    the 2Nx2N field of 32x32; every field and the evaluation total equal the oracle's"""
    W, H = 1920, 1080
    fr = synth.make_sequence(W, H, 2, seed=93)
    ids = [eng.upload_plane(f) for f in fr]
    shapes = [(32, 32, 0), (32, 32, 4), (32, 32, 5), (32, 32, 6), (32, 32, 7), (16, 16, 4), (16, 16, 7), (64, 64, 5), (64, 64, 6)]
    n = len(shapes)
    eng.eval_count(reset=True)
    got = eng.field_me_multi([ids[1]] * n, [ids[0]] * n, [s[0] for s in shapes], [s[1] for s in shapes], [s[2] for s in shapes],
                             32, J.HEX, 3, 30)
    total = 0
    for (pw, ph, pm), (res, mvp) in zip(shapes, got):
        exp, exp_mvp, evals = oracle.field_me(fr[1], fr[0], pw, ph, 32, J.HEX, 3, 30, part_mode=pm)
        assert res.shape == exp.shape, (pw, ph, pm)
        assert np.array_equal(res, exp), ((pw, ph, pm), np.argwhere(res != exp)[:5])
        assert np.array_equal(mvp, exp_mvp), (pw, ph, pm)
        total += evals
    assert eng.eval_count() == total
    for i in ids:
        eng.release_plane(i)


def test_1080p_fields_of_every_depth_and_part_mode_in_one_call(eng, oracle):
    """One call, one 1080p frame, a field per quadtree depth (64x64 .. 8x8) plus the 2NxN and Nx2N fields of the 16x16 and
    32x32 CUs: every field equals the oracle's, the evaluation counts add up, and the call takes about as long as its
    longest field (the 8x8 one) rather than the sum."""
    import time
    W, H = 1920, 1080
    fr = synth.make_sequence(W, H, 2, seed=92)
    ids = [eng.upload_plane(f) for f in fr]
    shapes = [(8, 8, 0), (16, 16, 0), (32, 32, 0), (64, 64, 0), (16, 8, 1), (8, 16, 2), (32, 16, 1), (16, 32, 2)]
    n = len(shapes)
    args = ([ids[1]] * n, [ids[0]] * n, [s[0] for s in shapes], [s[1] for s in shapes], [s[2] for s in shapes], 32, J.HEX, 3, 30)
    eng.field_me_multi(*args)                       # (warm-up)
    eng.eval_count(reset=True)
    t0 = time.perf_counter()
    got = eng.field_me_multi(*args)
    t_all = time.perf_counter() - t0
    total = 0
    for (pw, ph, pm), (res, mvp) in zip(shapes, got):
        exp, exp_mvp, evals = oracle.field_me(fr[1], fr[0], pw, ph, 32, J.HEX, 3, 30, part_mode=pm)
        assert np.array_equal(res, exp), ((pw, ph, pm), np.argwhere(res != exp)[:5])
        assert np.array_equal(mvp, exp_mvp), (pw, ph, pm)
        total += evals
    assert eng.eval_count() == total
    t0 = time.perf_counter()
    eng.field_me(ids[1], ids[0], 8, 8, 32, J.HEX, 3, 30)
    t_8 = time.perf_counter() - t0
    print("eight fields %.2f ms, the 8x8 field alone %.2f ms" % (1e3 * t_all, 1e3 * t_8))
    assert t_all < 2.5 * t_8
    for i in ids:
        eng.release_plane(i)


@pytest.mark.parametrize("pu,method,subme", [(16, J.HEX, 5), (64, J.STAR, 3), (8, J.HEX, 1)])
def test_1080p_field_matches_oracle(eng, oracle, pu, method, subme):
    """a whole 1080p frame: results, predictors and the evaluation count equal the CPU restatement's"""
    W, H = 1920, 1080
    fr = synth.make_sequence(W, H, 2, seed=91)
    ids = [eng.upload_plane(f) for f in fr]
    exp, exp_mvp, evals = oracle.field_me(fr[1], fr[0], pu, pu, 64, method, subme, 32)
    eng.eval_count(reset=True)
    got, got_mvp = eng.field_me(ids[1], ids[0], pu, pu, 64, method, subme, 32)
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:5]
    assert np.array_equal(got_mvp, exp_mvp)
    assert eng.eval_count() == evals
    assert (exp_mvp != 0).any(-1).mean() > 0.5         # the chain really carries predictors
    for i in ids:
        eng.release_plane(i)


def test_fields_of_several_references_in_one_call(eng, oracle):
    """x265amd_field_me_batch: a frame against three references (and one pair the other way round) in one launch;
    every field equals the single-field result of the oracle, and the call takes about as long as one field."""
    W, H = 640, 368
    fr = synth.make_sequence(W, H, 4, seed=23)
    ids = [eng.upload_plane(f) for f in fr]
    enc = [ids[3], ids[3], ids[3], ids[0]]
    ref = [ids[2], ids[1], ids[0], ids[3]]
    enc_np = [fr[3], fr[3], fr[3], fr[0]]
    ref_np = [fr[2], fr[1], fr[0], fr[3]]
    eng.eval_count(reset=True)
    got, got_mvp = eng.field_me_batch(enc, ref, 16, 16, 32, J.HEX, 3, 30)
    total = 0
    for f in range(4):
        exp, exp_mvp, evals = oracle.field_me(enc_np[f], ref_np[f], 16, 16, 32, J.HEX, 3, 30)
        assert np.array_equal(got[f], exp), (f, np.argwhere(got[f] != exp)[:5])
        assert np.array_equal(got_mvp[f], exp_mvp), f
        total += evals
    assert eng.eval_count() == total
    with pytest.raises(Exception):
        small = eng.upload_plane(synth.make_sequence(320, 192, 1, seed=1)[0])
        try:
            eng.field_me_batch([ids[3], small], [ids[2], small], 16, 16, 32, J.HEX, 3, 30)     # mixed picture sizes
        finally:
            eng.release_plane(small)
    for i in ids:
        eng.release_plane(i)


def test_field_rejects_bad_arguments(eng):
    from x265_amd.api import X265AmdError
    fr = synth.make_sequence(128, 64, 2, seed=3)
    ids = [eng.upload_plane(f) for f in fr]
    with pytest.raises(X265AmdError):
        eng.field_me(ids[0], ids[1], 20, 16, 16, J.HEX, 1, 30)
    # part modes: NxN (3) and numbers past nRx2N; asymmetric parts of an 8x8 or a non-square CU
    for pw, ph, pm in ((16, 16, 3), (16, 16, 8), (8, 8, 4), (32, 16, 6), (64, 64, 1)):
        with pytest.raises(X265AmdError):
            eng.field_me_multi([ids[0]], [ids[1]], [pw], [ph], [pm], 16, J.HEX, 1, 30)
    small = synth.make_sequence(128, 64, 1, seed=3)[0][16:-16, 16:-16].copy()       # margins of 64 < 78
    sid = eng.upload_plane(small, margin=64)
    with pytest.raises(X265AmdError, match="-5"):
        eng.field_me(sid, sid, 16, 16, 16, J.HEX, 1, 30)
    for i in ids + [sid]:
        eng.release_plane(i)


@pytest.mark.parametrize("bits", [8, 16])
def test_mc_cost_batch_matches_oracle(eng, oracle, bits):
    """costs against motion-compensated blocks (xGetTemplateCost's and the merge check's distortion): every
    phase, SAD / SATD / SA8D, PU sizes of all three kernel classes, MVs reaching into the margins"""
    from x265_amd.api import MC_PAIR_DTYPE, X265AmdError
    depth = 8 if bits == 8 else 10
    W, H = 256, 160
    fr = synth.make_sequence(W, H, 2, seed=66, bit_depth=depth)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    rng = np.random.default_rng(bits)
    for (w, h) in ((8, 8), (16, 16), (32, 16), (64, 64), (12, 16), (4, 8)):
        n = 40
        p = np.zeros(n, dtype=MC_PAIR_DTYPE)
        p["enc_plane"], p["ref_plane"] = ids[0], ids[1]
        p["pu_x"] = rng.integers(0, (W - w) // 4 + 1, n) * 4
        p["pu_y"] = rng.integers(0, (H - h) // 4 + 1, n) * 4
        p["qmv_x"] = rng.integers(-280, 280, n)
        p["qmv_y"] = rng.integers(-280, 280, n)
        p["qmv_x"][:4] = (0, 4, 2, 0)
        p["qmv_y"][:4] = (0, 0, 0, 3)
        for kind in (0, 1, 2):
            got = eng.mc_cost_batch(kind, w, h, p)
            exp = [oracle.mc_cost(fr[0], fr[1], int(q["pu_x"]), int(q["pu_y"]), w, h, int(q["qmv_x"]), int(q["qmv_y"]), kind, bit_depth=depth)
                   for q in p]
            assert got.tolist() == exp, (w, h, kind)
    bad = np.zeros(1, dtype=MC_PAIR_DTYPE)
    bad["enc_plane"], bad["ref_plane"], bad["qmv_x"] = ids[0], ids[1], -4 * 200
    with pytest.raises(X265AmdError, match="-5"):
        eng.mc_cost_batch(0, 16, 16, bad)
    for i in ids:
        eng.release_plane(i)

