"""GPU parity of the lookahead (half-resolution) path against the CPU restatement, which is itself
pinned to the compiled reference by tests/golden/lowres_u*.npz: plane construction, searches with a
half-resolution reference (subpelCompare's isLowres branch) and the whole-frame neighbour-predicted
search.  Everything goes through the C ABI (x265_amd/api.py -> libx265amd.so)."""
import os

import numpy as np
import pytest

import golden_cases as G
from x265_amd import jobs as J
from x265_amd import synth
from x265_amd.lowres import lowres_geometry

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from x265_amd.api import Engine
    e = Engine(0)
    yield e
    e.close()


def _gold(bits):
    return np.load(os.path.join(G.GOLDEN_DIR, "lowres_u%d.npz" % bits), allow_pickle=False)


@pytest.mark.parametrize("bits", [8, 16])
def test_lookahead_path_matches_reference_vectors(eng, bits):
    """the golden suite of the CPU tests (outputs of the compiled reference), replayed on the GPU"""
    gold = _gold(bits)
    depth = 8 if bits == 8 else 10
    fr = G.lowres_source_frames(bits)
    src = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    lr = [eng.lowres_create(s) for s in src]
    g = lowres_geometry(G.LOWRES_W, G.LOWRES_H, synth.MARGIN, synth.MARGIN)
    assert eng.lowres_geometry(lr[0]) == (g.width, g.lines, g.stride, g.cu_w, g.cu_h)
    for name, pid in (("crc_enc", lr[0]), ("crc_ref", lr[1])):
        planes = [eng.download_plane(pid, g.padded_shape, fr[0].dtype, which=k) for k in range(4)]
        for q in planes:
            q[:, g.width + 2 * synth.MARGIN:] = 0     # columns that only pad the stride to 32 (zero in the CPU arrays)
        assert np.array_equal(G.plane_crc(planes), gold[name]), name
    jobs = G.lowres_jobs(g, bits)
    got = eng.search(lr[0], [lr[1]], jobs)
    bad = np.nonzero(got != gold["jobs"])[0]
    assert bad.size == 0, (jobs[bad[:3]], got[bad[:3]], gold["jobs"][bad[:3]])
    for k, (method, subme, merange, qp) in enumerate(G.LOOKAHEAD_CONFIGS):
        got = eng.lookahead_me([lr[0]], [lr[1]], merange=merange, method=method, subme=subme, qp=qp)[0]
        exp = gold["field%d" % k]
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (k, bad[:5], got[bad[:3]], exp[bad[:3]])
    for p in lr + src:
        eng.release_plane(p)


@pytest.mark.parametrize("bits,w,h", [(8, 1920, 1080), (16, 832, 480), (8, 176, 144), (16, 3840, 2160)])
def test_lookahead_frames_match_oracle(eng, oracle, bits, w, h):
    """several frame pairs in one call (B-frame style: both directions), reference defaults
    (HEX, subme 1, merange 16), planes re-initialised in place for a second round"""
    depth = 8 if bits == 8 else 10
    fr = synth.make_sequence(w, h, 4, seed=31 + bits, bit_depth=depth)
    src = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    lr = [eng.lowres_create(s) for s in src[:3]]
    cpu = [oracle.lowres_init(f) for f in fr]
    g = cpu[0][0]
    for k in range(3):
        for which in range(4):
            got = eng.download_plane(lr[k], g.padded_shape, fr[0].dtype, which=which)
            assert np.array_equal(got[:, :g.width + 2 * synth.MARGIN], cpu[k][1][which][:, :g.width + 2 * synth.MARGIN]), (k, which)
    pairs = [(1, 0), (1, 2), (2, 1), (0, 2)]
    got = eng.lookahead_me([lr[a] for a, _ in pairs], [lr[b] for _, b in pairs], qp=12 if bits == 8 else 16)
    evals0 = eng.eval_count(reset=True)
    total = 0
    for i, (a, b) in enumerate(pairs):
        exp, ev = oracle.lookahead_me(cpu[a][1], cpu[b][1], g.cu_w, g.cu_h, 16, J.HEX, 1, 12 if bits == 8 else 16, bit_depth=depth)
        total += ev
        bad = np.nonzero(got[i] != exp)[0]
        assert bad.size == 0, (i, bad[:5], got[i][bad[:3]], exp[bad[:3]])
    assert evals0 >= total          # (the counter may hold evaluations of earlier tests)
    # new picture into an existing set
    eng.lowres_update(lr[0], src[3])
    got = eng.lookahead_me([lr[0]], [lr[1]])[0]
    exp, _ = oracle.lookahead_me(cpu[3][1], cpu[1][1], g.cu_w, g.cu_h, 16, J.HEX, 1, 12, bit_depth=depth)
    assert np.array_equal(got, exp)
    for p in lr + src:
        eng.release_plane(p)


def test_mixed_reference_kinds_in_one_batch(eng, oracle):
    """jobs against a half-resolution set and against an ordinary plane of the same geometry in one
    batch: the kernels built for half-resolution sets must also serve full-resolution references"""
    fr = G.lowres_source_frames(8)
    cpu = [oracle.lowres_init(f) for f in fr]
    g = cpu[0][0]
    src = [eng.upload_plane(f) for f in fr]
    lr = [eng.lowres_create(s) for s in src]
    # the F plane of frame 1 uploaded again as an ordinary plane (same stride as the set)
    plain = eng.upload_plane(np.ascontiguousarray(cpu[1][1][0]))
    jobs = G.lowres_jobs(g, 8)
    jobs["ref_idx"] = np.arange(len(jobs)) % 2
    got = eng.search(lr[0], [lr[1], plain], jobs)
    exp = np.zeros(len(jobs), dtype=got.dtype)
    a = jobs["ref_idx"] == 0
    ja, jb = jobs[a].copy(), jobs[~a].copy()
    ja["ref_idx"] = 0
    jb["ref_idx"] = 0
    exp[a] = oracle.search(cpu[0][1][0], [cpu[1][1]], ja, lowres=True)[0]
    exp[~a] = oracle.search(cpu[0][1][0], [cpu[1][1][0]], jb)[0]
    bad = np.nonzero(got != exp)[0]
    assert bad.size == 0, (jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
    for p in lr + src + [plain]:
        eng.release_plane(p)


@pytest.mark.parametrize("w,h", [(16, 16), (32, 16), (16, 48), (48, 48), (34, 22), (130, 18)])
def test_lookahead_tiny_frames(eng, oracle, w, h):
    """one CU, one row, one column, odd sizes that round up to whole CUs: every neighbour pattern of the
    predictor chain, all methods"""
    fr = synth.make_sequence(w, h, 2, seed=w * 100 + h)
    src = [eng.upload_plane(f) for f in fr]
    lr = [eng.lowres_create(s) for s in src]
    cpu = [oracle.lowres_init(f) for f in fr]
    g = cpu[0][0]
    assert eng.lowres_geometry(lr[0])[3:] == (g.cu_w, g.cu_h)
    for k in range(2):
        for which in range(4):
            got = eng.download_plane(lr[k], g.padded_shape, np.uint8, which=which)
            assert np.array_equal(got[:, :g.width + 2 * synth.MARGIN], cpu[k][1][which][:, :g.width + 2 * synth.MARGIN])
    for method in range(5):
        for subme in (1, 7):
            got = eng.lookahead_me([lr[0], lr[1]], [lr[1], lr[0]], merange=8, method=method, subme=subme, qp=20)
            for i, (a, b) in enumerate(((0, 1), (1, 0))):
                exp, _ = oracle.lookahead_me(cpu[a][1], cpu[b][1], g.cu_w, g.cu_h, 8, method, subme, 20)
                assert np.array_equal(got[i], exp), (method, subme, i)
    for p in lr + src:
        eng.release_plane(p)


def test_lowres_argument_checks(eng):
    from x265_amd.api import X265AmdError
    fr = synth.make_sequence(64, 64, 1, seed=3)
    src = eng.upload_plane(fr[0])
    lr = eng.lowres_create(src)
    with pytest.raises(X265AmdError):
        eng.lowres_create(lr)                   # a half-resolution set is not a source
    with pytest.raises(X265AmdError):
        eng.lookahead_me([src], [lr])           # needs half-resolution sets
    small = eng.upload_plane(synth.pad_plane(np.zeros((64, 64), np.uint8), margin=8), margin=8)
    with pytest.raises(X265AmdError):
        eng.lowres_create(small)                # margins below 16
    for p in (lr, src, small):
        eng.release_plane(p)


def test_tall_narrow_picture_with_long_vertical_vectors(eng, oracle):
    """a 32x280 picture of flat areas with steps (found by tests/fuzz_chain.py): its MVs reach 140 half-resolution
    pixels, so UMH's vertical cross singles - measured at x = omv.x + omv.y (encoder/motion.cpp:176,271-273) - land in
    the columns that pad the plane's stride, which must read as zero like the oracle's zero-initialised planes"""
    d = np.load(os.path.join(G.GOLDEN_DIR, "lookahead_tall_narrow_u8.npz"), allow_pickle=False)
    fr = [d["f0"], d["f1"]]
    src = [eng.upload_plane(f) for f in fr]
    lr = [eng.lowres_create(s) for s in src]
    cpu = [oracle.lowres_init(f) for f in fr]
    g = cpu[0][0]
    for k in range(2):
        for which in range(4):
            got = eng.download_plane(lr[k], g.padded_shape, np.uint8, which=which)
            assert np.array_equal(got, cpu[k][1][which]), (k, which)       # stride padding included
    longest = 0
    for method in range(4):
        for merange in (8, 24):
            got = eng.lookahead_me([lr[0], lr[1]], [lr[1], lr[0]], merange=merange, method=method, subme=1, qp=21)
            for i, (a, b) in enumerate(((0, 1), (1, 0))):
                exp, _ = oracle.lookahead_me(cpu[a][1], cpu[b][1], g.cu_w, g.cu_h, merange, method, 1, 21)
                assert np.array_equal(got[i], exp), (method, merange, i, np.nonzero(got[i] != exp)[0][:5])
                longest = max(longest, int(np.abs(exp["mvy"].astype(np.int32)).max()))
    assert longest > 4 * 64             # the long vectors are really there
    for p in lr + src:
        eng.release_plane(p)
