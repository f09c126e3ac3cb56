"""Randomised differential test: CPU restatement vs the compiled reference (oracle/_ref).
Skipped where the reference build is absent."""
import numpy as np
import pytest

from oracle_ffi import Reference, have_reference
from x265_amd import jobs as J
from x265_amd import synth

pytestmark = pytest.mark.skipif(not have_reference(8), reason="oracle/_ref not built (no /root/reference)")


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 10)])
def test_random_jobs(oracle, bits, depth):
    ref = Reference(bits)
    ref.set_bit_depth(depth)
    W, H = 352, 288
    rng = np.random.default_rng(123 + bits)
    for kind in ("motion", "noise"):
        fr = synth.make_sequence(W, H, 2, seed=31 + bits, bit_depth=depth if bits == 16 else 8, kind=kind)
        parts = []
        for _ in range(150):
            w, h = int(rng.choice(J.DIMS)), int(rng.choice(J.DIMS))
            method = int(rng.integers(0, 5))
            if method == J.UMH and w == 4 and h == 4:
                continue
            merange = int(rng.integers(4, 20)) if method == J.FULL else int(rng.integers(4, 70))
            px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
            py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
            k = int(rng.integers(0, 5))
            mvc = rng.integers(-300, 300, (k, 2)) if k else None
            parts.append(J.make_jobs(W, H, [px], [py], w, h, method=method, subme=int(rng.integers(0, 8)),
                                     merange=merange, qp=int(rng.integers(0, 52)),
                                     qmvp=(int(rng.integers(-250, 250)), int(rng.integers(-250, 250))), mvc=mvc))
        jobs = np.concatenate(parts)
        a, _ = oracle.search(fr[0], [fr[1]], jobs, threads=2, bit_depth=depth)
        b = ref.search(fr[0], [fr[1]], jobs)
        bad = np.nonzero(a != b)[0]
        assert bad.size == 0, (jobs[bad[:2]], a[bad[:2]], b[bad[:2]])


@pytest.mark.skipif(not have_reference("8v"), reason="vector build of oracle/_ref absent")
def test_vector_primitive_build_equals_c_build(oracle):
    """the reference's intrinsic primitives (its own differential-test property, test/testbench.cpp:107-134):
    searches through the SSE4.1-dispatched build return what the C build and the restatement return"""
    ref = Reference("8v")
    W, H = 352, 288
    rng = np.random.default_rng(77)
    fr = synth.make_sequence(W, H, 2, seed=9)
    parts = []
    for _ in range(120):
        w, h = int(rng.choice(J.DIMS)), int(rng.choice(J.DIMS))
        method = int(rng.integers(0, 5))
        if method == J.UMH and w == 4 and h == 4:
            continue
        px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        parts.append(J.make_jobs(W, H, [px], [py], w, h, method=method, subme=int(rng.integers(0, 8)),
                                 merange=int(rng.integers(4, 20)) if method == J.FULL else int(rng.integers(4, 70)),
                                 qp=int(rng.integers(0, 52)), qmvp=(int(rng.integers(-100, 100)), int(rng.integers(-100, 100)))))
    jobs = np.concatenate(parts)
    a, _ = oracle.search(fr[0], [fr[1]], jobs, threads=2)
    b = ref.search(fr[0], [fr[1]], jobs)
    bad = np.nonzero(a != b)[0]
    assert bad.size == 0, (jobs[bad[:2]], a[bad[:2]], b[bad[:2]])


def test_tall_narrow_lookahead_case(oracle):
    """the picture of tests/golden/lookahead_tall_narrow_u8.npz (long vertical vectors; UMH's cross singles read the
    stride padding of the half-resolution planes): restatement and compiled reference agree on zero-initialised planes"""
    import os
    import golden_cases as G
    d = np.load(os.path.join(G.GOLDEN_DIR, "lookahead_tall_narrow_u8.npz"), allow_pickle=False)
    fr = [d["f0"], d["f1"]]
    ref = Reference(8)
    cpu = [oracle.lowres_init(f) for f in fr]
    rcp = [ref.lowres_init(f) for f in fr]
    g = cpu[0][0]
    for method in range(4):
        exp, _ = oracle.lookahead_me(cpu[0][1], cpu[1][1], g.cu_w, g.cu_h, 24, method, 1, 21)
        got = ref.lookahead_me(rcp[0][1], rcp[1][1], g.cu_w, g.cu_h, 24, method, 1, 21)
        got = got[0] if isinstance(got, tuple) else got
        assert np.array_equal(exp, got), method
