"""CPU restatement (oracle/libme_oracle.so) against the golden vectors that were generated from
the compiled reference (oracle/gen_golden.py).  This is the pin that lets the restatement act
as the checker on the GPU box, where the reference itself is absent."""
import os
import zlib

import numpy as np
import pytest

import golden_cases as G


def _load(name):
    return np.load(os.path.join(G.GOLDEN_DIR, name), allow_pickle=False)


class OracleIp:
    def __init__(self, oracle, bits):
        self.o, self.bits = oracle, bits

    def ipfilter_run(self, kind, src, x0, y0, w, h, frac, depth):
        pdt = np.uint8 if self.bits == 8 else np.uint16
        dst = np.zeros((h, w), dtype=np.int16 if kind == 2 else pdt)
        ss = src.shape[1]
        self.o.ipfilter(kind, src, y0 * ss + x0, ss, dst, 0, w, w, h, frac, depth, self.bits // 8)
        return dst


@pytest.mark.parametrize("bits", [8, 16])
def test_primitives_match_reference_vectors(oracle, bits):
    gold = _load("prim_u%d.npz" % bits)
    got = G.run_prim_suite(oracle, bits)
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), k


@pytest.mark.parametrize("bits,depth", [(8, 8), (16, 8), (16, 10)])
def test_interpolation_matches_reference_vectors(oracle, bits, depth):
    gold = _load("ipfilter_u%d_d%d.npz" % (bits, depth))
    got = G.run_ip_suite(OracleIp(oracle, bits), bits, depth)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), k


def test_mvcost_tables_match_reference_vectors(oracle):
    gold = _load("mvcost.npz")
    for qp in G.MV_QPS:
        t = oracle.mvcost_table(qp)
        # entry 0 (MVD -32768) is never written by the reference (bitcost.cpp:52-56)
        assert np.array_equal(t[1:], gold["qp%d" % qp][1:]), qp
    for p, m, (c, b) in zip(gold["bc_mvp"].tolist(), gold["bc_mv"].tolist(), gold["bc_out"].tolist()):
        assert oracle.lib.meo_mvcost(32, p[0], p[1], m[0], m[1]) == c
        assert oracle.lib.meo_bitcost(p[0], p[1], m[0], m[1]) == b


@pytest.mark.parametrize("bits", [8, 16])
def test_search_matches_reference_vectors(oracle, bits):
    gold = _load("search_u%d.npz" % bits)
    jobs = G.search_jobs(bits)
    assert zlib.crc32(jobs.tobytes()) == int(gold["jobs_crc"][0])
    for kind in ("motion", "noise"):
        fr = G.search_frames(bits, kind)
        got, evals = oracle.search(fr[0], [fr[1], fr[2]], jobs, threads=4, bit_depth=8 if bits == 8 else 10)
        exp = gold["res_" + kind]
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (kind, jobs[bad[:3]], got[bad[:3]], exp[bad[:3]])
        assert evals > 0


def test_full_search_field_matches_reference_vectors(oracle):
    gold = _load("field_full.npz")
    jobs = G.field_jobs()
    assert zlib.crc32(jobs.tobytes()) == int(gold["jobs_crc"][0])
    fr = G.field_frames()
    got, _ = oracle.search(fr[0], [fr[1]], jobs, threads=4)
    assert np.array_equal(got, gold["res"])


@pytest.mark.parametrize("bits", [8, 16])
def test_lookahead_path_matches_reference_vectors(oracle, bits):
    """half-resolution planes (CRC), isLowres searches and the lookahead's neighbour-predicted MV fields"""
    gold = _load("lowres_u%d.npz" % bits)
    got = G.run_lowres_suite(oracle, bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.nonzero(gold[k] != got[k])[0][:5])


@pytest.mark.parametrize("bits", [8, 16])
def test_bidirectional_refinement_matches_reference_vectors(oracle, bits):
    gold = _load("bipred_u%d.npz" % bits)
    jobs = G.bipred_jobs(bits)
    assert zlib.crc32(jobs.tobytes()) == int(gold["jobs_crc"][0])
    mod, rf = G.bipred_frames(bits)
    got, _ = oracle.search(mod, [rf], jobs, bit_depth=8 if bits == 8 else 10)
    bad = np.nonzero(got != gold["res"])[0]
    assert bad.size == 0, (jobs[bad[:3]], got[bad[:3]], gold["res"][bad[:3]])


class OracleChain:
    def __init__(self, oracle, bits):
        self.o, self.depth = oracle, 8 if bits == 8 else 10

    def field_run(self, enc, ref, pw, ph, mr, method, subme, qp, part_mode=0, col=None):
        return self.o.field_me(enc, ref, pw, ph, mr, method, subme, qp, bit_depth=self.depth, part_mode=part_mode, col=col)[:2]


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_match_reference_vectors(oracle, bits):
    """whole-frame fields searched in raster order with predictors derived from the neighbours' results
    (AMVP choice by template SAD, xSetSearchRange, three neighbour candidates), as produced by the compiled
    reference's MotionEstimate and primitives under oracle/ref_shim.cpp's driver"""
    gold = _load("chain_u%d.npz" % bits)
    got = G.run_chain_suite(OracleChain(oracle, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_of_cu_parts_match_reference_vectors(oracle, bits):
    """the same for the two parts of 2NxN and Nx2N CUs (CUs in raster order, their parts one after the other; bounds from
    the CU's position; below-left candidate for the upper part, no above-right for the lower part of a 2NxN CU)"""
    gold = _load("chain_parts_u%d.npz" % bits)
    got = G.run_chain_parts_suite(OracleChain(oracle, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_of_asymmetric_parts_match_reference_vectors(oracle, bits):
    """the same for the two parts of 2NxnU / 2NxnD / nLx2N / nRx2N CUs of 16, 32 and 64 pixels: all twelve asymmetric PU
    shapes, PU sizes and positions from the quarter / three-quarter cut, neighbours as for 2NxN / Nx2N"""
    gold = _load("chain_amp_u%d.npz" % bits)
    got = G.run_chain_amp_suite(OracleChain(oracle, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])


@pytest.mark.parametrize("bits", [8, 16])
def test_predictor_chain_fields_with_temporal_candidate_match_reference_vectors(oracle, bits):
    """the temporal AMVP candidate: a co-located motion field per 16x16 block, looked up below-right of the PU (not across
    the picture edge or into the next CTU row) or else at its centre, joins the list while fewer than two spatial candidates
    exist; 2Nx2N, 2NxN, Nx2N and asymmetric parts"""
    gold = _load("chain_tmvp_u%d.npz" % bits)
    got = G.run_chain_tmvp_suite(OracleChain(oracle, bits), bits)
    assert set(gold.files) == set(got.keys())
    for k in gold.files:
        assert np.array_equal(gold[k], got[k]), (k, np.argwhere(gold[k] != got[k])[:3])
    # the candidate matters: the fields differ from those searched without it
    plain = G.run_chain_tmvp_suite(type("NoCol", (), {"field_run": lambda self, *a, col=None, **kw: OracleChain(oracle, bits).field_run(*a, **kw)})(), bits)
    assert sum(not np.array_equal(plain[k], got[k]) for k in got if k.startswith("mvp")) >= 8


def test_col_mv_scaling_follows_the_poc_distances(oracle):
    """xGetDistScaleFactor / scaleMv: equal distances leave the MV alone; twice the distance doubles it; half the distance halves
    it, a half rounding towards zero ((scale * mv + 127 + (scale * mv < 0)) >> 8); the opposite direction mirrors it; the factor saturates at +-16 (4095 / 256) and the result at
    the int16 range"""
    assert oracle.scale_col_mv(8, 4, 6, 2, 13, -7) == (13, -7)
    assert oracle.scale_col_mv(8, 4, 6, 4, 13, -7) == (26, -14)
    assert oracle.scale_col_mv(8, 6, 6, 2, 13, -7) == (6, -3)
    assert oracle.scale_col_mv(8, 10, 6, 4, 13, -7) == (-13, 7)
    assert oracle.scale_col_mv(100, 0, 6, 5, 100, -3000) == (1600, -32768)

