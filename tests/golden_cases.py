"""Deterministic inputs of the golden-vector suites.

The expected outputs under tests/golden/ were produced by running the *compiled reference*
(oracle/_ref) on exactly these inputs (oracle/gen_golden.py).  The same inputs are replayed
against the CPU restatement (not-gpu tests) and against the HIP path (gpu tests).
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J  # noqa: E402
from x265_amd import synth  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

PRIM_ITERS = 25
PRIM_INCR = 32
PRIM_STRIDE = 64
PRIM_BUFSIZE = PRIM_STRIDE * (64 + 16) + PRIM_INCR * 100   # buffer recipe of source/test/pixelharness.cpp:49-80


def prim_buffers(bits: int):
    """Two random pixel buffers (PixelHarness recipe, values 0..255 or 0..1023) + adversarial pairs."""
    rng = np.random.default_rng(1000 + bits)
    dt = np.uint8 if bits == 8 else np.uint16
    mx = 255 if bits == 8 else 1023
    a = rng.integers(0, mx + 1, PRIM_BUFSIZE).astype(dt)
    b = rng.integers(0, mx + 1, PRIM_BUFSIZE).astype(dt)
    z = np.zeros(PRIM_BUFSIZE, dt)
    m = np.full(PRIM_BUFSIZE, mx, dt)
    cb = np.where((np.arange(PRIM_BUFSIZE) % 2) == 0, mx, 0).astype(dt)
    adversarial = [(z, m), (m, z), (cb, z), (cb, m), (cb, np.roll(cb, 1))]
    return a, b, adversarial


IP_SIZES = ((8, 8), (16, 16), (64, 64), (65, 64), (64, 65), (12, 16), (4, 4))


def ip_source(bits: int):
    rng = np.random.default_rng(2000 + bits)
    dt = np.uint8 if bits == 8 else np.uint16
    mx = 255 if bits == 8 else 1023
    h, w = 96, 112
    src = rng.integers(0, mx + 1, (h, w)).astype(dt)
    # a band of extreme values to exercise the clips and the short casts
    src[40:48, :] = mx
    src[48:56, ::2] = 0
    ssrc = rng.integers(-12000, 12000, (h, w)).astype(np.int16)
    return src, ssrc


MV_QPS = (0, 12, 22, 27, 32, 37, 51)

SEARCH_W, SEARCH_H = 416, 240
SEARCH_SHAPES = ((64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (32, 64), (16, 64), (64, 16), (8, 4),
                 (4, 8), (12, 16), (24, 32), (48, 64), (4, 4), (32, 8), (16, 12))
SEARCH_SUBME = (0, 1, 2, 5, 7)


def search_frames(bits: int, kind: str):
    return synth.make_sequence(SEARCH_W, SEARCH_H, 3, seed=77 + bits, bit_depth=8 if bits == 8 else 10, kind=kind)


def search_jobs(bits: int) -> np.ndarray:
    """Job grid: 5 methods x 5 subme levels x 16 PU shapes x 3 predictor modes x ranges x QPs.
    ref_idx alternates between two reference frames."""
    rng = np.random.default_rng(3000 + bits)
    W, H = SEARCH_W, SEARCH_H
    out = []
    for method in range(5):
        for subme in SEARCH_SUBME:
            for (w, h) in SEARCH_SHAPES:
                if method == J.UMH and w == 4 and h == 4:
                    continue  # the reference asserts on UMH with 4x4 (motion.cpp:447)
                for mode in range(3):
                    merange = 16 if method == J.FULL else int(rng.choice([16, 64, 57]))
                    n = 1 if method == J.FULL else 3
                    px = rng.integers(0, (W - w) // 4 + 1, n) * 4
                    py = rng.integers(0, (H - h) // 4 + 1, n) * 4
                    if rng.random() < 0.25:   # picture corners / edges: clipped ranges
                        px[0] = 0 if rng.random() < 0.5 else W - w
                        py[0] = 0 if rng.random() < 0.5 else H - h
                    if mode == 0:
                        qmvp, mvc = (0, 0), None
                    elif mode == 1:
                        qmvp, mvc = (int(rng.integers(-60, 60)), int(rng.integers(-60, 60))), None
                    else:
                        qmvp = (int(rng.integers(-200, 200)), int(rng.integers(-200, 200)))
                        k = int(rng.integers(1, 5))
                        mvc = rng.integers(-150, 150, (k, 2))
                        if rng.random() < 0.3:
                            mvc[0] = 0
                    out.append(J.make_jobs(W, H, px, py, w, h, method=method, subme=subme, merange=merange,
                                           qp=int(rng.choice([22, 32, 12, 45])), qmvp=qmvp, mvc=mvc,
                                           ref_idx=int(rng.integers(0, 2))))
    return np.concatenate(out)


FIELD_W, FIELD_H = 256, 128


def field_frames():
    return synth.make_sequence(FIELD_W, FIELD_H, 2, seed=4242, bit_depth=8, kind="motion")


def field_jobs() -> np.ndarray:
    """Whole-frame FULL-search field (all 64/32/16/8 square PUs, merange 24, subme 0)."""
    return J.frame_jobs(FIELD_W, FIELD_H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=24, qp=32)


# bidirectional refinement (TEncSearch::xMotionEstimation)
def bipred_frames(bits: int):
    """(modified source plane = clip(2 * org - other prediction), reference plane)"""
    depth = 8 if bits == 8 else 10
    fr = synth.make_sequence(SEARCH_W, SEARCH_H, 3, seed=91 + bits, bit_depth=depth, kind="motion")
    mod = np.clip(2 * fr[0].astype(np.int32) - fr[2].astype(np.int32), 0, (1 << depth) - 1).astype(fr[0].dtype)
    return mod, fr[1]


def bipred_jobs(bits: int) -> np.ndarray:
    """searches of +-bipredSearchRange (4, also 2 and 7) around an initial MV, every PU shape"""
    rng = np.random.default_rng(7000 + bits)
    W, H = SEARCH_W, SEARCH_H
    out = []
    for (w, h) in SEARCH_SHAPES:
        for rep in range(6):
            px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
            py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
            j = J.make_jobs(W, H, [px], [py], w, h, method=J.BIPRED, subme=0, merange=4, qp=int(rng.choice([22, 32, 40])),
                            qmvp=(int(rng.integers(-80, 80)), int(rng.integers(-80, 80))))
            cx, cy = int(rng.integers(-20, 20)), int(rng.integers(-20, 20))       # the uni-directional MV, full-pel
            r = int(rng.choice([4, 4, 2, 7]))
            j["mvmin_x"], j["mvmax_x"], j["mvmin_y"], j["mvmax_y"] = cx - r, cx + r, cy - r, cy + r
            if rep == 5:
                j["mvmax_x"] = cx - r + int(rng.integers(0, 3))                  # narrow ranges: the single-SAD tail
            out.append(j)
    return np.concatenate(out)


# lookahead (half-resolution) path
LOWRES_W, LOWRES_H = 364, 214          # halves to 182 x 107: neither a multiple of 8, stride rounds up to 32
LOOKAHEAD_CONFIGS = ((1, 1, 16, 12), (0, 2, 16, 12), (3, 3, 8, 22), (2, 5, 16, 12), (4, 1, 4, 30))  # method, subme, merange, qp


def lowres_source_frames(bits: int):
    """full-resolution frames whose half-resolution planes the suites use"""
    return synth.make_sequence(LOWRES_W, LOWRES_H, 2, seed=555 + bits, bit_depth=8 if bits == 8 else 10, kind="motion")


def lowres_jobs(geom, bits: int) -> np.ndarray:
    """independent jobs on half-resolution planes: every method and sub-pel level, mostly 8x8 (the
    lookahead's CU) plus other shapes, sub-pel predictors and candidates"""
    rng = np.random.default_rng(5000 + bits)
    W, H = geom.width, geom.lines
    out = []
    for method in range(5):
        for subme in range(8):
            for (w, h) in ((8, 8), (8, 8), (16, 16), (32, 16), (8, 4), (4, 4)):
                if method == J.UMH and w == 4 and h == 4:
                    continue
                n = 2
                px = rng.integers(0, (W - w) // 4 + 1, n) * 4
                py = rng.integers(0, (H - h) // 4 + 1, n) * 4
                qmvp = (int(rng.integers(-40, 40)), int(rng.integers(-40, 40)))
                k = int(rng.integers(0, 5))
                mvc = rng.integers(-60, 60, (k, 2)) if k else None
                out.append(J.make_jobs(W, H, px, py, w, h, method=method, subme=subme,
                                       merange=8 if method == J.FULL else 16, qp=int(rng.choice([12, 22, 32])),
                                       qmvp=qmvp, mvc=mvc))
    return np.concatenate(out)


def plane_crc(planes) -> np.ndarray:
    import zlib
    return np.array([zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in planes], dtype=np.uint32)


def run_lowres_suite(impl, bits: int) -> dict:
    """impl: Reference or Oracle of tests/oracle_ffi.py"""
    fr = lowres_source_frames(bits)
    g, enc4 = impl.lowres_init(fr[0])
    _, ref4 = impl.lowres_init(fr[1])
    out = {"crc_enc": plane_crc(enc4), "crc_ref": plane_crc(ref4)}
    jobs = lowres_jobs(g, bits)
    r = impl.search(enc4[0], [ref4], jobs, lowres=True)
    out["jobs"] = r[0] if isinstance(r, tuple) else r
    for k, (method, subme, merange, qp) in enumerate(LOOKAHEAD_CONFIGS):
        r = impl.lookahead_me(enc4, ref4, g.cu_w, g.cu_h, merange, method, subme, qp)
        out["field%d" % k] = r[0] if isinstance(r, tuple) else r
    return out


# ----------------------------------------------------------------------------------------------
# Suite runners.  `impl` is any object with the primitive-level interface of
# tests/oracle_ffi.py (Reference / Oracle) or the product adapter of the gpu tests.

def run_prim_suite(impl, bits: int) -> dict:
    a, b, adversarial = prim_buffers(bits)
    res = {}
    cmpv = np.zeros((3, 64, PRIM_ITERS), dtype=np.int32)
    for kind in range(3):
        for part in range(64):
            for it in range(PRIM_ITERS):
                cmpv[kind, part, it] = impl.cmp(kind, part, a, 0, PRIM_STRIDE, b, it * PRIM_INCR, PRIM_STRIDE)
    res["cmp"] = cmpv
    res["sa8d_sq"] = np.array([impl.cmp(3, i, a, 0, PRIM_STRIDE, b, 5, PRIM_STRIDE) for i in range(5)], dtype=np.int32)
    adv = np.zeros((len(adversarial), 3, 64), dtype=np.int32)
    for i, (p, q) in enumerate(adversarial):
        for kind in range(3):
            for part in range(64):
                adv[i, kind, part] = impl.cmp(kind, part, p, 0, PRIM_STRIDE, q, 0, PRIM_STRIDE)
    res["adv"] = adv
    # sad_x3 / sad_x4 with the harness's odd reference stride 59 and refs at +0..+3
    x3 = np.zeros((64, PRIM_ITERS, 3), dtype=np.int32)
    x4 = np.zeros((64, PRIM_ITERS, 4), dtype=np.int32)
    for part in range(64):
        for it in range(PRIM_ITERS):
            j = it * PRIM_INCR
            x3[part, it] = impl.sad_xn(3, part, a, 0, b, [j, j + 1, j + 2], PRIM_STRIDE - 5)
            x4[part, it] = impl.sad_xn(4, part, a, 0, b, [j, j + 1, j + 2, j + 3], PRIM_STRIDE - 5)
    res["x3"] = x3
    res["x4"] = x4
    return res


def run_ip_suite(impl, bits: int, depth: int) -> dict:
    """impl.ipfilter_run(kind, src2d, x0, y0, w, h, frac, depth) -> 2-D output array"""
    src, ssrc = ip_source(bits)
    out = {}
    for kind in range(4):
        for frac in (1, 2, 3):
            for (w, h) in IP_SIZES:
                s = ssrc if kind == 3 else src
                o = impl.ipfilter_run(kind, s, 8, 8, w, h, frac, depth)
                out["k%d_f%d_%dx%d" % (kind, frac, w, h)] = np.ascontiguousarray(o)
    return out


# whole-frame fields with predInterSearch's predictor chain: (pu_w, pu_h, merange, method, subme, qp, content)
CHAIN_W, CHAIN_H = 208, 120
CHAIN_CASES = ((16, 16, 32, J.HEX, 5, 30, "motion"), (8, 8, 16, J.DIA, 2, 22, "motion"), (32, 32, 24, J.STAR, 3, 37, "motion"),
               (64, 64, 16, J.UMH, 1, 27, "noise"), (16, 8, 20, J.HEX, 7, 33, "noise"), (8, 8, 6, J.FULL, 0, 30, "motion"),
               (16, 16, 64, J.STAR, 5, 32, "motion"))


def chain_frames(bits: int, kind: str):
    return synth.make_sequence(CHAIN_W, CHAIN_H, 2, seed=4242 + bits, bit_depth=8 if bits == 8 else 10, kind=kind)


def run_chain_suite(impl, bits: int) -> dict:
    """impl.field_run(enc, ref, pu_w, pu_h, merange, method, subme, qp) -> (results[gh, gw], mvp[gh, gw, 2])"""
    out = {}
    for k, (pw, ph, mr, method, subme, qp, kind) in enumerate(CHAIN_CASES):
        fr = chain_frames(bits, kind)
        res, mvp = impl.field_run(fr[1], fr[0], pw, ph, mr, method, subme, qp)
        out["res%d" % k] = np.ascontiguousarray(res)
        out["mvp%d" % k] = np.ascontiguousarray(mvp)
    return out


# the same for the parts of 2NxN (part_mode 1) and Nx2N (2) CUs: (pu_w, pu_h, part_mode, merange, method, subme, qp, content);
# 208 x 120 leaves an odd PU row out of the 2NxN grids of 8- and 24-pixel-high parts
CHAIN_PART_CASES = ((16, 8, 1, 24, J.HEX, 5, 30, "motion"), (8, 16, 2, 24, J.HEX, 5, 30, "motion"), (32, 16, 1, 16, J.STAR, 3, 36, "motion"),
                    (16, 32, 2, 20, J.UMH, 2, 26, "noise"), (8, 4, 1, 12, J.DIA, 1, 33, "motion"), (4, 8, 2, 8, J.FULL, 0, 28, "motion"),
                    (64, 32, 1, 32, J.HEX, 7, 31, "noise"), (32, 64, 2, 16, J.DIA, 4, 24, "motion"))


def run_chain_parts_suite(impl, bits: int) -> dict:
    """impl.field_run(enc, ref, pu_w, pu_h, merange, method, subme, qp, part_mode=m) -> (results[gh, gw], mvp[gh, gw, 2])"""
    out = {}
    for k, (pw, ph, pm, mr, method, subme, qp, kind) in enumerate(CHAIN_PART_CASES):
        fr = chain_frames(bits, kind)
        res, mvp = impl.field_run(fr[1], fr[0], pw, ph, mr, method, subme, qp, part_mode=pm)
        out["res%d" % k] = np.ascontiguousarray(res)
        out["mvp%d" % k] = np.ascontiguousarray(mvp)
    return out


# the same for the asymmetric partitions of a CU of cu x cu (part_mode 4 = 2NxnU, 5 = 2NxnD, 6 = nLx2N, 7 = nRx2N, numbered as
# PartSize, Lib/TLibCommon/TypeDef.h:100-111): (cu, part_mode, merange, method, subme, qp, content); every one of the twelve
# asymmetric PU shapes (16x4 .. 48x64) appears, under every method
CHAIN_AMP_CASES = ((16, 4, 24, J.HEX, 5, 30, "motion"), (16, 5, 16, J.DIA, 2, 28, "motion"), (16, 6, 20, J.STAR, 3, 33, "motion"),
                   (16, 7, 12, J.UMH, 1, 26, "noise"), (32, 4, 24, J.UMH, 4, 31, "motion"), (32, 5, 20, J.STAR, 6, 25, "noise"),
                   (32, 6, 16, J.DIA, 3, 36, "motion"), (32, 7, 16, J.HEX, 7, 35, "noise"), (64, 4, 16, J.STAR, 2, 32, "motion"),
                   (64, 5, 32, J.HEX, 3, 29, "motion"), (64, 6, 8, J.FULL, 0, 30, "motion"), (64, 7, 24, J.UMH, 5, 27, "motion"),
                   (16, 4, 5, J.FULL, 2, 34, "noise"))


def run_chain_amp_suite(impl, bits: int) -> dict:
    """impl.field_run(enc, ref, cu, cu, merange, method, subme, qp, part_mode=m) -> (results[gh, gw], mvp[gh, gw, 2])"""
    out = {}
    for k, (cu, pm, mr, method, subme, qp, kind) in enumerate(CHAIN_AMP_CASES):
        fr = chain_frames(bits, kind)
        res, mvp = impl.field_run(fr[1], fr[0], cu, cu, mr, method, subme, qp, part_mode=pm)
        out["res%d" % k] = np.ascontiguousarray(res)
        out["mvp%d" % k] = np.ascontiguousarray(mvp)
    return out


# the same with the temporal AMVP candidate (TComDataCU::fillMvpCand, Lib/TLibCommon/TComDataCU.cpp:2635-2697): a co-located
# motion field per 16x16 block (a common drift plus noise, a fifth of the blocks without motion) joins the candidate list where
# fewer than two spatial candidates exist: (pu_w, pu_h, part_mode, merange, method, subme, qp, content)
CHAIN_TMVP_CASES = ((16, 16, 0, 32, J.HEX, 5, 30, "motion"), (8, 8, 0, 16, J.DIA, 2, 24, "motion"), (32, 32, 0, 24, J.STAR, 3, 35, "noise"),
                    (64, 64, 0, 16, J.UMH, 1, 28, "motion"), (16, 8, 1, 20, J.HEX, 7, 32, "motion"), (8, 16, 2, 12, J.UMH, 4, 27, "motion"),
                    (32, 32, 4, 24, J.HEX, 3, 31, "motion"), (16, 16, 7, 16, J.STAR, 2, 29, "noise"), (64, 64, 5, 8, J.FULL, 0, 30, "motion"),
                    (32, 32, 6, 16, J.DIA, 6, 33, "motion"), (8, 4, 1, 8, J.HEX, 1, 26, "motion"))


def chain_col_field(bits: int, k: int):
    rng = np.random.default_rng(9000 + bits + 17 * k)
    n = ((CHAIN_W + 15) // 16) * ((CHAIN_H + 15) // 16)
    col = np.zeros(n, dtype=J.RESULT_DTYPE)
    drift = rng.integers(-14, 15, 2)
    col["mvx"] = drift[0] + rng.integers(-5, 6, n)
    col["mvy"] = drift[1] + rng.integers(-5, 6, n)
    col["cost"] = np.where(rng.random(n) < 0.2, -1, 0)
    return col


def run_chain_tmvp_suite(impl, bits: int) -> dict:
    """impl.field_run(enc, ref, pu_w, pu_h, merange, method, subme, qp, part_mode=m, col=c) -> (results[gh, gw], mvp[gh, gw, 2])"""
    out = {}
    for k, (pw, ph, pm, mr, method, subme, qp, kind) in enumerate(CHAIN_TMVP_CASES):
        fr = chain_frames(bits, kind)
        res, mvp = impl.field_run(fr[1], fr[0], pw, ph, mr, method, subme, qp, part_mode=pm, col=chain_col_field(bits, k))
        out["res%d" % k] = np.ascontiguousarray(res)
        out["mvp%d" % k] = np.ascontiguousarray(mvp)
    return out
