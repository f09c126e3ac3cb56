#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz by running the COMPILED REFERENCE
(oracle/_ref/libx265ref{8,16}.so, built by oracle/Makefile from the unmodified sources under
/root/reference) on the deterministic inputs of tests/golden_cases.py.

Run in the container that has /root/reference:   python3 oracle/gen_golden.py
The fixtures hold inputs-by-seed and expected outputs only (no reference source text).
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import golden_cases as G  # noqa: E402
from oracle_ffi import Reference  # noqa: E402


class RefIp:
    def __init__(self, ref, bits):
        self.ref, self.bits = ref, bits

    def ipfilter_run(self, kind, src, x0, y0, w, h, frac, depth):
        self.ref.set_bit_depth(depth)
        pdt = np.uint8 if self.bits == 8 else np.uint16
        dst = np.zeros((h, w), dtype=np.int16 if kind == 2 else pdt)
        ss = src.shape[1]
        self.ref.ipfilter(kind, src, y0 * ss + x0, ss, dst, 0, w, w, h, frac)
        return dst


class RefChain:
    def __init__(self, ref):
        self.ref = ref

    def field_run(self, enc, refp, pw, ph, mr, method, subme, qp, part_mode=0, col=None):
        return self.ref.field_me(enc, refp, pw, ph, mr, method, subme, qp, part_mode=part_mode, col=col)


def gen_chain():
    """whole-frame fields with predInterSearch's predictor chain (oracle/ref_shim.cpp: ref_field_me_parts)"""
    for bits in (8, 16):
        ref = Reference(bits)
        ref.set_bit_depth(8 if bits == 8 else 10)
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "chain_u%d.npz" % bits), **G.run_chain_suite(RefChain(ref), bits))
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "chain_parts_u%d.npz" % bits), **G.run_chain_parts_suite(RefChain(ref), bits))
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "chain_amp_u%d.npz" % bits), **G.run_chain_amp_suite(RefChain(ref), bits))
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "chain_tmvp_u%d.npz" % bits), **G.run_chain_tmvp_suite(RefChain(ref), bits))
        print("chain fields, bits", bits)


def main():
    os.makedirs(G.GOLDEN_DIR, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "chain":
        gen_chain()
        return
    gen_chain()
    for bits in (8, 16):
        ref = Reference(bits)
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "prim_u%d.npz" % bits), **G.run_prim_suite(ref, bits))
        for depth in ((8,) if bits == 8 else (8, 10)):
            np.savez_compressed(os.path.join(G.GOLDEN_DIR, "ipfilter_u%d_d%d.npz" % (bits, depth)),
                                **G.run_ip_suite(RefIp(ref, bits), bits, depth))
        # search grid, motion + noise content
        jobs = G.search_jobs(bits)
        depth = 8 if bits == 8 else 10
        ref.set_bit_depth(depth)
        out = {"jobs_crc": np.array([zlib.crc32(jobs.tobytes())], dtype=np.uint32)}
        for kind in ("motion", "noise"):
            fr = G.search_frames(bits, kind)
            out["res_" + kind] = ref.search(fr[0], [fr[1], fr[2]], jobs)
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "search_u%d.npz" % bits), **out)
        print("bits", bits, "search jobs", len(jobs))
        # bidirectional refinement: reference primitives / filters / MV cost under the control flow of
        # xPatternSearch + xPatternRefinement (oracle/ref_shim.cpp: ref_bipred_search)
        bj = G.bipred_jobs(bits)
        mod, rf = G.bipred_frames(bits)
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "bipred_u%d.npz" % bits), res=ref.search(mod, [rf], bj),
                            jobs_crc=np.array([zlib.crc32(bj.tobytes())], dtype=np.uint32))
        # lookahead path: half-resolution planes, isLowres searches, neighbour-predicted MV fields
        np.savez_compressed(os.path.join(G.GOLDEN_DIR, "lowres_u%d.npz" % bits), **G.run_lowres_suite(ref, bits))
    ref = Reference(8)
    tabs = {}
    for qp in G.MV_QPS:
        t = ref.mvcost_table(qp)
        tabs["qp%d" % qp] = t
    # bitcost samples (bitcost.h:49-53)
    rng = np.random.default_rng(9)
    mvp = rng.integers(-300, 300, (200, 2))
    mv = mvp + rng.integers(-500, 500, (200, 2))
    tabs["bc_mvp"] = mvp.astype(np.int32)
    tabs["bc_mv"] = mv.astype(np.int32)
    tabs["bc_out"] = np.array([ref.mvcost_bits(32, p, m) for p, m in zip(mvp.tolist(), mv.tolist())], dtype=np.int32)
    np.savez_compressed(os.path.join(G.GOLDEN_DIR, "mvcost.npz"), **tabs)
    # whole-frame FULL field
    fr = G.field_frames()
    jobs = G.field_jobs()
    res = ref.search(fr[0], [fr[1]], jobs)
    np.savez_compressed(os.path.join(G.GOLDEN_DIR, "field_full.npz"), res=res,
                        jobs_crc=np.array([zlib.crc32(jobs.tobytes())], dtype=np.uint32))
    print("field jobs", len(jobs))


if __name__ == "__main__":
    main()
