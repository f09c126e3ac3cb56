#!/usr/bin/env python3
"""Developer aid (not collected by pytest): randomised parity campaign of the two chained whole-frame paths on
the GPU - the lookahead search over half-resolution plane sets and the predictor-chained field - against the
CPU oracle.  Random picture sizes (also ones that are no multiple of the CU / PU size), content kinds, pixel
types, search methods, sub-pel levels, ranges and QPs.

usage: python tests/fuzz_chain.py SECONDS [SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from oracle_ffi import Oracle
from x265_amd import synth
from x265_amd.api import Engine


def make_frames(rng, w, h, depth, kind):
    if kind == "motion":
        return synth.make_sequence(w, h, 2, seed=int(rng.integers(1, 1 << 30)), bit_depth=depth)
    top = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    out = []
    for _ in range(2):
        if kind == "noise":
            p = rng.integers(0, top + 1, size=(h, w))
        else:                               # flat with a few steps: many ties
            p = np.full((h, w), int(rng.integers(0, top + 1)))
            for _ in range(4):
                x, y = int(rng.integers(0, w)), int(rng.integers(0, h))
                p[y:, x:] = int(rng.integers(0, top + 1))
        out.append(synth.pad_plane(p.astype(dt), depth))
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    eng, oracle = Engine(0), Oracle()
    t0 = time.time()
    n_look = n_field = n_calls = cus = pus = 0
    while time.time() - t0 < budget:
        bits = int(rng.choice([8, 16]))
        depth = 8 if bits == 8 else 10
        kind = str(rng.choice(["motion", "motion", "noise", "flat"]))
        if rng.random() < 0.5:
            # ---- lookahead: half-resolution sets, 8x8 CUs, neighbour-predicted chain
            w, h = 2 * int(rng.integers(8, 330)), 2 * int(rng.integers(8, 190))
            fr = make_frames(rng, w, h, depth, kind)
            method, subme = int(rng.integers(0, 5)), int(rng.integers(0, 8))
            merange, qp = int(rng.choice([8, 16, 24])), int(rng.integers(8, 40))
            if method == 4:                 # (exhaustive: keep the oracle's share of the time small)
                merange = 8
            src = [eng.upload_plane(f, bit_depth=depth) for f in fr]
            lr = [eng.lowres_create(s) for s in src]
            got = eng.lookahead_me([lr[1], lr[0]], [lr[0], lr[1]], merange=merange, method=method, subme=subme, qp=qp)
            cpu = [oracle.lowres_init(f) for f in fr]
            g = cpu[0][0]
            for i, (a, b) in enumerate(((1, 0), (0, 1))):
                exp, _ = oracle.lookahead_me(cpu[a][1], cpu[b][1], g.cu_w, g.cu_h, merange, method, subme, qp, bit_depth=depth)
                if not np.array_equal(got[i], exp):
                    bad = np.nonzero(got[i] != exp)[0]
                    print("MISMATCH lookahead %dx%d bits %d depth %d %s method %d subme %d merange %d qp %d pair %d: CUs %s"
                          % (w, h, bits, depth, kind, method, subme, merange, qp, i, bad[:5]), flush=True)
                    sys.exit(1)
            for p in lr + src:
                eng.release_plane(p)
            n_look += 1
            cus += 2 * g.cu_w * g.cu_h
            what = "lookahead %4dx%-4d" % (w, h)
        else:
            # ---- predictor-chained fields, one to four in one call, each with its own PU size and partition mode
            #      (0 2Nx2N, 1 2NxN, 2 Nx2N, 4..7 the asymmetric ones) and, for half of the calls, co-located motion
            def draw_shape():
                pm = int(rng.choice([0, 0, 1, 2, 4, 5, 6, 7]))
                if pm >= 4:
                    pw = ph = int(rng.choice([16, 32, 64]))
                elif pm:
                    pw, ph = [(16, 8), (32, 16), (64, 32), (8, 4), (16, 16), (8, 8)][int(rng.integers(0, 6))]
                    if pm == 2:
                        pw, ph = ph, pw
                else:
                    pw, ph = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16)][int(rng.integers(0, 7))]
                return pw, ph, pm
            shapes = [draw_shape() for _ in range(int(rng.choice([1, 1, 2, 3, 4])))]
            cu_w = max(pw * (2 if pm == 2 else 1) for pw, ph, pm in shapes)
            cu_h = max(ph * (2 if pm == 1 else 1) for pw, ph, pm in shapes)
            w, h = 8 * int(rng.integers(max(2, cu_w // 8), 60)), 8 * int(rng.integers(max(2, cu_h // 8), 36))
            fr = make_frames(rng, w, h, depth, kind)
            method, subme = int(rng.integers(0, 5)), int(rng.integers(0, 8))
            merange, qp = int(rng.choice([8, 16, 32, 57])), int(rng.integers(10, 45))
            if method == 4:
                merange = 8
            ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
            cols = None
            if rng.random() < 0.5:                  # co-located motion fields for the temporal candidate, one per field
                from x265_amd.jobs import RESULT_DTYPE
                nb = ((w + 15) // 16) * ((h + 15) // 16)
                cols = np.zeros((len(shapes), nb), dtype=RESULT_DTYPE)
                for col in cols:
                    spread = int(rng.choice([3, 12, 60]))
                    col["mvx"] = rng.integers(-spread, spread + 1, nb) + int(rng.integers(-20, 21))
                    col["mvy"] = rng.integers(-spread, spread + 1, nb) + int(rng.integers(-20, 21))
                    col["cost"] = np.where(rng.random(nb) < rng.choice([0.0, 0.2, 0.7]), -1, 0)
            n = len(shapes)
            got_all = eng.field_me_multi([ids[1]] * n, [ids[0]] * n, [s[0] for s in shapes], [s[1] for s in shapes], [s[2] for s in shapes],
                                         merange, method, subme, qp, col=cols)
            for k, (pw, ph, pm) in enumerate(shapes):
                col = cols[k] if cols is not None else None
                exp, exp_mvp, _ = oracle.field_me(fr[1], fr[0], pw, ph, merange, method, subme, qp, bit_depth=depth, part_mode=pm, col=col)
                got, got_mvp = got_all[k]
                if not (np.array_equal(got, exp) and np.array_equal(got_mvp, exp_mvp)):
                    print("MISMATCH field %d of %s (col %s) %dx%d bits %d depth %d %s method %d subme %d merange %d qp %d: PUs %s"
                          % (k, shapes, col is not None, w, h, bits, depth, kind, method, subme, merange, qp, np.argwhere(got != exp)[:5]), flush=True)
                    sys.exit(1)
                n_field += 1
                pus += got.size
            for p in ids:
                eng.release_plane(p)
            n_calls += 1
            what = "fields %4dx%-4d %s%s" % (w, h, shapes, " +col" if cols is not None else "")
        if (n_look + n_calls) % 20 == 0:
            print("case %d ok: %s %2d-bit depth %2d %-6s (%d s)" % (n_look + n_calls, what, bits, depth, kind, time.time() - t0), flush=True)
    print("chained-path campaign: %d lookahead cases (%d CU searches) and %d chained fields in %d calls (%d PU searches), all bit-exact "
          "against the oracle (seed %d, %.0f s)" % (n_look, cus, n_field, n_calls, pus, seed, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
