#!/usr/bin/env python3
"""Timing of the lookahead path on a 1080p sequence: plane-set construction and whole-frame searches
for 1..N frame pairs per call (the reference's lookahead settings: HEX, subme 1, merange 16, qp 12)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from x265_amd import synth
from x265_amd.api import Engine
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
eng = Engine(0)
fr = synth.make_sequence(W, H, 4, seed=5)
src = [eng.upload_plane(f) for f in fr]
lr = [eng.lowres_create(s) for s in src]
eng.sync()
t0 = time.perf_counter()
for _ in range(20):
    for a, s in zip(lr, src):
        eng.lowres_update(a, s)
eng.sync()
dt = (time.perf_counter() - t0) / 80
print("lowres_update %dx%d: %.1f us per picture (%.1f GB/s of source + plane bytes)" %
      (W, H, dt * 1e6, (fr[0].nbytes + 4 * fr[0].nbytes / 4) / dt / 1e9))
eng.set_timing(True)
_, _, _, cw, ch = eng.lowres_geometry(lr[0])
for n in (1, 2, 4, 8, 16, 32, 64):
    enc = [lr[(i % 3) + 1] for i in range(n)]
    ref = [lr[i % 3 if (i % 3) != (i % 3) + 1 else 0] for i in range(n)]
    eng.eval_count(reset=True)
    best = 1e9
    for _ in range(3):
        eng.lookahead_me(enc, ref, fetch=False)
        best = min(best, eng.kernel_times()[1])
    ev = eng.eval_count(reset=True) / 3
    print("lookahead_me %3d pairs (%d x %d CUs each): %8.3f ms  = %7.3f ms per pair, %.3g evals/s, %.0f half-res frames/s" %
          (n, cw, ch, best, best / n, ev / (best * 1e-3), n / (best * 1e-3)))
