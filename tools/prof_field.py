#!/usr/bin/env python3
"""Developer aid: kernel time of the chained whole-frame field (x265amd_field_me) at 1080p."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=91)
ids = [eng.upload_plane(f) for f in fr]
eng.set_timing(True)
for pu, method, subme in ((16, J.HEX, 5), (8, J.HEX, 5), (32, J.HEX, 5), (64, J.HEX, 5), (16, J.STAR, 5), (16, J.HEX, 1)):
    for _ in range(2):
        eng.field_me(ids[1], ids[0], pu, pu, 64, method, subme, 32, fetch=False)
    eng.eval_count(reset=True)
    eng.field_me(ids[1], ids[0], pu, pu, 64, method, subme, 32, fetch=False)
    ms = eng.kernel_times()[1]
    ev = eng.eval_count()
    n = (W // pu) * (H // pu)
    print("field %2dx%-2d method %d subme %d: %.3f ms per frame, %d PUs, chain %d, %.1f us per chained PU, %.3g evals/s"
          % (pu, pu, method, subme, ms, n, W // pu + 2 * (H // pu), ms * 1e3 / (W // pu + 2 * (H // pu)), ev / (ms * 1e-3)), flush=True)
for nf in (2, 4, 8, 16):
    for _ in range(2):
        eng.field_me_batch([ids[1]] * nf, [ids[0]] * nf, 16, 16, 64, J.HEX, 5, 32, fetch=False)
    ms = eng.kernel_times()[1]
    print("field 16x16 HEX subme 5, %2d fields in one call: %.3f ms (%.3f ms per field)" % (nf, ms, ms / nf), flush=True)
