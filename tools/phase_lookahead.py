#!/usr/bin/env python3
"""Developer aid: per-phase shader-clock totals of the lookahead kernel (library built with -DXA_PROFILE,
X265AMD_LIB pointing at it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import synth
from x265_amd.api import Engine
eng = Engine(0)
fr = synth.make_sequence(1920, 1080, 2, seed=5)
src = [eng.upload_plane(f) for f in fr]
lr = [eng.lowres_create(s) for s in src]
eng.set_timing(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for _ in range(3):
    eng.lookahead_me([lr[1]] * n, [lr[0]] * n, fetch=False)
eng.eval_count(reset=True)
eng.lookahead_me([lr[1]] * n, [lr[0]] * n, fetch=False)
print("pairs", n, "kernel ms", eng.kernel_times()[1])
sys.stderr.flush()
print("evals", eng.eval_count())
