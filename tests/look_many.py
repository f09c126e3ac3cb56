import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from x265_amd import synth
from x265_amd.api import Engine
from oracle_ffi import Oracle
eng = Engine(0); o = Oracle()
fr = synth.make_sequence(1920, 1080, 2, seed=5)
src = [eng.upload_plane(f) for f in fr]
lr = [eng.lowres_create(s) for s in src]
g, e4 = o.lowres_init(fr[1]); _, r4 = o.lowres_init(fr[0])
exp, _ = o.lookahead_me(e4, r4, g.cu_w, g.cu_h, 16, 1, 1, 12)
for n in (128, 512):
    t0 = time.perf_counter()
    got = eng.lookahead_me([lr[1]] * n, [lr[0]] * n)
    dt = time.perf_counter() - t0
    ok = all(np.array_equal(got[i], exp) for i in range(n))
    print("pairs %d: %.1f ms (%.3f ms per pair), all equal to the oracle: %s" % (n, dt * 1e3, dt * 1e3 / n, ok), flush=True)
