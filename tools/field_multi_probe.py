"""Time of x265amd_field_me_multi on one 1080p frame: every PU size / partition mode alone, then all in one call."""
import sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from x265_amd.api import Engine
from x265_amd import jobs as J, synth
eng = Engine(0)
W, H = 1920, 1080
fr = synth.make_sequence(W, H, 2, seed=92)
ids = [eng.upload_plane(f) for f in fr]
shapes = [(8, 8, 0), (16, 16, 0), (32, 32, 0), (64, 64, 0), (16, 8, 1), (8, 16, 2), (32, 16, 1), (16, 32, 2)]
def run(sel, reps=3):
    n = len(sel)
    args = ([ids[1]] * n, [ids[0]] * n, [s[0] for s in sel], [s[1] for s in sel], [s[2] for s in sel], 32, J.HEX, 3, 30)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.field_me_multi(*args, fetch=False)
        eng.sync()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best
for s in shapes:
    gw, gh = eng.field_grid(W, H, *s)
    t = run([s])
    print("%2dx%-2d mode %d: %6.2f ms  %d PUs  %.1f us per chain step" % (s[0], s[1], s[2], t, gw * gh, 1e3 * t / ((2 if s[2] == 1 else 1) * (gw + 2 * (gh // (2 if s[2] == 1 else 1))))), flush=True)
print("squares together: %.2f ms" % run(shapes[:4]))
print("8x8 + 16x16: %.2f ms" % run(shapes[:2]))
print("all eight: %.2f ms" % run(shapes))
amp = [(c, c, m) for c in (16, 32, 64) for m in (4, 5, 6, 7)]
print("twelve asymmetric fields (16 / 32 / 64 x modes 4..7): %.2f ms" % run(amp))
print("all twenty: %.2f ms" % run(shapes + amp))
