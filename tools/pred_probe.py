"""FULL +-64 frame (1080p, all four square PU levels) whose PUs have their own predictors, scattered `spread` quarter-pels
around a vector per 64x64 cell: time per run of the planned list.  X265AMD_FLAGS=256 keeps PUs with different predictors in
separate bundles (each then scans on its own)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from x265_amd.api import Engine, JobList
from x265_amd import jobs as J, synth
eng = Engine(0)
W, H = 1920, 1080
fr = synth.make_sequence(W, H, 2, seed=7)
ids = [eng.upload_plane(f) for f in fr]
rng = np.random.default_rng(5)
def mk(spread):
    parts = []
    cells = rng.integers(-20, 21, size=(32 * 17, 2))
    for s in (64, 32, 16, 8):
        x, y = J.square_pu_grid(W, H, s)
        cell = (y // 64) * 32 + x // 64
        base = cells[cell]
        q = base + (rng.integers(-spread, spread + 1, size=(x.size, 2)) if spread else 0)
        parts.append(J.make_jobs(W, H, x, y, s, s, method=J.FULL, subme=0, merange=64, qp=32, qmvp=(q[:, 0].astype(np.int16), q[:, 1].astype(np.int16))))
    return np.concatenate(parts)
for spread in (0, 3, 9, 20):
    jobs = mk(spread)
    L = JobList(eng, jobs, 1, 1)
    t = []
    for k in range(6):
        t0 = time.perf_counter()
        L.run(ids[0], [ids[1]], fetch=False); eng.sync()
        t.append(time.perf_counter() - t0)
    print("spread", spread, "n", jobs.size, "ms", ["%.3f" % (1e3 * v) for v in t], flush=True)
