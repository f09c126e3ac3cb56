#!/usr/bin/env python3
"""Developer aid: host-visible latency of small synchronous batches (x265amd_me_search_batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=5)
ids = [eng.upload_plane(f) for f in fr]
for n, size, method in ((1, 16, J.HEX), (1, 64, J.HEX), (1, 16, J.BIPRED), (30, 16, J.HEX), (120, 16, J.HEX), (8040, 16, J.HEX)):
    x, y = J.square_pu_grid(W, H, size)
    jobs = J.make_jobs(W, H, x[:n], y[:n], size, size, method=method, subme=5, merange=64, qp=32)
    if method == J.BIPRED:
        jobs["mvmin_x"], jobs["mvmax_x"], jobs["mvmin_y"], jobs["mvmax_y"] = -4, 4, -4, 4
    for _ in range(5):
        eng.search(ids[0], [ids[1]], jobs)
    K = 200 if n < 1000 else 20
    t0 = time.perf_counter()
    for _ in range(K):
        eng.search(ids[0], [ids[1]], jobs)
    dt = (time.perf_counter() - t0) / K
    print("%5d job(s) %dx%d method %d: %.1f us per call (%.2f us per job)" % (n, size, size, method, dt * 1e6, dt * 1e6 / n), flush=True)
