#!/usr/bin/env python3
"""Developer aid: is the single-wavefront search kernel bound by throughput or by its longest jobs?
Times the 1080p 16x16 HEX + subme 5 frame with every k-th job only, and the evaluation-count spread."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
ids = [eng.upload_plane(enc), eng.upload_plane(ref)]
jobs = J.frame_jobs(W, H, sizes=(16,), method=J.HEX, subme=5, merange=64, qp=32)
eng.set_timing(True)
for k in (1, 2, 4, 8, 16, 64):
    jl = eng.make_joblist(jobs[::k])
    best = 1e9
    for _ in range(4):
        jl.run(ids[0], ids[1:], fetch=False)
        best = min(best, eng.kernel_times()[1])
    jl.close()
    print("every %2d-th job (%5d jobs): %.4f ms" % (k, len(jobs[::k]), best), flush=True)
# per-job evaluation counts from the CPU restatement (the reference's counts)
from oracle_ffi import Oracle
o = Oracle()
ev = np.array([o.search(enc, [ref], jobs[i:i + 1])[1] for i in range(0, len(jobs), 16)])
print("evals per job (sample of %d): mean %.1f  p50 %d  p90 %d  p99 %d  max %d" %
      (len(ev), ev.mean(), np.percentile(ev, 50), np.percentile(ev, 90), np.percentile(ev, 99), ev.max()))

# longest jobs first (needs X265AMD_FLAGS=8: the library then keeps the caller's order)
allev = np.array([o.search(enc, [ref], jobs[i:i + 1])[1] for i in range(len(jobs))])
for name, order in (("picture order", np.arange(len(jobs))), ("longest first", np.argsort(-allev, kind="stable")),
                    ("shortest first", np.argsort(allev, kind="stable"))):
    jl = eng.make_joblist(jobs[order])
    best = 1e9
    for _ in range(4):
        jl.run(ids[0], ids[1:], fetch=False)
        best = min(best, eng.kernel_times()[1])
    jl.close()
    print("%-14s: %.4f ms" % (name, best), flush=True)

# the library's own rebalancing (run without X265AMD_FLAGS=8): lengths of the last run, coarse groups;
# and how well one frame's lengths predict the next frame's
enc2 = eng.upload_plane(synth.pad_plane(synth.moved_frame(base, 2)))
jl = eng.make_joblist(jobs)
def t(e):
    best = 1e9
    for _ in range(4):
        jl.run(e, ids[1:], fetch=False)
        best = min(best, eng.kernel_times()[1])
    return best
print("planned order, frame 1: %.4f ms   frame 2: %.4f ms" % (t(ids[0]), t(enc2)))
jl.run(ids[0], ids[1:], fetch=False)
jl.rebalance()
print("rebalanced on frame 1 -> frame 1: %.4f ms   frame 2: %.4f ms" % (t(ids[0]), t(enc2)))
