#!/usr/bin/env python3
"""How well does the lookahead's half-resolution field predict the length of the pattern searches, and what is a launch
order from it worth?  1080p HEX + subme 5, 16x16 + 64x64 PUs (the bench's leg)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from x265_amd import jobs as J, synth
from x265_amd.api import Engine

W, H = bench.W, bench.H
eng = Engine(0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
eid, rid = eng.upload_plane(enc), eng.upload_plane(ref)
hex_jobs, _ = bench.build_job_lists()
eng.set_timing(True)
lr_e, lr_r = eng.lowres_create(eid), eng.lowres_create(rid)
_, _, _, cw, ch = eng.lowres_geometry(lr_e)

def t(jl, reps=5):
    best = 1e9
    for _ in range(reps):
        jl.run(eid, [rid], fetch=False)
        best = min(best, eng.kernel_times()[1])
    return best

jl = eng.make_joblist(hex_jobs)
print("planned order               %.4f ms" % t(jl))
actual = jl.eval_counts().astype(np.float64)
cx = np.clip((hex_jobs["pu_x"] + hex_jobs["w"] // 2) >> 4, 0, cw - 1); cy = np.clip((hex_jobs["pu_y"] + hex_jobs["h"] // 2) >> 4, 0, ch - 1)
for mr in (16, 32):
    field = eng.lookahead_me([lr_e], [lr_r], merange=mr)[0].reshape(ch, cw)
    f = field[cy, cx]
    dx = 2 * f["mvx"].astype(np.int64) - hex_jobs["qmvp_x"]; dy = 2 * f["mvy"].astype(np.int64) - hex_jobs["qmvp_y"]
    for name, pred in (("L1", np.abs(dx) + np.abs(dy)), ("Linf", np.maximum(np.abs(dx), np.abs(dy))), ("hexsteps", np.maximum(np.abs(dx) / 2, np.abs(dy)))):
        for s in (16, 64):
            m = hex_jobs["w"] == s
            print("lookahead merange %d  %-8s  %dx%d: correlation with the evaluation counts %.3f" % (mr, name, s, s, np.corrcoef(pred[m], actual[m])[0, 1]))
        jl.order_by_length(pred.astype(np.uint32))
        print("   ordered by it            %.4f ms" % t(jl))
jl.order_by_length(actual.astype(np.uint32))
print("ordered by the true counts  %.4f ms" % t(jl))
# the true displacement of the synthetic sequence per 64x64 block is not available to an encoder; the field of a finer search is
