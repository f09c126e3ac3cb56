#!/usr/bin/env python3
"""The pipelined PCIe-inclusive frame loop alone (for rocprofv3 --kernel-trace): planes of frame k + 1 refilled through the
library beside the search of frame k, field of frame k downloaded behind its search."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from x265_amd import synth
from x265_amd.api import Engine

W, H = bench.W, bench.H
dev = torch.device("cuda", 0)
eng = Engine(0)
base = synth.base_frame(W, H)
ref_np = synth.pad_plane(base); enc_np = synth.pad_plane(synth.moved_frame(base, 1))
enc_pin = torch.from_numpy(enc_np).pin_memory(); ref_pin = torch.from_numpy(ref_np).pin_memory()
_, full_jobs = bench.build_job_lists()
sets = [{"enc_id": eng.upload_plane(enc_np), "ref_id": eng.upload_plane(ref_np)} for _ in range(2)]
jls = [eng.make_joblist(full_jobs) for _ in range(2)]
outs = [torch.empty(len(full_jobs) * 8, dtype=torch.uint8).pin_memory() for _ in range(2)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for rep in range(2):
    eng.sync()
    t0 = time.perf_counter()
    stamps = []
    for k in range(n):
        nxt = sets[(k + 1) & 1]
        a = time.perf_counter()
        eng.refill_plane_async(nxt["enc_id"], enc_pin.data_ptr())
        b = time.perf_counter()
        eng.refill_plane_async(nxt["ref_id"], ref_pin.data_ptr())
        c = time.perf_counter()
        st = sets[k & 1]
        jls[k & 1].run(st["enc_id"], [st["ref_id"]], fetch=False)
        d = time.perf_counter()
        jls[k & 1].results_async(outs[k & 1].data_ptr())
        e = time.perf_counter()
        stamps.append((b - a, c - b, d - c, e - d))
    t1 = time.perf_counter()
    print("host us per call (refill enc, refill ref, run, results_async), last 5 frames:", [[round(x * 1e6) for x in s] for s in stamps[-5:]])
    eng.sync()
    print("pass %d: %.3f ms per frame (host queueing %.3f ms per frame)" % (rep, (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3))
