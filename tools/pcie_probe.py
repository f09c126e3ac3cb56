#!/usr/bin/env python3
"""Where does a PCIe-inclusive FULL frame's time go?  Variants of the pipelined frame loop of bench.py (planes up on a copy
stream, search, field down), timed over 50 frames each."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from x265_amd import jobs as J, synth
from x265_amd.api import Engine

W, H = bench.W, bench.H
dev = torch.device("cuda", 0)
eng = Engine(0)
base = synth.base_frame(W, H)
ref_np = synth.pad_plane(base); enc_np = synth.pad_plane(synth.moved_frame(base, 1))
stride = ref_np.shape[1]
enc_pin = torch.from_numpy(enc_np).pin_memory(); ref_pin = torch.from_numpy(ref_np).pin_memory()
_, full_jobs = bench.build_job_lists()
copy_stream = torch.cuda.Stream(device=dev)
sets = []
for b in range(2):
    e_t = torch.from_numpy(enc_np).to(dev); r_t = torch.from_numpy(ref_np).to(dev)
    sets.append({"enc": e_t, "ref": r_t, "enc_id": eng.wrap_device_plane(e_t.data_ptr(), W, H, stride, 1, 8),
                 "ref_id": eng.wrap_device_plane(r_t.data_ptr(), W, H, stride, 1, 8),
                 "filled": torch.cuda.Event(), "free": torch.cuda.Event(), "used": False})
jls = [eng.make_joblist(full_jobs) for _ in range(2)]
outs = [torch.empty(len(full_jobs) * 8, dtype=torch.uint8).pin_memory() for _ in range(2)]

def upload(k):
    st = sets[k & 1]
    with torch.cuda.stream(copy_stream):
        if st["used"]:
            copy_stream.wait_event(st["free"])
        st["enc"].copy_(enc_pin, non_blocking=True)
        st["ref"].copy_(ref_pin, non_blocking=True)
        st["filled"].record(copy_stream)

def frame_prefetch(k, down=True):
    """planes of frame k + 1 are queued BEFORE the search of frame k"""
    if k == 0 or not sets[k & 1].get("ahead"):
        upload(k)
    upload(k + 1)
    sets[(k + 1) & 1]["ahead"] = True
    st = sets[k & 1]
    eng.wait_event(st["filled"])
    jls[k & 1].run(st["enc_id"], [st["ref_id"]], fetch=False)
    if down:
        jls[k & 1].results_async(outs[k & 1].data_ptr())
    eng.record_event(st["free"])
    st["used"] = True

def frame(k, up=True, down=True, search=True, events=True, same_stream=False, prefetch=False):
    if prefetch:
        return frame_prefetch(k, down)
    st = sets[k & 1]
    if up and same_stream:
        es = eng.torch_stream()
        with torch.cuda.stream(es):
            st["enc"].copy_(enc_pin, non_blocking=True)
            st["ref"].copy_(ref_pin, non_blocking=True)
    elif up:
        with torch.cuda.stream(copy_stream):
            if st["used"] and events:
                copy_stream.wait_event(st["free"])
            st["enc"].copy_(enc_pin, non_blocking=True)
            st["ref"].copy_(ref_pin, non_blocking=True)
            if events:
                st["filled"].record(copy_stream)
        if events:
            eng.wait_event(st["filled"])
    if search:
        jls[k & 1].run(st["enc_id"], [st["ref_id"]], fetch=False)
    if down:
        jls[k & 1].results_async(outs[k & 1].data_ptr())
    if events and up and not same_stream:
        eng.record_event(st["free"])
        st["used"] = True

def timed(name, **kw):
    for st in sets:
        st["used"] = False
        st["ahead"] = False
    for k in range(4):
        frame(k, **kw)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(50):
        frame(k, **kw)
    eng.sync(); torch.cuda.synchronize()
    print("%-52s %.3f ms per frame" % (name, (time.perf_counter() - t0) / 50 * 1e3), flush=True)

for k in range(2):
    jls[k].run(sets[k]["enc_id"], [sets[k]["ref_id"]], fetch=False)
timed("search only (resident)", up=False, down=False)
timed("search + field down", up=False, down=True)
timed("planes up (copy stream, events) + search", up=True, down=False)
timed("planes up + search + field down (pipelined)", up=True, down=True)
timed("planes up on the copy stream, no events, + search", up=True, down=False, events=False)
timed("planes up only (copy stream)", up=True, down=False, search=False, events=False)
timed("planes up on the engine's stream + search + down", up=True, down=True, same_stream=True)
timed("planes of k + 1 queued before search k + down", prefetch=True)
timed("the same without the download", prefetch=True, down=False)

def frame_lib(k, down=True):
    """the library's own refill: planes of frame k + 1 queued before the search of frame k"""
    nxt = sets[(k + 1) & 1]
    eng.refill_plane_async(nxt["enc_id"], enc_pin.data_ptr())
    eng.refill_plane_async(nxt["ref_id"], ref_pin.data_ptr())
    st = sets[k & 1]
    jls[k & 1].run(st["enc_id"], [st["ref_id"]], fetch=False)
    if down:
        jls[k & 1].results_async(outs[k & 1].data_ptr())

for name, dn in (("library refill of k + 1 before search k + down", True), ("the same without the download", False)):
    for k in range(4):
        frame_lib(k, dn)
    eng.sync()
    t0 = time.perf_counter()
    for k in range(50):
        frame_lib(k, dn)
    eng.sync()
    print("%-52s %.3f ms per frame" % (name, (time.perf_counter() - t0) / 50 * 1e3), flush=True)
