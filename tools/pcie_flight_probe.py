"""Developer aid: PCIe-inclusive 1080p exhaustive frames (both planes up, the field down, every frame) with D frames in flight on
the engine's single-stream sets, P plane slots and the refill queued `ahead` frames before its frame; ms per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
base = synth.base_frame(W, H)
ref_np = synth.pad_plane(base); enc_np = synth.pad_plane(synth.moved_frame(base, 1))
enc_pin = torch.from_numpy(enc_np).pin_memory(); ref_pin = torch.from_numpy(ref_np).pin_memory()
jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
jl0 = eng.make_joblist(jobs)
e0, r0 = eng.upload_plane(enc_np), eng.upload_plane(ref_np)
want = jl0.run(e0, [r0])
K = 60
for D, P, ahead in ((1, 2, 1), (2, 3, 1), (3, 4, 1), (3, 5, 2), (2, 4, 2), (4, 5, 1), (4, 6, 2)):
    slots = [(eng.upload_plane(enc_np), eng.upload_plane(ref_np)) for _ in range(P)]
    lists = [eng.make_joblist(jobs) for _ in range(D)]
    outs = [torch.zeros(len(jobs) * 8, dtype=torch.uint8).pin_memory() for _ in range(D)]

    def frame(k):
        nxt = slots[(k + ahead) % P]
        eng.refill_plane_async(nxt[0], enc_pin.data_ptr())
        eng.refill_plane_async(nxt[1], ref_pin.data_ptr())
        st = slots[k % P]
        if D == 1:
            lists[0].run(st[0], [st[1]], fetch=False)
        else:
            lists[k % D].run_overlapped(st[0], [st[1]], 4 + k % D)
        lists[k % D].results_async(outs[k % D].data_ptr())

    for k in range(2 * P):
        frame(k)
    eng.sync()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(K):
            frame(k)
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / K * 1e3)
    ok = all(np.array_equal(o.numpy().view(J.RESULT_DTYPE), want) for o in outs)
    print("in flight %d, plane slots %d, refill %d ahead: %.4f ms per frame, fields equal %s" % (D, P, ahead, best, ok), flush=True)
    for l in lists: l.close()
    for s in slots:
        eng.release_plane(s[0]); eng.release_plane(s[1])
