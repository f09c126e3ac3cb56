#!/usr/bin/env python3
"""Developer aid: throughput of the HEX + subme 5 frame with 1, 2, 3 frames in flight (one engine =
one set of HIP streams per frame in flight; consecutive frames alternate between engines)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
dev = torch.device("cuda", 0)
base = synth.base_frame(W, H)
ref = synth.pad_plane(base); enc = synth.pad_plane(synth.moved_frame(base, 1))
et = torch.from_numpy(enc).to(dev); rt = torch.from_numpy(ref).to(dev)
leg = sys.argv[1] if len(sys.argv) > 1 else "hex"
if leg == "full":
    jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
else:
    jobs = J.frame_jobs(W, H, sizes=(64, 16), method=J.HEX, subme=5, merange=64, qp=32)
for depth in (1, 2, 3):
    engs = []
    dummies = []
    for _ in range(depth):
        engs.append(Engine(0))
        # (streams map to the process's hardware queues in creation order: X265AMD_PROBE_SHIFT dummy streams between two
        # engines shift the second engine's streams onto other queues than the first one's)
        for _ in range(int(os.environ.get("X265AMD_PROBE_SHIFT", "0"))):
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            st = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0      # (torch's streams come from a pool made at once)
            dummies.append(st)
    ctx = []
    for e in engs:
        eid = e.wrap_device_plane(et.data_ptr(), W, H, ref.shape[1], 1, 8)
        rid = e.wrap_device_plane(rt.data_ptr(), W, H, ref.shape[1], 1, 8)
        ctx.append((e, e.make_joblist(jobs), eid, rid))
    for k in range(8):
        e, jl, eid, rid = ctx[k % depth]
        jl.run(eid, [rid], fetch=False)
    for e in engs: e.sync()
    K = 60
    t0 = time.perf_counter()
    for k in range(K):
        e, jl, eid, rid = ctx[k % depth]
        jl.run(eid, [rid], fetch=False)
    for e in engs: e.sync()
    dt = time.perf_counter() - t0
    print("frames in flight %d: %.4f ms per frame" % (depth, dt / K * 1e3), flush=True)
    for (e, jl, _, _) in ctx:
        jl.close(); e.close()
