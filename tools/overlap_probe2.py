"""Developer aid: 1080p exhaustive frame, one at a time (kernel time, CRC) and four in flight on sets 4..7, for the library in X265AMD_LIB."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=5)
ids = [eng.upload_plane(f) for f in fr]
jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
lists = [eng.make_joblist(jobs) for _ in range(4)]
eng.set_timing(True)
kt = []
for k in range(12):
    lists[0].run(ids[1], [ids[0]], fetch=False); eng.sync()
    kt.append(eng.kernel_times()[0])
eng.set_timing(False)
K = 80
best = 1e9
for rep in range(4):
    eng.sync(); t0 = time.perf_counter()
    for k in range(K):
        lists[k & 3].run_overlapped(ids[1], [ids[0]], 4 + (k & 3))
    eng.sync(); best = min(best, (time.perf_counter() - t0) / K * 1e3)
res = lists[3].results()
print("%s: exhaustive kernels alone %.4f ms; four in flight %.4f ms per frame; crc %08x" % (
    os.path.basename(os.environ.get("X265AMD_LIB", "libx265amd.so")), float(np.median(kt[2:])), best, zlib.crc32(np.ascontiguousarray(res).tobytes())), flush=True)
