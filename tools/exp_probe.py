"""Developer aid: the 1080p FULL +-64 frame (all four square PU levels) through a planned list with the library named by
X265AMD_LIB; prints the median HIP-event times of the exhaustive kernels and of the search kernels behind them, and a CRC of the
results (so that variants meant to be equivalent can be compared)."""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine

W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=5)
ids = [eng.upload_plane(f) for f in fr]
eng.set_timing(True)
jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
jl = eng.make_joblist(jobs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
full, tail = [], []
for k in range(n + 3):
    jl.run(ids[1], [ids[0]], fetch=False)
    eng.sync()
    if k >= 3:
        a, b = eng.kernel_times()
        full.append(a); tail.append(b)
import time
eng.set_timing(False)
wall = []
for rep in range(5):
    eng.sync()
    t0 = time.perf_counter()
    for k in range(n):
        jl.run(ids[1], [ids[0]], fetch=False)
    eng.sync()
    wall.append((time.perf_counter() - t0) / n * 1e3)
res = jl.run(ids[1], [ids[0]])
crc = zlib.crc32(np.ascontiguousarray(res).tobytes())
print("%s chunks=%s flags=%s: jobs %d exhaustive %.4f ms (min %.4f) search %.4f ms; wall per run %.4f ms (min of 5 x %d); crc %08x" % (
      os.path.basename(os.environ.get("X265AMD_LIB", "libx265amd.so")), os.environ.get("X265AMD_CHUNKS", "-"), os.environ.get("X265AMD_FLAGS", "-"), len(jobs),
      float(np.median(full)), min(full), float(np.median(tail)), min(wall), n, crc), flush=True)
