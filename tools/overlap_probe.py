"""Developer aid: the 1080p exhaustive frame (42 900 PUs; argument `hex`: the HEX + subme 5 frame) with one frame at a time and
with two to four frames in flight (x265amd_joblist_run_overlapped, lists taking the engine's stream sets in turn); ms per frame."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
leg = sys.argv[1] if len(sys.argv) > 1 else "full"
eng = Engine(0)
fr = synth.make_sequence(W, H, 3, seed=5)
ids = [eng.upload_plane(f) for f in fr]
if leg == "full":
    jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
else:
    jobs = J.frame_jobs(W, H, sizes=(64, 16), method=J.HEX, subme=5, merange=64, qp=32)
lists = [eng.make_joblist(jobs) for _ in range(4)]
K = 60
for mode, NS in (("one at a time", 1), ("in flight", 2), ("in flight", 3), ("in flight", 4), ("in flight, one stream each", 1), ("in flight, one stream each", 2), ("in flight, one stream each", 3), ("in flight, one stream each", 4)):
    for rep in range(2):
        eng.sync()
        t0 = time.perf_counter()
        for k in range(K):
            if mode == "one at a time":
                lists[k & 1].run(ids[1 + (k & 1)], [ids[0]], fetch=False)
            else:
                lists[k % NS].run_overlapped(ids[1 + (k & 1)], [ids[0]], k % NS + (4 if 'one stream' in mode else 0))
        eng.sync()
        dt = (time.perf_counter() - t0) / K * 1e3
    print("%s %d %s: %.4f ms per frame" % (leg, NS, mode, dt), flush=True)
