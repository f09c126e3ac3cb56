"""Developer aid: phase clocks of the shared-window SAD pass (library built with -DXA_PROFILE, X265AMD_LIB pointing at it) for a
frame of `rows` rows of 64x64 cells, 1920 wide: one row = 270 work items (about one per CU), 17 rows = the 1080p frame."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
eng = Engine(0)
for rows in [int(a) for a in sys.argv[1:]] or [1, 17]:
    W, H = 1920, 64 * rows
    fr = synth.make_sequence(W, H, 2, seed=5)
    ids = [eng.upload_plane(f) for f in fr]
    eng.set_timing(True)
    jobs = J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=64, qp=32)
    jl = eng.make_joblist(jobs)
    for _ in range(3):
        jl.run(ids[1], [ids[0]], fetch=False)
    eng.sync()
    eng.eval_count(reset=True)
    jl.run(ids[1], [ids[0]], fetch=False)
    eng.sync()
    print("rows %d: exhaustive kernels %.4f ms" % (rows, eng.kernel_times()[0]), flush=True)
    sys.stderr.flush()
    eng.eval_count(reset=True)
    jl.close()
