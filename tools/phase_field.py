"""Developer aid: phase clocks of the predictor-chained field kernel (library built with -DXA_PROFILE, named by X265AMD_LIB):
one 1080p field per run; x265amd_eval_count prints the clocks summed over the PUs of every 8th row to stderr."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
W, H = 1920, 1080
eng = Engine(0)
fr = synth.make_sequence(W, H, 2, seed=91)
ids = [eng.upload_plane(f) for f in fr]
for pu, method, subme in ((8, J.HEX, 3), (16, J.HEX, 3), (32, J.HEX, 3)):
    eng.field_me(ids[1], ids[0], pu, pu, 64, method, subme, 32, fetch=False)
    eng.sync()
    eng.eval_count(reset=True)
    eng.field_me(ids[1], ids[0], pu, pu, 64, method, subme, 32, fetch=False)
    eng.sync()
    print("---- the measured run:", file=sys.stderr, flush=True)
    n_rows = len(range(0, H // pu, 8))
    print("field %dx%d method %d subme %d: %d sampled rows x %d PUs = %d PUs; divide the clocks below by that" %
          (pu, pu, method, subme, n_rows, W // pu, n_rows * (W // pu)), file=sys.stderr, flush=True)
    eng.eval_count(reset=True)
