#!/usr/bin/env python3
"""Timing of an exhaustive +-16 search of EVERY partition made of 8x8 cells of every 64x64 CTU of a 4K 10-bit frame
(the 85 squares 64..8 plus the 124 symmetric / asymmetric rectangles of the 16-, 32- and 64-pixel CUs) in one job
list, with the SA8D and the SAD metric."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
eng = Engine(0)
eng.set_timing(True)
for (W, H, depth, mer) in ((3840, 2160, 10, 16), (1920, 1080, 8, 64)):
  fr = synth.make_sequence(W, H, 2, seed=5, bit_depth=depth)
  ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
  rx, ry, rw, rh = [], [], [], []
  for n2 in (16, 32, 64):
      q = n2 // 4
      shapes = [(0, 0, n2, n2 // 2), (0, n2 // 2, n2, n2 // 2), (0, 0, n2 // 2, n2), (n2 // 2, 0, n2 // 2, n2)]
      if n2 > 16:
          shapes += [(0, 0, n2, q), (0, q, n2, n2 - q), (0, 0, n2, n2 - q), (0, n2 - q, n2, q),
                     (0, 0, q, n2), (q, 0, n2 - q, n2), (0, 0, n2 - q, n2), (n2 - q, 0, q, n2)]
      cy, cx = np.mgrid[0:H - n2 + 1:n2, 0:W - n2 + 1:n2]
      for (ox, oy, w, h) in shapes:
          rx.append(cx.ravel() + ox); ry.append(cy.ravel() + oy)
          rw.append(np.full(cx.size, w)); rh.append(np.full(cx.size, h))
  rx, ry, rw, rh = (np.concatenate(a) for a in (rx, ry, rw, rh))
  for icost, name in (((J.SA8D, "SA8D"), (J.SAD, "SAD")) if depth == 10 else ((J.SAD, "SAD"),)):
      parts = [J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=0, merange=mer, qp=27, icost=icost)]
      for (w, h) in sorted(set(zip(rw.tolist(), rh.tolist()))):
          sel = (rw == w) & (rh == h)
          parts.append(J.make_jobs(W, H, rx[sel], ry[sel], w, h, method=J.FULL, subme=0, merange=mer, qp=27, icost=icost))
      jobs = np.concatenate(parts)
      jl = eng.make_joblist(jobs, bytes_per_px=1 if depth == 8 else 2)
      eng.eval_count(reset=True)
      best = (1e9, 1e9)
      for _ in range(2):
          jl.run(ids[1], [ids[0]], fetch=False)
          best = min(best, eng.kernel_times())
      ev = eng.eval_count(reset=True) / 2
      jl.close()
      print("%dx%d %d-bit FULL+-%d %-4s all partitions: jobs %7d  exhaustive kernels %8.3f ms  tail %8.3f ms  %.3g evals/s" %
            (W, H, depth, mer, name, len(jobs), best[0], best[1], ev / ((best[0] + best[1]) * 1e-3)), flush=True)
  for i in ids:
    eng.release_plane(i)
