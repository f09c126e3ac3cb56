#!/usr/bin/env python3
"""Developer aid: a long randomised parity campaign of the search entry points against the CPU restatement
(oracle/libme_oracle.so, the checker the tests use).  Random frame sizes, contents, bit depths, PU shapes, methods,
sub-pel levels, predictors, ranges and cost kinds; every batch goes through the host-batch call, a planned job list, and two
planned lists in flight on two of the engine's stream sets.

    python tests/fuzz_parity.py [seconds] [seed]      (default 240 s)

Prints one line per batch and a summary; exits non-zero at the first mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from x265_amd import jobs as J, synth
from x265_amd.api import Engine
from oracle_ffi import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
rng = np.random.default_rng(seed)
eng, orc = Engine(0), Oracle()
t0 = time.time()
total = batches = 0
while time.time() - t0 < budget:
    bits, depth = [(8, 8), (16, 10), (16, 8), (16, 12)][int(rng.integers(0, 4))]
    W = int(rng.integers(5, 40)) * 16
    H = int(rng.integers(4, 24)) * 16
    kind = ["motion", "noise", "flat"][int(rng.integers(0, 3))]
    bd = depth if bits == 16 else 8
    if kind == "flat":     # every candidate ties: only the MV cost and the scan order decide
        flat = np.full((H, W), int(rng.integers(0, 1 << bd)), dtype=np.uint16 if bits == 16 else np.uint8)
        fr = [synth.pad_plane(flat, bd)] * 3
    else:
        fr = synth.make_sequence(W, H, 3, seed=int(rng.integers(1, 1 << 30)), bit_depth=bd, kind=kind)
    if bits == 16:
        fr = [np.ascontiguousarray(f.astype(np.uint16)) for f in fr]      # (depth 8 in 16-bit planes included)
    parts = []
    mode = int(rng.integers(0, 3))
    if mode == 0:          # independent random jobs
        for _ in range(int(rng.integers(100, 600))):
            w, h = int(rng.choice(J.DIMS)), int(rng.choice(J.DIMS))
            if w > W or h > H:
                continue
            method = int(rng.integers(0, 5))
            if method == J.UMH and w == 4 and h == 4:
                continue
            merange = int(rng.integers(1, 24)) if method == J.FULL else int(rng.integers(1, 80))
            px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
            py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
            k = int(rng.integers(0, 5))
            mvc = rng.integers(-320, 320, (k, 2)) if k else None
            icost = int(rng.choice([0, 0, 0, 1, 2]))
            parts.append(J.make_jobs(W, H, [px], [py], w, h, method=method, subme=int(rng.integers(0, 8)), merange=merange,
                                     qp=int(rng.integers(0, 52)), qmvp=(int(rng.integers(-260, 260)), int(rng.integers(-260, 260))),
                                     mvc=mvc, icost=icost, ref_idx=int(rng.integers(0, 2))))
    elif mode == 1:        # whole frames of partitions sharing predictor and QP (the shared-pass exhaustive kernels)
        sizes = [s for s in (64, 32, 16, 8) if rng.random() < 0.7 and s <= min(W, H)] or [16]
        icost = int(rng.choice([0, 0, 2]))
        qmvp = (int(rng.integers(-60, 60)), int(rng.integers(-60, 60)))
        parts.append(J.frame_jobs(W, H, sizes=tuple(sizes), method=J.FULL, subme=int(rng.integers(0, 4)), merange=int(rng.integers(4, 40)),
                                  qp=int(rng.integers(10, 45)), qmvp=qmvp, icost=icost))
        for (w, h) in ((64, 32), (32, 64), (32, 16), (16, 32), (16, 8), (8, 16), (64, 16), (16, 64), (64, 48), (32, 24), (8, 32)):
            if rng.random() < 0.4 and w <= W and h <= H:
                x, y = J.square_pu_grid(W, H, max(w, h))
                parts.append(J.make_jobs(W, H, x, y, w, h, method=J.FULL, subme=int(rng.integers(0, 3)), merange=int(rng.integers(4, 40)),
                                         qp=int(rng.integers(10, 45)), qmvp=qmvp, icost=icost))
    else:                  # whole frames of pattern searches, mixed sizes (all job classes side by side)
        for s in (64, 32, 16, 8):
            if s <= min(W, H) and rng.random() < 0.8:
                parts.append(J.frame_jobs(W, H, sizes=(s,), method=int(rng.integers(0, 4)), subme=int(rng.integers(0, 8)),
                                          merange=int(rng.integers(8, 70)), qp=int(rng.integers(10, 45)),
                                          qmvp=(int(rng.integers(-40, 40)), int(rng.integers(-40, 40)))))
    if not parts:
        continue
    jobs = np.concatenate(parts)
    exp, evals = orc.search(fr[0], [fr[1], fr[2]], jobs, threads=16, bit_depth=depth)
    ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
    got = eng.search(ids[0], [ids[1], ids[2]], jobs)
    jl = eng.make_joblist(jobs, n_refs=2, bytes_per_px=fr[0].itemsize)
    got2 = jl.run(ids[0], [ids[1], ids[2]])
    # ... and with frames in flight: the same list and a second one queued on a random pair of the engine's stream sets
    jl_b = eng.make_joblist(jobs, n_refs=2, bytes_per_px=fr[0].itemsize)
    sa, sb = [(0, 1), (4, 5), (6, 7), (2, 3), (5, 0)][int(rng.integers(0, 5))]
    for rep in range(2):
        jl.run_overlapped(ids[0], [ids[1], ids[2]], sa)
        jl_b.run_overlapped(ids[0], [ids[1], ids[2]], sb)
    got3, got4 = jl.results(), jl_b.results()
    jl.close()
    jl_b.close()
    for i in ids:
        eng.release_plane(i)
    for name, g in (("host batch", got), ("job list", got2), ("job list, stream set %d" % sa, got3), ("second job list, stream set %d" % sb, got4)):
        bad = np.nonzero(g != exp)[0]
        if bad.size:
            print("MISMATCH (%s): mode %d bits %d depth %d %dx%d %s: %d of %d jobs; first: %s got %s want %s"
                  % (name, mode, bits, depth, W, H, kind, bad.size, len(jobs), jobs[bad[0]], g[bad[0]], exp[bad[0]]), flush=True)
            sys.exit(1)
    total += len(jobs); batches += 1
    print("batch %3d ok: mode %d %2d-bit depth %2d %4dx%-4d %-6s %6d jobs  (%.0f s)" % (batches, mode, bits, depth, W, H, kind, len(jobs), time.time() - t0), flush=True)
print("fuzz campaign: %d batches, %d jobs, all bit-exact against the oracle (seed %d, %.0f s)" % (batches, total, seed, time.time() - t0))
