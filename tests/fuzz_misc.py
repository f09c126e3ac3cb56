#!/usr/bin/env python3
"""Developer aid (not collected by pytest): randomised parity campaign of the entry points the other two campaigns
(fuzz_parity.py, fuzz_chain.py) do not reach - independent search jobs against half-resolution plane sets through
the batch call, bidirectional-refinement jobs (X265AMD_BIPRED), batched costs of motion-compensated blocks, and the
two entry points that take job arrays resident in device memory (the allocating one and the allocation-free one) -
against the CPU oracle.

usage: python tests/fuzz_misc.py SECONDS [SEED]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from fuzz_chain import make_frames
from oracle_ffi import Oracle
from x265_amd import jobs as J
from x265_amd.api import MC_PAIR_DTYPE, Engine

SHAPES = ((4, 4), (8, 4), (4, 8), (8, 8), (16, 8), (8, 16), (16, 16), (16, 12), (12, 16), (16, 4), (32, 16), (16, 32), (32, 32),
          (32, 24), (24, 32), (64, 32), (32, 64), (64, 64), (64, 48), (48, 64), (64, 16), (16, 64))


def fail(what, jobs, got, exp):
    bad = np.nonzero(got != exp)[0]
    print("MISMATCH %s: %d of %d differ; first job %s got %s exp %s" % (what, bad.size, len(got), jobs[bad[0]], got[bad[0]], exp[bad[0]]), flush=True)
    sys.exit(1)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    eng, oracle = Engine(0), Oracle()
    t0 = time.time()
    count = {"lowres": 0, "bipred": 0, "mc": 0, "device": 0}
    work = {"lowres": 0, "bipred": 0, "mc": 0, "device": 0}
    cases = 0
    while time.time() - t0 < budget:
        bits = int(rng.choice([8, 16]))
        depth = 8 if bits == 8 else 10
        kind = str(rng.choice(["motion", "motion", "noise", "flat"]))
        mode = str(rng.choice(["lowres", "bipred", "mc", "device"]))
        if mode == "lowres":
            w, h = 2 * int(rng.integers(40, 260)), 2 * int(rng.integers(40, 150))
            fr = make_frames(rng, w, h, depth, kind)
            src = [eng.upload_plane(f, bit_depth=depth) for f in fr]
            lr = [eng.lowres_create(s) for s in src]
            cpu = [oracle.lowres_init(f) for f in fr]
            g = cpu[0][0]
            parts = []
            for _ in range(int(rng.integers(4, 40))):
                pw, ph = SHAPES[int(rng.integers(0, 13))]
                if pw > g.width or ph > g.lines:
                    continue
                method = int(rng.integers(0, 5))
                if method == J.UMH and pw == 4 and ph == 4:
                    continue
                n = int(rng.integers(1, 6))
                px = rng.integers(0, (g.width - pw) // 4 + 1, n) * 4
                py = rng.integers(0, (g.lines - ph) // 4 + 1, n) * 4
                k = int(rng.integers(0, 5))
                parts.append(J.make_jobs(g.width, g.lines, px, py, pw, ph, method=method, subme=int(rng.integers(0, 8)),
                                         merange=int(rng.integers(1, 10)) if method == J.FULL else int(rng.integers(1, 40)),
                                         qp=int(rng.integers(8, 45)), qmvp=(int(rng.integers(-60, 60)), int(rng.integers(-60, 60))),
                                         mvc=rng.integers(-80, 80, (k, 2)) if k else None))
            if parts:
                jobs = np.concatenate(parts)
                got = eng.search(lr[0], [lr[1]], jobs)
                exp, _ = oracle.search(cpu[0][1][0], [cpu[1][1]], jobs, lowres=True, bit_depth=depth)
                if not np.array_equal(got, exp):
                    fail("half-resolution jobs %dx%d %d-bit %s" % (w, h, bits, kind), jobs, got, exp)
                work[mode] += len(jobs)
            for p in lr + src:
                eng.release_plane(p)
        elif mode == "bipred":
            w, h = 8 * int(rng.integers(10, 60)), 8 * int(rng.integers(10, 36))
            fr = make_frames(rng, w, h, depth, kind)
            ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
            parts = []
            for _ in range(int(rng.integers(4, 60))):
                pw, ph = SHAPES[int(rng.integers(0, len(SHAPES)))]
                if pw > w or ph > h:
                    continue
                px = int(rng.integers(0, (w - pw) // 4 + 1)) * 4
                py = int(rng.integers(0, (h - ph) // 4 + 1)) * 4
                j = J.make_jobs(w, h, [px], [py], pw, ph, method=J.BIPRED, subme=0, merange=4, qp=int(rng.integers(10, 45)),
                                qmvp=(int(rng.integers(-80, 80)), int(rng.integers(-80, 80))))
                cx, cy = int(rng.integers(-20, 20)), int(rng.integers(-20, 20))
                r = int(rng.choice([4, 4, 2, 7, 1]))
                j["mvmin_x"], j["mvmax_x"], j["mvmin_y"], j["mvmax_y"] = cx - r, cx + r, cy - r, cy + r
                if rng.random() < 0.2:
                    j["mvmax_x"] = cx - r + int(rng.integers(0, 3))
                parts.append(j)
            if parts:
                jobs = np.concatenate(parts)
                exp, _ = oracle.search(fr[0], [fr[1]], jobs, bit_depth=depth)
                got = eng.search(ids[0], [ids[1]], jobs)
                if not np.array_equal(got, exp):
                    fail("bidirectional jobs %dx%d %d-bit %s (batch call)" % (w, h, bits, kind), jobs, got, exp)
                jl = eng.make_joblist(jobs, bytes_per_px=bits // 8)
                got = jl.run(ids[0], [ids[1]])
                jl.close()
                if not np.array_equal(got, exp):
                    fail("bidirectional jobs %dx%d %d-bit %s (planned list)" % (w, h, bits, kind), jobs, got, exp)
                work[mode] += len(jobs)
            for p in ids:
                eng.release_plane(p)
        elif mode == "device":
            import torch
            w, h = 8 * int(rng.integers(10, 60)), 8 * int(rng.integers(10, 36))
            fr = make_frames(rng, w, h, depth, kind)
            ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
            parts = []
            for _ in range(int(rng.integers(4, 60))):
                pw, ph = SHAPES[int(rng.integers(0, len(SHAPES)))]
                method = int(rng.integers(0, 5))
                if pw > w or ph > h or (method == J.UMH and pw == 4 and ph == 4):
                    continue
                px = int(rng.integers(0, (w - pw) // 4 + 1)) * 4
                py = int(rng.integers(0, (h - ph) // 4 + 1)) * 4
                k = int(rng.integers(0, 5))
                parts.append(J.make_jobs(w, h, [px], [py], pw, ph, method=method, subme=int(rng.integers(0, 8)),
                                         merange=int(rng.integers(1, 24)) if method == J.FULL else int(rng.integers(1, 64)),
                                         qp=int(rng.integers(8, 45)), qmvp=(int(rng.integers(-100, 100)), int(rng.integers(-100, 100))),
                                         mvc=rng.integers(-200, 200, (k, 2)) if k else None, icost=int(rng.choice([0, 0, 0, 1, 2])),
                                         ref_idx=int(rng.integers(0, 2))))
            if parts:
                jobs = np.concatenate(parts)
                exp, _ = oracle.search(fr[0], [fr[1], fr[0]], jobs, bit_depth=depth)
                dev = torch.device("cuda", 0)
                dj = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
                for which in ("allocating", "allocation-free"):
                    do = torch.zeros(len(jobs) * 8, dtype=torch.uint8, device=dev)
                    if which == "allocating":
                        eng.search_device(ids[0], [ids[1], ids[0]], dj.data_ptr(), len(jobs), do.data_ptr())
                    else:
                        eng.search_device_noalloc(ids[0], [ids[1], ids[0]], dj.data_ptr(), len(jobs), do.data_ptr())
                    eng.sync()
                    got = do.cpu().numpy().view(J.RESULT_DTYPE)
                    if not np.array_equal(got, exp):
                        fail("device-resident jobs %dx%d %d-bit %s (%s entry point)" % (w, h, bits, kind, which), jobs, got, exp)
                work[mode] += len(jobs)
            for p in ids:
                eng.release_plane(p)
        else:
            w, h = 8 * int(rng.integers(10, 50)), 8 * int(rng.integers(10, 30))
            fr = make_frames(rng, w, h, depth, kind)
            ids = [eng.upload_plane(f, bit_depth=depth) for f in fr]
            for _ in range(3):
                pw, ph = SHAPES[int(rng.integers(0, len(SHAPES)))]
                if pw > w or ph > h:
                    continue
                n = int(rng.integers(1, 30))
                p = np.zeros(n, dtype=MC_PAIR_DTYPE)
                p["enc_plane"], p["ref_plane"] = ids[0], ids[1]
                p["pu_x"] = rng.integers(0, (w - pw) // 4 + 1, n) * 4
                p["pu_y"] = rng.integers(0, (h - ph) // 4 + 1, n) * 4
                p["qmv_x"] = rng.integers(-280, 280, n)
                p["qmv_y"] = rng.integers(-280, 280, n)
                ck = int(rng.integers(0, 3))
                got = eng.mc_cost_batch(ck, pw, ph, p)
                exp = [oracle.mc_cost(fr[0], fr[1], int(q["pu_x"]), int(q["pu_y"]), pw, ph, int(q["qmv_x"]), int(q["qmv_y"]), ck, bit_depth=depth)
                       for q in p]
                if got.tolist() != exp:
                    print("MISMATCH mc_cost %dx%d PU %dx%d kind %d %d-bit %s" % (w, h, pw, ph, ck, bits, kind), flush=True)
                    sys.exit(1)
                work[mode] += n
            for p in ids:
                eng.release_plane(p)
        count[mode] += 1
        cases += 1
        if cases % 50 == 0:
            print("case %d ok (%s, %d-bit, %s; %d s)" % (cases, mode, bits, kind, time.time() - t0), flush=True)
    print("campaign: %d half-resolution batches (%d jobs), %d bidirectional batches (%d jobs, each through both entry paths), "
          "%d motion-compensated cost cases (%d costs), %d device-resident batches (%d jobs, each through the allocating and the "
          "allocation-free entry point), all bit-exact against the oracle (seed %d, %.0f s)"
          % (count["lowres"], work["lowres"], count["bipred"], work["bipred"], count["mc"], work["mc"], count["device"], work["device"],
             seed, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
