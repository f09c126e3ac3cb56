"""GPU parity on the larger BASELINE configurations (4K, several reference frames, 10-bit pixels,
every HEVC PU shape), on samples of PUs small enough for the CPU oracle to finish in seconds."""
import numpy as np
import pytest

from x265_amd import jobs as J
from x265_amd import synth

pytestmark = pytest.mark.gpu

W4K, H4K = 3840, 2160


@pytest.fixture(scope="module")
def eng():
    from x265_amd.api import Engine
    e = Engine(0)
    yield e
    e.close()


def _sampled(pic_w, pic_h, w, h, count, rng, **kw):
    """`count` PUs of size w x h at random 4-aligned positions (picture edges included)."""
    xs = rng.integers(0, (pic_w - w) // 4 + 1, count) * 4
    ys = rng.integers(0, (pic_h - h) // 4 + 1, count) * 4
    xs[:2] = (0, pic_w - w)            # the picture corners: clipped ranges
    ys[:2] = (0, pic_h - h)
    return J.make_jobs(pic_w, pic_h, xs, ys, w, h, **kw)


def test_4k_three_references_multi_candidate_search(eng, oracle):
    """4K 8-bit, 3 reference frames, every PU searched in each reference: HEX / STAR / UMH
    (sad_x3 / sad_x4 steps) with sub-pel refinement, and FULL +-64."""
    rng = np.random.default_rng(2024)
    fr = synth.make_sequence(W4K, H4K, 4, seed=77)
    ids = [eng.upload_plane(f) for f in fr]
    parts = []
    for ref in range(3):
        for method, subme, n in ((J.HEX, 5, 60), (J.STAR, 2, 30), (J.UMH, 3, 30), (J.DIA, 7, 30)):
            for (w, h) in ((64, 64), (32, 32), (16, 16), (8, 8)):
                parts.append(_sampled(W4K, H4K, w, h, n // 2, rng, method=method, subme=subme, merange=64, qp=32,
                                      qmvp=(int(rng.integers(-60, 60)), int(rng.integers(-60, 60))),
                                      mvc=rng.integers(-80, 80, (2, 2)), ref_idx=ref))
        for (w, h, n) in ((64, 64, 4), (32, 32, 6), (16, 16, 10), (8, 8, 16)):
            parts.append(_sampled(W4K, H4K, w, h, n, rng, method=J.FULL, subme=0, merange=64, qp=32, ref_idx=ref))
    jobs = np.concatenate(parts)
    exp, evals = oracle.search(fr[0], fr[1:], jobs, threads=8)
    eng.eval_count(reset=True)
    jl = eng.make_joblist(jobs, n_refs=3)
    got = jl.run(ids[0], ids[1:])
    jl.close()
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:5]
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


def test_bench_job_list_sampled_against_the_oracle(eng, oracle):
    """The exact job list bench.py's headline runs (1080p, exhaustive +-64 SAD search of every square PU 64/32/16/8:
    42 900 PUs, the bottom CTU row cut at 56 of 64 lines) through a planned list, as the bench does; more than 1000
    PUs compared with the oracle: a random sample of every level plus every PU of the partial bottom cell row's
    rightmost and leftmost CTUs (clipped ranges in both directions)."""
    import bench
    _, full_jobs = bench.build_job_lists()
    assert len(full_jobs) == 42900
    base = synth.base_frame(bench.W, bench.H)
    ref = synth.pad_plane(base)
    enc = synth.pad_plane(synth.moved_frame(base, 1))
    ids = [eng.upload_plane(enc), eng.upload_plane(ref)]
    jl = eng.make_joblist(full_jobs)
    got = jl.run(ids[0], [ids[1]])
    jl.close()
    rng = np.random.default_rng(2025)
    pick = []
    for size, count in ((64, 10), (32, 40), (16, 250), (8, 700)):
        idx = np.nonzero(full_jobs["w"] == size)[0]
        pick.append(rng.choice(idx, count, replace=False))
    bottom = np.nonzero((full_jobs["pu_y"] >= 1024) & ((full_jobs["pu_x"] < 64) | (full_jobs["pu_x"] >= bench.W - 64)))[0]
    assert len(bottom) > 80
    sel = np.unique(np.concatenate(pick + [bottom]))
    assert len(sel) >= 1000
    exp, _ = oracle.search(enc, [ref], np.ascontiguousarray(full_jobs[sel]), threads=8)
    bad = np.nonzero(got[sel] != exp)[0]
    assert bad.size == 0, (len(bad), full_jobs[sel][bad[:3]], got[sel][bad[:3]], exp[bad[:3]])
    for i in ids:
        eng.release_plane(i)


def test_two_frames_in_flight_equal_one_at_a_time(eng, oracle):
    """x265amd_joblist_run_overlapped: the headline's job list for two different current frames, one list per frame,
    queued alternately on the engine's two stream sets (six runs back to back, no host wait between them).  Both
    fields equal the fields of the same lists run one at a time, and a sample of each equals the oracle."""
    import bench
    _, full_jobs = bench.build_job_lists()
    base = synth.base_frame(bench.W, bench.H)
    ref = synth.pad_plane(base)
    encs = [synth.pad_plane(synth.moved_frame(base, k)) for k in (1, 2)]
    rid = eng.upload_plane(ref)
    eids = [eng.upload_plane(e) for e in encs]
    lists = [eng.make_joblist(full_jobs) for _ in range(2)]
    alone = [lists[k].run(eids[k], [rid]) for k in range(2)]
    assert (alone[0] != alone[1]).any()
    eng.sync()
    for rep in range(3):
        for k in range(2):
            lists[k].run_overlapped(eids[k], [rid], k)
    got = [lists[k].results() for k in range(2)]
    # ... and with the sets exchanged (a list's next run waits for its last one on the other set)
    for k in range(2):
        lists[k].run_overlapped(eids[k], [rid], 1 - k)
    eng.sync()
    got2 = [lists[k].results() for k in range(2)]
    for k in range(2):
        assert (got[k] == alone[k]).all() and (got2[k] == alone[k]).all()
    sel = np.arange(0, len(full_jobs), 173)
    for k in range(2):
        exp, _ = oracle.search(encs[k], [ref], np.ascontiguousarray(full_jobs[sel]), threads=8)
        assert (got[k][sel] == exp).all()
    for jl in lists:
        jl.close()
    for i in eids + [rid]:
        eng.release_plane(i)


# the PU shapes of one 64x64 CTU: 2Nx2N, 2NxN, Nx2N at depths 0-3, AMP at depths 0-2, 8x4 / 4x8 at depth 3
HEVC_PU_SHAPES = [(64, 64), (64, 32), (32, 64), (64, 16), (64, 48), (16, 64), (48, 64),
                  (32, 32), (32, 16), (16, 32), (32, 8), (32, 24), (8, 32), (24, 32),
                  (16, 16), (16, 8), (8, 16), (16, 4), (16, 12), (4, 16), (12, 16),
                  (8, 8), (8, 4), (4, 8)]


def test_4k_10bit_all_pu_shapes_sa8d(eng, oracle):
    """4K, 10-bit samples in 16-bit pixels, all 24 HEVC PU shapes, integer-pel search with the
    SA8D cost (FULL +-16 and HEX), plus the reference's own metric (SAD) for the same PUs."""
    rng = np.random.default_rng(7)
    fr = synth.make_sequence(W4K, H4K, 2, seed=5, bit_depth=10)
    ids = [eng.upload_plane(f, bit_depth=10) for f in fr]
    parts = []
    for (w, h) in HEVC_PU_SHAPES:
        parts.append(_sampled(W4K, H4K, w, h, 3, rng, method=J.FULL, subme=0, merange=16, qp=27, icost=J.SA8D))
        parts.append(_sampled(W4K, H4K, w, h, 6, rng, method=J.HEX, subme=0, merange=64, qp=27, icost=J.SA8D,
                              qmvp=(int(rng.integers(-8, 8)) * 4, int(rng.integers(-8, 8)) * 4)))
        parts.append(_sampled(W4K, H4K, w, h, 3, rng, method=J.FULL, subme=0, merange=16, qp=27))
    jobs = np.concatenate(parts)
    exp, evals = oracle.search(fr[0], [fr[1]], jobs, threads=8, bit_depth=10)
    eng.eval_count(reset=True)
    got = eng.search(ids[0], [ids[1]], jobs)
    assert np.array_equal(got, exp), np.nonzero(got != exp)[0][:5]
    assert eng.eval_count() == evals
    for i in ids:
        eng.release_plane(i)


def _single_rank_group():
    """a process group of ONE rank on this GPU, so that the collective of the multi-GPU path really runs
    (RCCL when it initialises, else gloo); returns the backend name"""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist.get_backend()
    import os
    port = 29700 + (os.getpid() % 200)
    last = None
    for backend in ("nccl", "gloo"):
        try:
            kw = {"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}
            dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, **kw)
            return backend
        except Exception as e:      # pragma: no cover - depends on the box
            last = e
            port += 1
    raise last


@pytest.mark.parametrize("in_flight", [1, 2, 4])
def test_4k_lookahead_batch_through_plane_ring(oracle, in_flight):
    """(in_flight > 1: the frames take the engine's stream sets in turn - two-stream sets 0 / 1 or single-stream sets 4 .. 7 -
    and every hand-over of the ring goes to the set of the frame it belongs to.)
    BASELINE's 4K 8-frame lookahead batch on the rank that would own all of it: the frames of the
    batch are searched one after the other, each against the reference plane that arrives through
    PlaneRing (broadcast into one of two device buffers wrapped with x265amd_plane_wrap_device, the
    next plane's broadcast queued beside the current frame's searches, all hand-overs stream events).
    Results of every frame are compared with the oracle on a sample of PUs (all methods + FULL)."""
    import torch
    import torch.distributed as dist
    from x265_amd import dist as xd
    from x265_amd.api import Engine
    backend = _single_rank_group()
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    NF = 8
    rng = np.random.default_rng(31)
    fr = synth.make_sequence(W4K, H4K, NF + 2, seed=99)          # frames 0, 1: the two references; 2..9: the batch
    stride = fr[0].shape[1]
    parts = []
    for method, subme, n in ((J.HEX, 5, 40), (J.STAR, 2, 12), (J.UMH, 3, 12), (J.DIA, 7, 12)):
        for (w, h) in ((64, 64), (16, 16), (8, 8)):
            parts.append(_sampled(W4K, H4K, w, h, n // 2, rng, method=method, subme=subme, merange=64, qp=32,
                                  qmvp=(int(rng.integers(-40, 40)), int(rng.integers(-40, 40)))))
    for (w, h, n) in ((64, 64, 2), (32, 32, 3), (16, 16, 6), (8, 8, 10)):
        parts.append(_sampled(W4K, H4K, w, h, n, rng, method=J.FULL, subme=0, merange=64, qp=32))
    jobs = np.concatenate(parts)
    jls = [eng.make_joblist(jobs) for _ in range(NF)]          # one list (and result buffer) per frame of the batch
    # the owner's copies of the reference planes (host side) and the two-buffer ring on the device
    host_refs = [torch.from_numpy(fr[0]).pin_memory(), torch.from_numpy(fr[1]).pin_memory()]
    bufs = [torch.zeros(fr[0].shape, dtype=torch.uint8, device=dev) for _ in range(2)]
    buf_ids = [eng.wrap_device_plane(b.data_ptr(), W4K, H4K, stride, 1, 8) for b in bufs]
    encs = [torch.from_numpy(f).to(dev) for f in fr[2:]]
    enc_ids = [eng.wrap_device_plane(t.data_ptr(), W4K, H4K, stride, 1, 8) for t in encs]
    ring = xd.PlaneRing(bufs, engine=eng, src=0, overlap=True)

    def publish(k):
        # the owner writes reference k (the frames alternate between the two references) into the ring
        # buffer it is about to broadcast, once the searches still reading that buffer are done
        ring.publish(k, host_refs[k & 1])

    sets = {1: (0,), 2: (0, 1), 4: (4, 5, 6, 7)}[in_flight]
    publish(0)
    for k in range(NF):                     # nothing in this loop waits on the host
        st = sets[k % len(sets)]
        b = ring.acquire(k, st)
        assert b == (k & 1)
        if in_flight == 1:
            jls[k].run(enc_ids[k], [buf_ids[b]], fetch=False)
        else:
            jls[k].run_overlapped(enc_ids[k], [buf_ids[b]], st)
        if k + 1 < NF:
            publish(k + 1)
            ring.prefetch(k + 1)
        ring.release(k, st)
    eng.sync()
    for k in range(NF):
        got = jls[k].results()
        exp, _ = oracle.search(fr[2 + k], [fr[k & 1]], jobs, threads=8)
        assert np.array_equal(got, exp), (backend, k, np.nonzero(got != exp)[0][:5])
        jls[k].close()
    eng.close()
    dist.destroy_process_group()


def test_planes_refilled_and_fields_fetched_asynchronously(eng, oracle):
    """Pipelined frames: two sets of planes refilled from page-locked host memory beside the searches
    (x265amd_plane_refill_async) and every frame's field fetched behind its search into page-locked memory
    (x265amd_joblist_results_async), the host waiting once at the end - every frame's field must be the oracle's for
    THAT frame's pictures (a refill that started too early or a search that did not wait for one would mix frames)."""
    import torch
    W, H, NF = 416, 240, 6
    fr = synth.make_sequence(W, H, NF + 1, seed=41)
    pins = [torch.from_numpy(f).pin_memory() for f in fr]
    jobs = np.concatenate([J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=1, merange=24, qp=30),
                           J.frame_jobs(W, H, sizes=(16, 8), method=J.HEX, subme=5, merange=32, qp=30)])
    sets = [{"enc": eng.upload_plane(fr[1]), "ref": eng.upload_plane(fr[0])}, {"enc": eng.create_plane(W, H), "ref": eng.create_plane(W, H)}]
    jls = [eng.make_joblist(jobs) for _ in range(NF)]
    outs = [torch.zeros(len(jobs) * 8, dtype=torch.uint8).pin_memory() for _ in range(NF)]
    for k in range(NF):                     # frame k: picture k + 1 against picture k; nothing in this loop waits
        if k + 1 < NF:
            nxt = sets[(k + 1) & 1]
            eng.refill_plane_async(nxt["enc"], pins[k + 2].data_ptr())
            eng.refill_plane_async(nxt["ref"], pins[k + 1].data_ptr())
        st = sets[k & 1]
        jls[k].run(st["enc"], [st["ref"]], fetch=False)
        jls[k].results_async(outs[k].data_ptr())
    eng.sync()
    for k in range(NF):
        exp, _ = oracle.search(fr[k + 1], [fr[k]], jobs, threads=8)
        got = outs[k].numpy().view(J.RESULT_DTYPE)
        assert np.array_equal(got, exp), (k, np.nonzero(got != exp)[0][:5])
        jls[k].close()
    for st in sets:
        eng.release_plane(st["enc"])
        eng.release_plane(st["ref"])


@pytest.mark.parametrize("sets", [(4, 5), (0, 1), (5, 6, 7)])
def test_refills_and_downloads_with_frames_in_flight(eng, oracle, sets):
    """The same pipeline with several frames in flight: frame k runs on stream set sets[k mod D] (one list per set, reused
    every D frames), its planes live in one of D + 2 plane slots refilled two frames ahead, its field comes down on its own
    set.  A refill waits for the readers of its slot on EVERY set and a run for the refill of its planes; every field must be
    the oracle's for that frame's pictures."""
    import torch
    W, H, NF = 416, 240, 12
    D, P, AHEAD = len(sets), len(sets) + 2, 2
    fr = synth.make_sequence(W, H, NF + 1, seed=43)
    pins = [torch.from_numpy(f).pin_memory() for f in fr]
    jobs = np.concatenate([J.frame_jobs(W, H, sizes=(64, 32, 16, 8), method=J.FULL, subme=1, merange=24, qp=30),
                           J.frame_jobs(W, H, sizes=(16, 8), method=J.HEX, subme=5, merange=32, qp=30)])
    slots = [{"enc": eng.create_plane(W, H), "ref": eng.create_plane(W, H)} for _ in range(P)]
    lists = [eng.make_joblist(jobs) for _ in range(D)]
    outs = [torch.zeros(len(jobs) * 8, dtype=torch.uint8).pin_memory() for _ in range(NF)]

    def refill(k):                          # the pictures of frame k: picture k + 1 against picture k
        eng.refill_plane_async(slots[k % P]["enc"], pins[k + 1].data_ptr())
        eng.refill_plane_async(slots[k % P]["ref"], pins[k].data_ptr())

    for k in range(AHEAD):
        refill(k)
    for k in range(NF):                     # nothing in this loop waits on the host
        if k + AHEAD < NF:
            refill(k + AHEAD)
        st = slots[k % P]
        lists[k % D].run_overlapped(st["enc"], [st["ref"]], sets[k % D])
        lists[k % D].results_async(outs[k].data_ptr())
    eng.sync()
    for k in range(NF):
        exp, _ = oracle.search(fr[k + 1], [fr[k]], jobs, threads=8)
        got = outs[k].numpy().view(J.RESULT_DTYPE)
        assert np.array_equal(got, exp), (k, np.nonzero(got != exp)[0][:5])
    for jl in lists:
        jl.close()
    for st in slots:
        eng.release_plane(st["enc"])
        eng.release_plane(st["ref"])
