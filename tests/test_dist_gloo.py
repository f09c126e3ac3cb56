"""world_size-2 test of the frame-sharded multi-GPU path on CPU (gloo): frame -> rank dealing,
reference-plane broadcast and result gather.  The search itself is stood in for by the CPU
oracle here (no GPU in this test); what is checked is that the sharded run reproduces the
single-process run frame for frame."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H, NFRAMES = 192, 128, 5


def _inputs():
    from x265_amd import jobs as J, synth
    frames = synth.make_sequence(W, H, NFRAMES + 1, seed=11)
    jobs = J.frame_jobs(W, H, sizes=(32, 16), method=J.HEX, subme=2, merange=16, qp=30)
    return frames, jobs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle_ffi import Oracle
    from x265_amd import dist as xd
    frames, jobs = _inputs()
    # rank 0 owns the reference plane (frame 0); the others start with zeros and receive it
    ref = torch.from_numpy(frames[0].copy()) if rank == 0 else torch.zeros(frames[0].shape, dtype=torch.uint8)
    xd.broadcast_plane(ref, src=0)
    assert np.array_equal(ref.numpy(), frames[0])
    # a ring of two buffers, filled by the owner through publish(); with overlap plane k + 1 is broadcast
    # while plane k is in use, without it (the default) the broadcast is queued by acquire(k + 1)
    for overlap in (True, False):
        bufs = [torch.zeros(frames[0].shape, dtype=torch.uint8) for _ in range(2)]
        ring = xd.PlaneRing(bufs, src=0, overlap=overlap) if overlap else xd.PlaneRing(bufs, src=0)
        assert ring.overlap == overlap
        for k in range(4):
            if ring.ready < k:
                assert ring.publish(k, torch.from_numpy(frames[k])) == k % 2     # a no-op on ranks other than src
            b = ring.acquire(k)
            assert b == k % 2 and np.array_equal(bufs[b].numpy(), frames[k]), k
            ring.wait_free(k + 1)                                                 # nothing to wait for on the CPU
            ring.publish(k + 1, torch.from_numpy(frames[k + 1]))
            ring.prefetch(k + 1)
            assert ring.ready == (k + 1 if overlap else k)
            assert np.array_equal(bufs[b].numpy(), frames[k]), k                  # the buffer in use is untouched
            if rank != 0 and not overlap:
                assert not np.array_equal(bufs[1 - b].numpy(), frames[k + 1])     # not broadcast yet
            ring.release(k)
    # row-wise publishing: the owner hands out the padded plane 48 rows at a time
    rows = torch.from_numpy(frames[2].copy()) if rank == 0 else torch.zeros(frames[2].shape, dtype=torch.uint8)
    for r0 in range(0, rows.shape[0], 48):
        n = min(48, rows.shape[0] - r0)
        xd.broadcast_plane_rows(rows, r0, n, src=0)
        assert np.array_equal(rows.numpy()[:r0 + n], frames[2][:r0 + n])
        if rank != 0 and r0 + n < rows.shape[0]:
            assert not rows.numpy()[r0 + n:].any()              # rows not yet published stay untouched
    mine = xd.shard_frames(NFRAMES, rank, world)
    o = Oracle()
    local = [o.search(frames[f + 1], [ref.numpy()], jobs)[0] for f in mine]
    allres = xd.gather_results(local, mine, NFRAMES, dst=0)
    if rank == 0:
        q.put([r.tobytes() for r in allres])
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_broadcast_and_gather():
    from x265_amd import dist as xd
    assert xd.shard_frames(8, 3, 8) == [3] and xd.shard_frames(5, 1, 2) == [1, 3]
    assert sorted(sum((xd.shard_frames(7, r, 3) for r in range(3)), [])) == list(range(7))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    from oracle_ffi import Oracle
    frames, jobs = _inputs()
    o = Oracle()
    for f in range(NFRAMES):
        exp = o.search(frames[f + 1], [frames[0]], jobs)[0]
        assert got[f] == exp.tobytes(), f
