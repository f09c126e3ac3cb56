"""The allocation-free entry point (x265amd_me_search_batch_device_noalloc) against the entry point it is a
variant of (x265amd_me_search_batch_device): bit-equal results on two calls in a row, the second with a
smaller input, and no device memory taken or given back by the calls."""
import numpy as np
import pytest

from x265_amd import jobs as J
from x265_amd import synth

pytestmark = pytest.mark.gpu


def _jobs(W, H, rng, n_each):
    parts = []
    for (w, h) in ((64, 64), (32, 16), (16, 16), (8, 8), (4, 8)):
        xs = rng.integers(0, (W - w) // 4 + 1, n_each) * 4
        ys = rng.integers(0, (H - h) // 4 + 1, n_each) * 4
        parts.append(J.make_jobs(W, H, xs, ys, w, h, method=J.FULL, subme=2, merange=12, qp=30,
                                 qmvp=(int(rng.integers(-20, 20)), int(rng.integers(-20, 20)))))
        parts.append(J.make_jobs(W, H, xs, ys, w, h, method=J.HEX, subme=5, merange=32, qp=34))
        parts.append(J.make_jobs(W, H, xs[: n_each // 2], ys[: n_each // 2], w, h, method=J.STAR, subme=1, merange=24, qp=22))
    return np.concatenate(parts)


def test_noalloc_entry_point_equals_the_allocating_one(oracle):
    import torch
    from x265_amd.api import Engine, X265AmdError
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    W, H = 416, 240
    fr = synth.make_sequence(W, H, 2, seed=77)
    ids = [eng.upload_plane(f) for f in fr]
    rng = np.random.default_rng(5)
    big = _jobs(W, H, rng, 24)
    small = _jobs(W, H, rng, 5)[::2]            # other jobs, fewer of them: stale scratch entries must not matter
    def free_bytes():
        return torch.cuda.mem_get_info(dev)[0]          # hipMemGetInfo: free device memory, whoever allocated

    def run(entry, jobs):
        dj = torch.from_numpy(np.frombuffer(np.ascontiguousarray(jobs).tobytes(), dtype=np.uint8).copy()).to(dev)
        do = torch.zeros(len(jobs) * 8, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        before = free_bytes()
        entry(ids[0], [ids[1]], dj.data_ptr(), len(jobs), do.data_ptr())
        eng.sync()
        after = free_bytes()
        return do.cpu().numpy().view(J.RESULT_DTYPE), before - after

    ref_big, _ = run(eng.search_device, big)
    ref_small, _ = run(eng.search_device, small)
    got_big, d1 = run(eng.search_device_noalloc, big)
    got_small, d2 = run(eng.search_device_noalloc, small)
    assert np.array_equal(got_big, ref_big), np.nonzero(got_big != ref_big)[0][:5]
    assert np.array_equal(got_small, ref_small), np.nonzero(got_small != ref_small)[0][:5]
    assert d1 == 0 and d2 == 0, (d1, d2)        # free device memory unchanged across the calls
    exp, _ = oracle.search(fr[0], [fr[1]], small, threads=4)
    assert np.array_equal(got_small, exp)
    # beyond the fixed scratch: a clear error, nothing launched
    dj = torch.zeros(48, dtype=torch.uint8, device=dev)
    with pytest.raises(X265AmdError, match="65536"):
        eng.search_device_noalloc(ids[0], [ids[1]], dj.data_ptr(), 65537, dj.data_ptr())
    for i in ids:
        eng.release_plane(i)
    eng.close()


def test_device_array_entry_points_refuse_bad_job_fields(oracle):
    """Jobs that live only in device memory cannot be checked on the host: the kernels apply the host entry
    points' field checks per job, give a failing job (MV 0, cost X265AMD_ERR_INVALID) without touching a plane
    or cost table through its fields, and search the jobs around it as usual - through both device-array
    entry points; a second context using the same GPU's scratch array right behind the first gets the same
    results (the library serialises the two calls on the GPU)."""
    import torch
    from x265_amd.api import Engine
    dev = torch.device("cuda", 0)
    eng, eng2 = Engine(0), Engine(0)
    W, H = 416, 240
    fr = synth.make_sequence(W, H, 2, seed=78)
    ids = [eng.upload_plane(f) for f in fr]
    ids2 = [eng2.upload_plane(f) for f in fr]
    rng = np.random.default_rng(6)
    jobs = _jobs(W, H, rng, 6)
    bad = {3: ("qp", 200), 10: ("ref_idx", 7), 17: ("numc", 9), 25: ("method", 77), 31: ("w", 5), 40: ("subme", 8), 44: ("icost", 3)}
    broken = jobs.copy()
    for i, (field, value) in bad.items():
        broken[i][field] = value
    good = np.array([i for i in range(len(jobs)) if i not in bad])
    exp, _ = oracle.search(fr[0], [fr[1]], np.ascontiguousarray(jobs[good]), threads=4)
    dj = torch.from_numpy(np.frombuffer(broken.tobytes(), dtype=np.uint8).copy()).to(dev)
    outs = []
    for e, p, entry in ((eng, ids, "search_device"), (eng, ids, "search_device_noalloc"), (eng2, ids2, "search_device_noalloc")):
        do = torch.zeros(len(jobs) * 8, dtype=torch.uint8, device=dev)
        getattr(e, entry)(p[0], [p[1]], dj.data_ptr(), len(jobs), do.data_ptr())
        outs.append((e, do))
    for e, do in outs:
        e.sync()
        got = do.cpu().numpy().view(J.RESULT_DTYPE)
        assert np.array_equal(got[good], exp), np.nonzero(got[good] != exp)[0][:5]
        for i in bad:
            assert (int(got[i]["mvx"]), int(got[i]["mvy"]), int(got[i]["cost"])) == (0, 0, -2), (i, got[i])
    eng.close()
    eng2.close()
