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
