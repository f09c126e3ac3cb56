"""Developer aid: reads the per-work-item trace of the shared-window pass (X265AMD_HIER_TRACE=<file>, -DXA_PROFILE build) and prints
how many work items are resident over time (100 MHz counter: 10 ns per tick)."""
import sys
import numpy as np
t = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 3).astype(np.int64)
n = int(sys.argv[2])
t = t[:n]
t0 = t[:, 0].min()
t -= t0
end = t[:, 2].max()
print("items %d, span %.1f us; staging mean %.1f us, whole item mean %.1f us (min %.1f, max %.1f)" % (
    n, end / 100.0, (t[:, 1] - t[:, 0]).mean() / 100.0, (t[:, 2] - t[:, 0]).mean() / 100.0,
    (t[:, 2] - t[:, 0]).min() / 100.0, (t[:, 2] - t[:, 0]).max() / 100.0))
step = max(1, end // 40)
for a in range(0, int(end) + 1, int(step)):
    res = int(((t[:, 0] <= a) & (t[:, 2] > a)).sum())
    stg = int(((t[:, 0] <= a) & (t[:, 1] > a)).sum())
    started = int((t[:, 0] <= a).sum())
    print("t %6.1f us: resident %5d (staging %5d), started %5d" % (a / 100.0, res, stg, started))
