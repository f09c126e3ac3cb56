"""Developer aid: runs tests/cpp/adapter_demo on a 512x256 synthetic frame pair and prints the hit statistics of its chained-field
walks (per shape, and the positions of the parts that missed for the 2NxN and nRx2N fields)."""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
from x265_amd import synth
W, H, M = 512, 256, synth.MARGIN
fr = synth.make_sequence(W, H, 2, seed=404)
fr[0].tofile("/tmp/enc.raw"); fr[1].tofile("/tmp/ref.raw")
out = subprocess.run(["tests/cpp/adapter_demo", str(W), str(H), str(M), "/tmp/enc.raw", "/tmp/ref.raw"], capture_output=True, text=True, timeout=100)
for l in out.stdout.splitlines():
    if l.startswith(("mchainshape", "mchainstats", "chainstats")):
        print(l)
rows = [l.split() for l in out.stdout.splitlines() if l.startswith("mchain ")]
for pm, gw in ((1, 16), (7, 32)):
    sel = [r for r in rows if r[3] == str(pm)]
    print("mode", pm)
    for y in range(len(sel) // gw if pm == 7 else 16):
        pass
    # misses by PU position
    print(" ".join("%s,%s" % (r[4], r[5]) for r in sel if r[-1] == "0"))
print(out.stderr[-500:])
