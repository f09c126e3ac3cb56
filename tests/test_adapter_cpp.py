"""The C++ host mirror of the reference's MotionEstimate interface (x265_amd/host/x265_amd_host.hpp)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "tests", "cpp", "adapter_demo")


def test_adapter_header_compiles():
    """header + demo driver compile against the C ABI (syntax only: no GPU needed)"""
    src = os.path.join(ROOT, "tests", "cpp", "adapter_demo.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", src])


@pytest.mark.gpu
def test_adapter_results_match_oracle(oracle, tmp_path):
    from x265_amd import jobs as J, synth
    assert os.path.exists(DEMO), "build it with __graft_entry__.build()"
    W, H, M = 512, 256, synth.MARGIN
    fr = synth.make_sequence(W, H, 2, seed=404)
    e, r = str(tmp_path / "enc.raw"), str(tmp_path / "ref.raw")
    fr[0].tofile(e)
    fr[1].tofile(r)
    out = subprocess.run([DEMO, str(W), str(H), str(M), e, r], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l.split() for l in out.stdout.strip().splitlines()]
    stride = W + 2 * M
    # (1) single calls
    for l in [l for l in lines if l[0] == "single"]:
        method = int(l[1])
        jb = J.make_jobs(W, H, [48], [32], 16, 16, method=method, subme=5, merange=16, qp=30, qmvp=(5, -3),
                         mvc=[[8, -4], [-12, 6]])
        jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = -16, -16, 16, 16
        exp, _ = oracle.search(fr[0], [fr[1]], jb)
        assert (int(l[2]), int(l[3]), int(l[4])) == (int(exp["mvx"][0]), int(exp["mvy"][0]), int(exp["cost"][0])), l
    # (1b) linear offsets anywhere in the padded plane map to the same PU as explicit coordinates
    rows = [l for l in lines if l[0] == "srcpu"]
    assert len(rows) == 6
    for l in rows:
        x, y = int(l[1]), int(l[2])
        assert l[3:6] == l[6:9], l
        jb = J.make_jobs(W, H, [x], [y], 16, 16, method=J.HEX, subme=2, merange=8, qp=30, qmvp=(2, 1))
        jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = -8, -8, 8, 8
        exp, _ = oracle.search(fr[0], [fr[1]], jb)
        assert (int(l[3]), int(l[4]), int(l[5])) == (int(exp["mvx"][0]), int(exp["mvy"][0]), int(exp["cost"][0])), l
    # (2) batch
    xs = np.arange(0, W - 15, 16)
    jb = J.make_jobs(W, H, xs, np.full_like(xs, 16), 16, 16, method=J.HEX, subme=2, merange=24, qp=30)
    jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = -24, -16, 24, 16
    exp, _ = oracle.search(fr[0], [fr[1]], jb)
    got = [l for l in lines if l[0] == "batch"]
    assert len(got) == len(jb)
    for l, ex in zip(got, exp):
        assert (int(l[2]), int(l[3]), int(l[4])) == (int(ex["mvx"]), int(ex["mvy"]), int(ex["cost"]))
    # (3) buf costs: PU (64,32) 32x32 vs ref block at (60,35)
    flat0, flat1 = fr[0].reshape(-1), fr[1].reshape(-1)
    eo, ro = (32 + M) * stride + 64 + M, (35 + M) * stride + 60 + M
    part = J.partition_from_sizes(32, 32)
    buf = [l for l in lines if l[0] == "buf"][0]
    assert [int(v) for v in buf[1:]] == [oracle.cmp(k, part, flat0, eo, stride, flat1, ro, stride) for k in (0, 1, 2)]
    # (4) primitive table: 16x16 at plane origin, ref at +5; sad_x4 reads fenc with stride 64
    o0 = M * stride + M
    p16 = J.partition_from_sizes(16, 16)
    prim = [l for l in lines if l[0] == "prim"][0]
    exp3 = [oracle.cmp(k, p16, flat0, o0, stride, flat1, o0 + 5, stride) for k in (0, 1, 2)]
    x4 = oracle.sad_xn(4, p16, flat0, o0, flat1, [o0, o0 + 1, o0 + 2, o0 + 3], stride)
    assert [int(v) for v in prim[1:]] == exp3 + [int(x4[0]), int(x4[3])]
    # (5) lookahead mirror: the whole-frame field of the half-resolution pair
    g, e4 = oracle.lowres_init(fr[0])
    _, r4 = oracle.lowres_init(fr[1])
    exp, _ = oracle.lookahead_me(e4, r4, g.cu_w, g.cu_h, 16, J.HEX, 1, 12)
    look = [l for l in lines if l[0] == "look"][0]
    assert [int(v) for v in look[1:]] == [g.cu_w, g.cu_h, g.cu_w * g.cu_h]
    cus = [l for l in lines if l[0] == "cu"]
    assert len(cus) == len(exp)
    for l, ex in zip(cus, exp):
        assert (int(l[2]), int(l[3]), int(l[4])) == (int(ex["mvx"]), int(ex["mvy"]), int(ex["cost"])), l
    # (6) motion field planned ahead: hits return the planned search's result, misses are searched on the spot;
    #     either way the answer is the oracle's for the arguments actually passed
    rows = [l for l in lines if l[0] == "field"]
    assert len(rows) == W // 16
    n_hit = 0
    for l in rows:
        x, foreseen = int(l[1]), int(l[2])
        n_hit += foreseen
        jb = J.make_jobs(W, H, [x], [48], 16, 16, method=J.STAR, subme=3, merange=20, qp=30,
                         qmvp=(4, -(x // 16) if foreseen else 7), mvc=[[x // 8, 2]])
        jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = -20, -20, 20, 20
        exp, _ = oracle.search(fr[0], [fr[1]], jb)
        assert (int(l[3]), int(l[4]), int(l[5])) == (int(exp["mvx"][0]), int(exp["mvy"][0]), int(exp["cost"][0])), l
    stats = [l for l in lines if l[0] == "fieldstats"][0]
    assert [int(v) for v in stats[1:]] == [n_hit, len(rows) - n_hit, len(rows)]
    # (4b) bidirectional refinement through the mirror
    jb = J.make_jobs(W, H, [96], [64], 32, 16, method=J.BIPRED, subme=0, merange=0, qp=30, qmvp=(9, -7))
    jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = -1, -6, 7, 2
    exp, _ = oracle.search(fr[0], [fr[1]], jb)
    bp = [l for l in lines if l[0] == "bipred"][0]
    assert [int(v) for v in bp[1:]] == [int(exp["mvx"][0]), int(exp["mvy"][0]), int(exp["cost"][0])]
    # (7) the chained field behind the unchanged call site: every request - hit or miss - returns what the oracle
    #     computes for the arguments actually passed, and at least 90 % of the requests are served from the field
    rows = [l for l in lines if l[0] == "chain"]
    assert len(rows) == (W // 16) * (H // 16)
    parts = []
    for l in rows:
        v = [int(t) for t in l[1:]]
        jb = J.make_jobs(W, H, [v[0] * 16], [v[1] * 16], 16, 16, method=J.HEX, subme=5, merange=32, qp=30, qmvp=(v[2], v[3]),
                         mvc=[[v[8], v[9]], [v[10], v[11]], [v[12], v[13]]])
        jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = v[4], v[5], v[6], v[7]
        parts.append(jb)
    exp, _ = oracle.search(fr[0], [fr[1]], np.concatenate(parts), threads=4)
    for l, ex in zip(rows, exp):
        assert (int(l[15]), int(l[16]), int(l[17])) == (int(ex["mvx"]), int(ex["mvy"]), int(ex["cost"])), l
    st = [int(v) for v in [l for l in lines if l[0] == "chainstats"][0][1:]]
    assert st[0] + st[1] == st[2] and st[0] >= 0.9 * st[2], st
    # (8) fields of several PU sizes and partition modes from one device call, each behind its own walk of the CUs
    rows = [l for l in lines if l[0] == "mchain"]
    n_exp = (W // 16) * (H // 16) + (W // 32) * (H // 32) + 2 * (W // 16) * (H // 16) + 2 * (W // 16) * (H // 16)
    n_exp += 2 * (W // 32) * (H // 32) + 2 * (W // 16) * (H // 16)          # the 2NxnU field of 32x32, the nRx2N field of 16x16
    assert len(rows) == n_exp
    parts = []
    for l in rows:
        v = [int(t) for t in l[1:]]
        jb = J.make_jobs(W, H, [v[3]], [v[4]], v[0], v[1], method=J.HEX, subme=3, merange=24, qp=28, qmvp=(v[5], v[6]),
                         mvc=[[v[11], v[12]], [v[13], v[14]], [v[15], v[16]]])
        jb["mvmin_x"], jb["mvmin_y"], jb["mvmax_x"], jb["mvmax_y"] = v[7], v[8], v[9], v[10]
        parts.append(jb)
    exp, _ = oracle.search(fr[0], [fr[1]], np.concatenate(parts), threads=4)
    for l, ex in zip(rows, exp):
        assert (int(l[18]), int(l[19]), int(l[20])) == (int(ex["mvx"]), int(ex["mvy"]), int(ex["cost"])), l
    # every part coded before its field's intra part is a hit: the misses are the parts whose neighbourhood the walk's own decision
    # (the intra part's zero MV) changed, and what follows from them
    seen_intra = {}
    for l in rows:
        key = (l[3], l[1], l[2]) if l[3] in ("0", "1", "2") else (l[3],)      # (one field per asymmetric mode; its parts differ in size)
        if l[21] == "1":
            seen_intra[key] = True
        elif not seen_intra.get(key):
            assert l[22] == "1", l
    st = [int(v) for v in [l for l in lines if l[0] == "mchainstats"][0][1:]]
    assert st[0] + st[1] == st[2] == n_exp and st[0] >= 0.9 * st[2], st
