"""Build the HIP library (x265_amd/csrc -> x265_amd/libx265amd.so) with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "x265_amd.hip")
PART = os.path.join(HERE, "csrc", "x265_amd_part.hip")
N_PARTS = 7
OUT = os.environ.get("X265AMD_OUT", os.path.join(HERE, "libx265amd.so"))
PUBLIC_H = os.path.join(os.path.dirname(HERE), "include", "x265_amd.h")
# headers the instantiation units (x265_amd_part.hip) include, and those only the main unit includes
PART_H = [os.path.join(HERE, "csrc", f) for f in ("x265_amd_inst.h", "me_device.h", "me_search.h", "me_kernels.h", "me_lowres.h", "me_field.h", "me_lane.h")]
MAIN_ONLY_H = [os.path.join(HERE, "csrc", f) for f in ("me_full.h", "me_hier.h", "me_had.h")]
DEPS = [SRC, PART, PUBLIC_H] + PART_H + MAIN_ONLY_H


def source_hash() -> str:
    """SHA-1 (12 hex digits) over the library's sources, in a fixed order: what x265amd_build_info() reports"""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(d for d in DEPS if os.path.exists(d)):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def _stale(obj: str, deps) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
             "-ffp-contract=off",         # the MV cost table must round exactly like the reference build
             *os.environ.get("X265AMD_EXTRA_FLAGS", "").split(),
             "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
    # the heavy kernel instantiations are split over N_PARTS translation units that compile side by side with the main one
    objdir = OUT + ".obj"
    os.makedirs(objdir, exist_ok=True)
    # (an object is rebuilt only when one of ITS sources is newer; the flags are recorded beside the objects, so a change
    # of flags rebuilds everything)
    stamp = os.path.join(objdir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    # (the source hash goes into the main unit only, which is rebuilt whenever any source changes anyway)
    jobs = [([hipcc, *flags, '-DXA_SRC_HASH="%s"' % source_hash(), "-c", "-o", os.path.join(objdir, "main.o"), SRC], [SRC, PUBLIC_H] + PART_H + MAIN_ONLY_H)]
    for k in range(1, N_PARTS + 1):
        jobs.append(([hipcc, *flags, "-DXA_PART=%d" % k, "-c", "-o", os.path.join(objdir, "part%d.o" % k), PART], [PART, PUBLIC_H] + PART_H))
    objs = [c[c.index("-o") + 1] for c, _ in jobs]
    todo = [c for c, deps in jobs if force or not same_flags or _stale(c[c.index("-o") + 1], deps)]
    procs = []
    for cmd in todo:
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), todo[rcs.index(max(rcs))])
    open(stamp, "w").write(" ".join(flags))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    build(force="-f" in sys.argv)
