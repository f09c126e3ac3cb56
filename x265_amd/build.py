"""Build the HIP library (x265_amd/csrc -> x265_amd/libx265amd.so) with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "x265_amd.hip")
PART = os.path.join(HERE, "csrc", "x265_amd_part.hip")
N_PARTS = 5
OUT = os.environ.get("X265AMD_OUT", os.path.join(HERE, "libx265amd.so"))
DEPS = [SRC, PART] + [os.path.join(HERE, "csrc", f) for f in ("x265_amd_inst.h", "me_device.h", "me_search.h", "me_kernels.h", "me_full.h", "me_hier.h", "me_had.h", "me_lowres.h", "me_field.h")] + \
       [os.path.join(os.path.dirname(HERE), "include", "x265_amd.h")]


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


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
    jobs = [([hipcc, *flags, "-c", "-o", os.path.join(objdir, "main.o"), SRC])]
    for k in range(1, N_PARTS + 1):
        jobs.append([hipcc, *flags, "-DXA_PART=%d" % k, "-c", "-o", os.path.join(objdir, "part%d.o" % k), PART])
    procs = []
    for cmd in jobs:
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise subprocess.CalledProcessError(max(rcs), jobs[rcs.index(max(rcs))])
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [c[c.index("-o") + 1] for c in jobs]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    return OUT


if __name__ == "__main__":
    build(force="-f" in sys.argv)
