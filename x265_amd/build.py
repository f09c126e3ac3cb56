"""Build the HIP library (x265_amd/csrc -> x265_amd/libx265amd.so) with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "x265_amd.hip")
OUT = os.environ.get("X265AMD_OUT", os.path.join(HERE, "libx265amd.so"))
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("me_device.h", "me_search.h", "me_kernels.h", "me_full.h", "me_hier.h", "me_had.h", "me_lowres.h", "me_field.h")] + \
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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off",           # the MV cost table must round exactly like the reference build
           *os.environ.get("X265AMD_EXTRA_FLAGS", "").split(),
           "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
           "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="-f" in sys.argv)
