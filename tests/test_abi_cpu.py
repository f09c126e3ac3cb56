"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU and exports
every symbol include/x265_amd.h declares; struct layouts match the numpy images."""
import ctypes as C
import os
import re

import numpy as np

from x265_amd import api
from x265_amd.jobs import JOB_DTYPE, RESULT_DTYPE, partition_from_sizes, search_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    hdr = open(os.path.join(ROOT, "include", "x265_amd.h")).read()
    names = set(re.findall(r"\b(x265amd_[a-z0-9_]+)\s*\(", hdr))
    names -= {"x265amd_pixelcmp_t", "x265amd_pixelcmp_x3_t", "x265amd_pixelcmp_x4_t"}
    assert len(names) == 56, sorted(names)
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_partition_enum_and_struct_sizes():
    lib = api.load_library()
    dims = (4, 8, 12, 16, 24, 32, 48, 64)
    for w in dims:
        for h in dims:
            assert lib.x265amd_partition_from_sizes(w, h) == partition_from_sizes(w, h)
    assert lib.x265amd_partition_from_sizes(20, 8) == -1
    assert JOB_DTYPE.itemsize == 48 and RESULT_DTYPE.itemsize == 8 and api.PAIR_DTYPE.itemsize == 24
    assert JOB_DTYPE.fields["mvc"][1] == 32 and JOB_DTYPE.fields["merange"][1] == 24
    assert C.sizeof(api.Primitives) == 8 * (64 * 5 + 5)


def test_no_device_is_an_error_not_a_fallback():
    """Without a GPU the context cannot be created; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        return
    lib = api.load_library()
    h = C.c_void_p()
    assert lib.x265amd_create(0, C.byref(h)) == -1   # X265AMD_ERR_NO_DEVICE
    assert not h.value


def test_search_range_rule():
    # reference: TEncSearch.cpp:3007-3024 with clipMv (TComDataCU.cpp:2724-2736)
    x0, y0, x1, y1 = search_range(1920, 1080, 0, 0, 0, 0, 64)
    assert (x0, y0, x1, y1) == (-64, -64, 64, 64)
    x0, y0, x1, y1 = search_range(1920, 1080, 1912, 1072, 0, 0, 64)
    assert (x1, y1) == (15, 15) and (x0, y0) == (-64, -64)
    x0, y0, x1, y1 = search_range(1920, 1080, 64, 64, -4 * 200, 0, 64)
    assert x0 == -135 and x1 == -71   # predictor clipped to -135 px, window follows
