"""Geometry of the lookahead's half-resolution pictures (host-side mirror of Lowres::create,
reference: source/common/lowres.cpp:30-58, and of Lookahead's CU counts, encoder/slicetype.cpp:74-75)."""
from __future__ import annotations

from dataclasses import dataclass

LOWRES_CU_SIZE = 8          # X265_LOWRES_CU_SIZE, source/common/lowres.h:38
LOOKAHEAD_QP = 12           # X265_LOOKAHEAD_QP at 8-bit, source/encoder/slicetype.h:31
LOOKAHEAD_MERANGE = 16      # source/encoder/slicetype.cpp:73
LOOKAHEAD_SUBME = 1         # source/encoder/slicetype.cpp:72
LOOKAHEAD_METHOD = 1        # X265_HEX_SEARCH, source/encoder/slicetype.cpp:71


@dataclass(frozen=True)
class LowresGeometry:
    width: int              # pixels, a multiple of 8
    lines: int
    stride: int             # pixels; (source width / 2 + 2 * margin_x) rounded up to 32
    margin_x: int
    margin_y: int
    cu_w: int               # widthInCU
    cu_h: int               # heightInCU

    @property
    def padded_shape(self):
        return (self.lines + 2 * self.margin_y, self.stride)

    @property
    def plane_elems(self):
        return self.stride * (self.lines + 2 * self.margin_y)


def lowres_geometry(src_width: int, src_height: int, margin_x: int, margin_y: int) -> LowresGeometry:
    width, lines = src_width // 2, src_height // 2
    stride = width + 2 * margin_x
    if stride & 31:
        stride += 32 - (stride & 31)
    cu_w = (width + LOWRES_CU_SIZE - 1) // LOWRES_CU_SIZE
    cu_h = (lines + LOWRES_CU_SIZE - 1) // LOWRES_CU_SIZE
    return LowresGeometry(cu_w * LOWRES_CU_SIZE, cu_h * LOWRES_CU_SIZE, stride, margin_x, margin_y, cu_w, cu_h)
