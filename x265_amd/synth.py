"""Deterministic synthetic luma planes for motion-estimation tests and benchmarks.

Planes are laid out the way the encoder's picture buffers are
(reference: source/Lib/TLibCommon/TComPicYuv.cpp:68-120): a border of
``MARGIN`` replicated pixels on every side, stride = width + 2*MARGIN, and the
"plane pointer" addresses pixel (0,0) inside the padded allocation.

Everything here is numpy-only and regenerates identically everywhere from
(seed, frame index), so no pixel data has to be stored or shipped.
"""
from __future__ import annotations

import numpy as np

MARGIN = 80  # 64 (max CU) + 16, reference: TComPicYuv.cpp:83-85


def _hash32(a: np.ndarray) -> np.ndarray:
    """xorshift-multiply integer hash on uint32 arrays (vectorised, counter based)."""
    a = a.astype(np.uint32, copy=True)
    a ^= a << np.uint32(13)
    a ^= a >> np.uint32(17)
    a ^= a << np.uint32(5)
    a *= np.uint32(0x9E3779B1)
    a ^= a >> np.uint32(15)
    a *= np.uint32(0x85EBCA77)
    a ^= a >> np.uint32(13)
    return a


def _noise(seed: int, h: int, w: int, amp: int) -> np.ndarray:
    """uniform integer noise in [-amp, amp], shape (h, w)"""
    idx = np.arange(h * w, dtype=np.uint32).reshape(h, w)
    with np.errstate(over="ignore"):
        r = _hash32(idx + np.uint32((seed * 0x632BE5AB) & 0xFFFFFFFF))
    return (r % np.uint32(2 * amp + 1)).astype(np.int32) - amp


def _tri(v: np.ndarray) -> np.ndarray:
    """triangle wave with period 2 and range [-1, 1]"""
    f = v - np.floor(v)
    return 4.0 * np.abs(f - 0.5) - 1.0


def base_frame(width: int, height: int, seed: int = 0x1234ABCD, bit_depth: int = 8) -> np.ndarray:
    """Frame 0: smooth structure + texture + noise, int32 array (height, width) in sample units."""
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    p = 128.0 + 64.0 * np.sin(x / 37.0) * np.cos(y / 23.0) + 32.0 * _tri((x + 2.0 * y) / 22.0)
    p = np.rint(p).astype(np.int32) + _noise(seed, height, width, 12)
    p = np.clip(p, 0, 255)
    if bit_depth > 8:
        p = np.clip(p * (1 << (bit_depth - 8)) + _noise(seed ^ 0x55AA, height, width, (1 << (bit_depth - 8)) - 1),
                    0, (1 << bit_depth) - 1)
    return p


def moved_frame(base: np.ndarray, t: int, seed: int = 0x1234ABCD, bit_depth: int = 8,
                global_mv=(3, -2), local_amp: int = 40, block: int = 64) -> np.ndarray:
    """Frame t > 0: the base frame displaced by a global motion of ``global_mv`` per frame
    plus a per-``block`` local offset in [-local_amp, local_amp], plus fresh low noise."""
    h, w = base.shape
    by, bx = (h + block - 1) // block, (w + block - 1) // block
    with np.errstate(over="ignore"):
        r = _hash32(np.arange(by * bx * 2, dtype=np.uint32) + np.uint32(((seed + t) * 0x2545F491) & 0xFFFFFFFF))
    if local_amp > 0:
        off = (r % np.uint32(2 * local_amp + 1)).astype(np.int32) - local_amp
    else:
        off = np.zeros(by * bx * 2, dtype=np.int32)
    lx = off[0::2].reshape(by, bx)
    ly = off[1::2].reshape(by, bx)
    yy, xx = np.mgrid[0:h, 0:w]
    sx = xx - global_mv[0] * t - lx[yy // block, xx // block]
    sy = yy - global_mv[1] * t - ly[yy // block, xx // block]
    sx = np.clip(sx, 0, w - 1)
    sy = np.clip(sy, 0, h - 1)
    amp = 2 * (1 << (bit_depth - 8))
    out = base[sy, sx] + _noise(seed + 977 * t, h, w, amp)
    return np.clip(out, 0, (1 << bit_depth) - 1)


def noise_frame(width: int, height: int, seed: int, bit_depth: int = 8) -> np.ndarray:
    """i.i.d. uniform samples: the structure-free worst case."""
    idx = np.arange(width * height, dtype=np.uint32).reshape(height, width)
    with np.errstate(over="ignore"):
        r = _hash32(idx ^ np.uint32((seed * 0x7FEB352D) & 0xFFFFFFFF))
    return (r >> np.uint32(32 - bit_depth)).astype(np.int32)


def pad_plane(img: np.ndarray, bit_depth: int = 8, margin: int = MARGIN) -> np.ndarray:
    """Replicate borders; returns the padded (height+2m, width+2m) array of uint8/uint16."""
    dt = np.uint8 if bit_depth <= 8 else np.uint16
    return np.ascontiguousarray(np.pad(img.astype(dt), margin, mode="edge"))


def make_sequence(width: int, height: int, nframes: int, seed: int = 0x1234ABCD, bit_depth: int = 8,
                  kind: str = "motion"):
    """List of padded planes: frame 0 = base, frame t = base moved t steps (kind='motion'),
    or independent noise frames (kind='noise')."""
    if kind == "noise":
        return [pad_plane(noise_frame(width, height, seed + i, bit_depth), bit_depth) for i in range(nframes)]
    base = base_frame(width, height, seed, bit_depth)
    frames = [pad_plane(base, bit_depth)]
    for t in range(1, nframes):
        frames.append(pad_plane(moved_frame(base, t, seed, bit_depth), bit_depth))
    return frames
