"""ctypes bindings for the CHECKERS used by the tests:

* ``Oracle``   -- oracle/libme_oracle.so, this repo's plain-C restatement (travels to the GPU box);
* ``Reference``-- oracle/_ref/libx265ref{8,16}.so, the reference's own translation units compiled
                  unmodified by oracle/Makefile (only exists where /root/reference was available
                  at build time; the prebuilt .so travels, the sources do not).

Test infrastructure only: nothing under x265_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "libme_oracle.so")
REF_SO = {8: os.path.join(ROOT, "oracle", "_ref", "libx265ref8.so"),
          16: os.path.join(ROOT, "oracle", "_ref", "libx265ref16.so"),
          # 8-bit pixels with the reference's intrinsic (SSE3 .. SSE4.1) primitives dispatched over the C ones
          "8v": os.path.join(ROOT, "oracle", "_ref", "libx265ref8v.so")}

import sys
sys.path.insert(0, ROOT)
from x265_amd.jobs import JOB_DTYPE, RESULT_DTYPE  # noqa: E402
from x265_amd.synth import MARGIN  # noqa: E402



def field_grid(pic_w, pic_h, pu_w, pu_h, part_mode=0):
    """(PUs per row, PU rows) of a chained field: whole CUs only.  part_mode 1 / 2: CUs of pu_w x 2 pu_h / 2 pu_w x pu_h;
    4, 5 / 6, 7 (asymmetric partitions): CUs of pu_w x pu_h with two PU rows / two PU columns each"""
    gw, gh = pic_w // pu_w, pic_h // pu_h
    if part_mode == 1:
        gh &= ~1
    if part_mode == 2:
        gw &= ~1
    if part_mode in (4, 5):
        gh *= 2
    if part_mode in (6, 7):
        gw *= 2
    return gw, gh

class MeoPlane(C.Structure):
    _fields_ = [("pix", C.c_void_p), ("stride", C.c_ssize_t), ("bytes_per_px", C.c_int),
                ("bit_depth", C.c_int), ("lowres", C.c_void_p * 4)]


def plane_origin(padded: np.ndarray, margin: int = MARGIN) -> int:
    """address of pixel (0,0) inside a padded plane"""
    return padded.ctypes.data + (margin * padded.shape[1] + margin) * padded.itemsize


def _ptr(a: np.ndarray, off_elems: int = 0) -> int:
    return a.ctypes.data + off_elems * a.itemsize


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            raise FileNotFoundError(ORACLE_SO + " (run `make -C oracle oracle` or __graft_entry__.build())")
        L = self.lib = C.CDLL(ORACLE_SO)
        for suf in ("u8", "u16"):
            f = getattr(L, "meo_cmp_" + suf)
            f.restype = C.c_int
            f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t]
            f = getattr(L, "meo_sad_xn_" + suf)
            f.restype = None
            f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_ssize_t, C.c_void_p]
            f = getattr(L, "meo_ipfilter_" + suf)
            f.restype = None
            f.argtypes = [C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_int]
        L.meo_mvcost_table.argtypes = [C.c_int, C.c_void_p]
        L.meo_mvcost.restype = C.c_int
        L.meo_bitcost.restype = C.c_int
        L.meo_partition_from_sizes.restype = C.c_int
        L.meo_search_batch.restype = C.c_int64
        L.meo_search_batch.argtypes = [C.POINTER(MeoPlane), C.POINTER(MeoPlane), C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.meo_buf_cost.restype = C.c_int
        L.meo_buf_cost.argtypes = [C.POINTER(MeoPlane), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t]
        L.meo_lowres_core_u8.argtypes = [C.c_void_p] * 5 + [C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int]
        L.meo_lowres_init.restype = None
        L.meo_lowres_init.argtypes = [C.c_void_p, C.c_ssize_t, C.c_int, C.POINTER(C.c_void_p), C.c_ssize_t,
                                      C.c_int, C.c_int, C.c_int, C.c_int]
        L.meo_lookahead_me.restype = C.c_int64
        L.meo_lookahead_me.argtypes = [C.POINTER(MeoPlane), C.POINTER(MeoPlane)] + [C.c_int] * 6 + [C.c_void_p]

    @staticmethod
    def _suf(a):
        return "u8" if a.itemsize == 1 else "u16"

    def cmp(self, kind, idx, fenc, fenc_off, fs, fref, fref_off, rs):
        return getattr(self.lib, "meo_cmp_" + self._suf(fenc))(kind, idx, _ptr(fenc, fenc_off), fs, _ptr(fref, fref_off), rs)

    def sad_xn(self, n, part, fenc, fenc_off, fref, fref_offs, rs):
        ptrs = (C.c_void_p * n)(*[_ptr(fref, o) for o in fref_offs])
        res = np.zeros(n, dtype=np.int32)
        getattr(self.lib, "meo_sad_xn_" + self._suf(fenc))(n, part, _ptr(fenc, fenc_off), ptrs, rs, res.ctypes.data)
        return res

    def mvcost_table(self, qp):
        out = np.zeros(65536, dtype=np.uint16)
        self.lib.meo_mvcost_table(qp, out.ctypes.data)
        return out

    def ipfilter(self, kind, src, src_off, ss, dst, dst_off, ds, w, h, frac, depth, pix_itemsize):
        suf = "u8" if pix_itemsize == 1 else "u16"
        getattr(self.lib, "meo_ipfilter_" + suf)(kind, _ptr(src, src_off), ss, _ptr(dst, dst_off), ds, w, h, frac, depth)

    @staticmethod
    def make_plane(padded: np.ndarray, bit_depth=None, margin: int = MARGIN, lowres=None) -> MeoPlane:
        p = MeoPlane()
        p.pix = plane_origin(padded, margin)
        p.stride = padded.shape[1]
        p.bytes_per_px = padded.itemsize
        p.bit_depth = bit_depth if bit_depth is not None else (8 if padded.itemsize == 1 else 10)
        if lowres is not None:
            for i in range(4):
                p.lowres[i] = plane_origin(lowres[i], margin)
        return p

    def lowres_init(self, src_padded: np.ndarray, margin: int = MARGIN):
        """(geometry, [F, H, V, C] padded planes) of a border-replicated source plane"""
        from x265_amd.lowres import lowres_geometry
        sw, sh = src_padded.shape[1] - 2 * margin, src_padded.shape[0] - 2 * margin
        g = lowres_geometry(sw, sh, margin, margin)
        planes = [np.zeros(g.padded_shape, dtype=src_padded.dtype) for _ in range(4)]
        ptrs = (C.c_void_p * 4)(*[plane_origin(p, margin) for p in planes])
        self.lib.meo_lowres_init(plane_origin(src_padded, margin), src_padded.shape[1], src_padded.itemsize, ptrs,
                                 g.stride, g.width, g.lines, margin, margin)
        return g, planes

    def lookahead_me(self, enc4, ref4, cu_w, cu_h, merange, method, subme, qp, bit_depth=None, margin: int = MARGIN):
        """results[cu_h * cu_w] (raster) and evaluation count of one lookahead motion search"""
        enc = self.make_plane(enc4[0], bit_depth, margin, lowres=enc4)
        ref = self.make_plane(ref4[0], bit_depth, margin, lowres=ref4)
        out = np.zeros(cu_w * cu_h, dtype=RESULT_DTYPE)
        ev = self.lib.meo_lookahead_me(C.byref(enc), C.byref(ref), cu_w, cu_h, merange, method, subme, qp, out.ctypes.data)
        return out, ev

    def mc_cost(self, enc_padded, ref_padded, pu_x, pu_y, w, h, qx, qy, kind, bit_depth=None, margin: int = MARGIN):
        enc = self.make_plane(enc_padded, bit_depth, margin)
        ref = self.make_plane(ref_padded, bit_depth, margin)
        return self.lib.meo_mc_cost(C.byref(enc), C.byref(ref), pu_x, pu_y, w, h, qx, qy, kind)

    def scale_col_mv(self, cur_poc, cur_ref_poc, col_poc, col_ref_poc, mvx, mvy):
        """xGetColMVP's scaling of a co-located MV by the POC distances"""
        x, y = C.c_int16(mvx), C.c_int16(mvy)
        self.lib.meo_scale_col_mv.restype = None
        self.lib.meo_scale_col_mv(cur_poc, cur_ref_poc, col_poc, col_ref_poc, C.byref(x), C.byref(y))
        return x.value, y.value

    def field_me(self, enc_padded, ref_padded, pu_w, pu_h, merange, method, subme, qp, bit_depth=None, margin: int = MARGIN,
                 part_mode: int = 0, col=None):
        """(results[gh, gw], mvp[gh, gw, 2], evals) of the whole-frame field with predInterSearch's predictor chain
        (part_mode 1 / 2: the parts of 2NxN / Nx2N CUs, 4..7: of 2NxnU / 2NxnD / nLx2N / nRx2N CUs of pu_w x pu_h)"""
        enc = self.make_plane(enc_padded, bit_depth, margin)
        ref = self.make_plane(ref_padded, bit_depth, margin)
        pw, ph = enc_padded.shape[1] - 2 * margin, enc_padded.shape[0] - 2 * margin
        gw, gh = field_grid(pw, ph, pu_w, pu_h, part_mode)
        out = np.zeros((gh, gw), dtype=RESULT_DTYPE)
        mvp = np.zeros((gh, gw, 2), dtype=np.int16)
        if col is not None:                 # (the co-located picture's motion per 16x16 block, RESULT_DTYPE, cost < 0 = none)
            col = np.ascontiguousarray(col, dtype=RESULT_DTYPE)
            assert col.size == ((pw + 15) // 16) * ((ph + 15) // 16)
        self.lib.meo_field_me_tmvp.restype = C.c_int64
        ev = self.lib.meo_field_me_tmvp(C.byref(enc), C.byref(ref), pw, ph, pu_w, pu_h, part_mode,
                                        col.ctypes.data_as(C.c_void_p) if col is not None else None, merange, method, subme, qp,
                                        out.ctypes.data_as(C.c_void_p), mvp.ctypes.data_as(C.c_void_p))
        return out, mvp, ev

    def search(self, enc_padded, ref_padded_list, jobs, threads=1, bit_depth=None, margin: int = MARGIN, lowres=False):
        """returns (results, evals).  lowres: every entry of ref_padded_list is a list of the four
        half-resolution planes (the search then takes the isLowres branches)"""
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        if not isinstance(ref_padded_list, (list, tuple)) or (lowres and isinstance(ref_padded_list[0], np.ndarray)):
            ref_padded_list = [ref_padded_list]
        enc = self.make_plane(enc_padded, bit_depth, margin)
        if lowres:
            refs = (MeoPlane * len(ref_padded_list))(*[self.make_plane(r[0], bit_depth, margin, lowres=r)
                                                       for r in ref_padded_list])
        else:
            refs = (MeoPlane * len(ref_padded_list))(*[self.make_plane(r, bit_depth, margin) for r in ref_padded_list])
        out = np.zeros(len(jobs), dtype=RESULT_DTYPE)
        evals = self.lib.meo_search_batch(C.byref(enc), refs, jobs.ctypes.data, len(jobs), out.ctypes.data, threads)
        return out, evals


class Reference:
    """The compiled reference: C primitives only (cpuid=1) for bits 8 / 16; "8v" = 8-bit pixels with the
    reference's intrinsic primitives selected by its own run-time dispatch (the CPU-baseline build)."""

    def __init__(self, bits=8):
        so = REF_SO[bits]
        if not os.path.exists(so):
            raise FileNotFoundError(so)
        L = self.lib = C.CDLL(so)
        self.vec = bits == "8v"
        bits = 8 if self.vec else bits
        self.bits = bits
        L.ref_init()
        assert L.ref_pixel_bytes() == bits // 8 and bool(L.ref_with_vec()) == self.vec
        L.ref_cmp.restype = C.c_int
        L.ref_cmp.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t]
        L.ref_sad_x3.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_ssize_t, C.c_void_p]
        L.ref_sad_x4.argtypes = [C.c_int] + [C.c_void_p] * 5 + [C.c_ssize_t, C.c_void_p]
        L.ref_ipfilter_pp.argtypes = [C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ref_ipfilter_ps_h.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ref_ipfilter_sp_v.argtypes = [C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int]
        L.ref_mvcost_table.argtypes = [C.c_int, C.c_void_p]
        L.ref_mvcost.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_void_p]
        L.ref_me_new.restype = C.c_void_p
        L.ref_me_free.argtypes = [C.c_void_p]
        L.ref_me_search.restype = C.c_int
        L.ref_me_search.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_me_buf.restype = C.c_int
        L.ref_me_buf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_ssize_t]
        L.ref_lowres_core.argtypes = [C.c_void_p] * 5 + [C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int]
        L.ref_bipred_search.restype = C.c_uint
        L.ref_bipred_search.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_lookahead_me.restype = None
        L.ref_lookahead_me.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_ssize_t] + \
                                      [C.c_int] * 6 + [C.c_void_p, C.c_void_p]
        self.me = L.ref_me_new()

    def set_bit_depth(self, depth):
        self.lib.ref_set_bit_depth(depth)

    def cmp(self, kind, idx, fenc, fenc_off, fs, fref, fref_off, rs):
        return self.lib.ref_cmp(kind, idx, _ptr(fenc, fenc_off), fs, _ptr(fref, fref_off), rs)

    def sad_xn(self, n, part, fenc, fenc_off, fref, fref_offs, rs):
        res = np.zeros(16, dtype=np.int32)
        ptrs = [_ptr(fref, o) for o in fref_offs]
        if n == 3:
            self.lib.ref_sad_x3(part, _ptr(fenc, fenc_off), *ptrs, rs, res.ctypes.data)
        else:
            self.lib.ref_sad_x4(part, _ptr(fenc, fenc_off), *ptrs, rs, res.ctypes.data)
        return res[:n].copy()

    def mvcost_table(self, qp):
        out = np.zeros(65536, dtype=np.uint16)
        self.lib.ref_mvcost_table(qp, out.ctypes.data)
        return out

    def mvcost_bits(self, qp, mvp, mv):
        a, b = C.c_int(), C.c_int()
        self.lib.ref_mvcost(qp, mvp[0], mvp[1], mv[0], mv[1], C.byref(a), C.byref(b))
        return a.value, b.value

    def ipfilter(self, kind, src, src_off, ss, dst, dst_off, ds, w, h, frac):
        if kind in (0, 1):
            self.lib.ref_ipfilter_pp(kind, _ptr(src, src_off), ss, _ptr(dst, dst_off), ds, w, h, frac)
        elif kind == 2:
            self.lib.ref_ipfilter_ps_h(_ptr(src, src_off), ss, _ptr(dst, dst_off), ds, w, h, frac)
        else:
            self.lib.ref_ipfilter_sp_v(_ptr(src, src_off), ss, _ptr(dst, dst_off), ds, w, h, frac)

    def lowres_init(self, src_padded: np.ndarray, margin: int = MARGIN):
        """frame_init_lowres_core of the compiled reference + replicated borders (numpy): what
        Lowres::init produces for a source whose margins hold replicated borders"""
        from x265_amd.lowres import lowres_geometry
        sw, sh = src_padded.shape[1] - 2 * margin, src_padded.shape[0] - 2 * margin
        g = lowres_geometry(sw, sh, margin, margin)
        inner = [np.zeros((g.lines, g.width), dtype=src_padded.dtype) for _ in range(4)]
        self.lib.ref_lowres_core(plane_origin(src_padded, margin), *[p.ctypes.data for p in inner],
                                 src_padded.shape[1], g.width, g.width, g.lines)
        planes = []
        for p in inner:
            q = np.zeros(g.padded_shape, dtype=src_padded.dtype)
            q[:, :g.width + 2 * margin] = np.pad(p, margin, mode="edge")
            planes.append(q)
        return g, planes

    def lookahead_me(self, enc4, ref4, cu_w, cu_h, merange, method, subme, qp, margin: int = MARGIN):
        ep = (C.c_void_p * 4)(*[plane_origin(p, margin) for p in enc4])
        rp = (C.c_void_p * 4)(*[plane_origin(p, margin) for p in ref4])
        mv = np.zeros((cu_w * cu_h, 2), dtype=np.int16)
        cost = np.zeros(cu_w * cu_h, dtype=np.int32)
        self.lib.ref_lookahead_me(self.me, ep, rp, enc4[0].shape[1], cu_w, cu_h, merange, method, subme, qp,
                                  mv.ctypes.data, cost.ctypes.data)
        out = np.zeros(cu_w * cu_h, dtype=RESULT_DTYPE)
        out["mvx"], out["mvy"], out["cost"] = mv[:, 0], mv[:, 1], cost
        return out

    def field_me(self, enc_padded, ref_padded, pu_w, pu_h, merange, method, subme, qp, margin: int = MARGIN, part_mode: int = 0,
                 col=None):
        """(results[gh, gw], mvp[gh, gw, 2]): the reference's MotionEstimate driven over a PU grid with
        predInterSearch's predictor chain (oracle/ref_shim.cpp: ref_field_me_parts)"""
        pw, ph = enc_padded.shape[1] - 2 * margin, enc_padded.shape[0] - 2 * margin
        gw, gh = field_grid(pw, ph, pu_w, pu_h, part_mode)
        mv = np.zeros((gh * gw, 2), dtype=np.int16)
        cost = np.zeros(gh * gw, dtype=np.int32)
        mvp = np.zeros((gh, gw, 2), dtype=np.int16)
        assert enc_padded.shape == ref_padded.shape
        if col is not None:
            col = np.ascontiguousarray(col, dtype=RESULT_DTYPE)
            assert col.size == ((pw + 15) // 16) * ((ph + 15) // 16)
        self.lib.ref_field_me_tmvp.restype = None
        self.lib.ref_field_me_tmvp.argtypes = ([C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t] + [C.c_int] * 5 + [C.c_void_p] +
                                               [C.c_int] * 4 + [C.c_void_p] * 3)
        self.lib.ref_field_me_tmvp(self.me, plane_origin(enc_padded, margin), plane_origin(ref_padded, margin), enc_padded.shape[1],
                                   pw, ph, pu_w, pu_h, part_mode, col.ctypes.data if col is not None else None,
                                   merange, method, subme, qp, mv.ctypes.data, cost.ctypes.data, mvp.ctypes.data)
        out = np.zeros(gh * gw, dtype=RESULT_DTYPE)
        out["mvx"], out["mvy"], out["cost"] = mv[:, 0], mv[:, 1], cost
        return out.reshape(gh, gw), mvp

    def search(self, enc_padded, ref_padded_list, jobs, margin: int = MARGIN, lowres=False):
        """Runs every job through MotionEstimate::motionEstimate; returns RESULT_DTYPE array.
        Jobs with icost != 0 have no reference counterpart and are rejected.
        lowres: every entry of ref_padded_list is a list of the four half-resolution planes."""
        if not isinstance(ref_padded_list, (list, tuple)) or (lowres and isinstance(ref_padded_list[0], np.ndarray)):
            ref_padded_list = [ref_padded_list]
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        out = np.zeros(len(jobs), dtype=RESULT_DTYPE)
        enc0 = plane_origin(enc_padded, margin)
        es = enc_padded.shape[1]
        for i, j in enumerate(jobs):
            assert j["icost"] == 0
            ref = ref_padded_list[int(j["ref_idx"])]
            lr = None
            if lowres:
                lr = (C.c_void_p * 4)(*[plane_origin(p, margin) for p in ref])
                ref = ref[0]
            assert ref.shape[1] == es, "the reference applies one blockOffset to both planes"
            mvmin = np.array([j["mvmin_x"], j["mvmin_y"]], dtype=np.int16)
            mvmax = np.array([j["mvmax_x"], j["mvmax_y"]], dtype=np.int16)
            qmvp = np.array([j["qmvp_x"], j["qmvp_y"]], dtype=np.int16)
            mvc = np.ascontiguousarray(j["mvc"], dtype=np.int16)
            omv = np.zeros(2, dtype=np.int16)
            if int(j["method"]) == 5:       # the bidirectional-refinement search (no MotionEstimate method)
                cost = self.lib.ref_bipred_search(self.me, enc0, es, plane_origin(ref, margin), ref.shape[1],
                                                  int(j["pu_y"]) * es + int(j["pu_x"]), int(j["w"]), int(j["h"]),
                                                  mvmin.ctypes.data, mvmax.ctypes.data, qmvp.ctypes.data, int(j["qp"]),
                                                  omv.ctypes.data)
                out[i] = (omv[0], omv[1], cost)
                continue
            cost = self.lib.ref_me_search(self.me, enc0, es, plane_origin(ref, margin), ref.shape[1], lr,
                                          int(j["pu_y"]) * es + int(j["pu_x"]), int(j["w"]), int(j["h"]),
                                          mvmin.ctypes.data, mvmax.ctypes.data, qmvp.ctypes.data,
                                          int(j["numc"]), mvc.ctypes.data, int(j["merange"]),
                                          int(j["method"]), int(j["subme"]), int(j["qp"]), omv.ctypes.data)
            out[i] = (omv[0], omv[1], cost)
        return out


def have_reference(bits=8) -> bool:
    return os.path.exists(REF_SO[bits])
