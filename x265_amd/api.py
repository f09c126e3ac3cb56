"""ctypes binding of the x265_amd C ABI (include/x265_amd.h) plus thin host-side helpers.

The library is the product: every compute call below runs HIP kernels on an MI355X.  If the
shared library or the GPU is missing this module raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .jobs import JOB_DTYPE, RESULT_DTYPE
from .synth import MARGIN

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("X265AMD_LIB", os.path.join(_HERE, "libx265amd.so"))

SAD, SATD, SA8D = 0, 1, 2

PAIR_DTYPE = np.dtype([("enc_plane", "<i4"), ("enc_x", "<i4"), ("enc_y", "<i4"),
                       ("ref_plane", "<i4"), ("ref_x", "<i4"), ("ref_y", "<i4")], align=True)

MC_PAIR_DTYPE = np.dtype([("enc_plane", "<i4"), ("pu_x", "<i4"), ("pu_y", "<i4"),
                          ("ref_plane", "<i4"), ("qmv_x", "<i4"), ("qmv_y", "<i4")], align=True)

PIXELCMP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_ssize_t, C.c_void_p, C.c_ssize_t)
PIXELCMP_X3 = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p)
PIXELCMP_X4 = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_void_p)


class Primitives(C.Structure):
    """x265amd_primitives: the slice of EncoderPrimitives this library fills."""
    _fields_ = [("sad", PIXELCMP * 64), ("sad_x3", PIXELCMP_X3 * 64), ("sad_x4", PIXELCMP_X4 * 64),
                ("satd", PIXELCMP * 64), ("sa8d_inter", PIXELCMP * 64), ("sa8d", PIXELCMP * 5)]


class X265AmdError(RuntimeError):
    pass


_lib = None


def load_library() -> C.CDLL:
    """Load libx265amd.so; raises if it has not been built (see x265_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise X265AmdError("%s is missing: build it with `python -m x265_amd.build` "
                           "(the HIP kernels are the only compute path)" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime.  A process must hold exactly one: when torch is
    # installed, load it first so that this library binds to the runtime torch uses (device
    # memory and streams are then shared with torch tensors).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ci, cip = C.c_void_p, C.c_int, C.POINTER(C.c_int)
    L.x265amd_create.argtypes = [ci, C.POINTER(vp)]
    L.x265amd_destroy.argtypes = [vp]
    L.x265amd_destroy.restype = None
    L.x265amd_last_error.argtypes = [vp]
    L.x265amd_last_error.restype = C.c_char_p
    L.x265amd_sync.argtypes = [vp]
    L.x265amd_stream.argtypes = [vp]
    L.x265amd_stream.restype = vp
    plane_args = [vp, vp, ci, ci, ci, ci, C.c_ssize_t, ci, ci, cip]
    L.x265amd_plane_upload.argtypes = plane_args
    L.x265amd_plane_wrap_device.argtypes = plane_args
    L.x265amd_debug_state.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.x265amd_wait_event.argtypes = [vp, vp]
    L.x265amd_record_event.argtypes = [vp, vp]
    L.x265amd_wait_event_on.argtypes = [vp, vp, ci]
    L.x265amd_record_event_on.argtypes = [vp, vp, ci]
    L.x265amd_plane_geometry.argtypes = [vp, ci, cip, cip, cip, cip, C.POINTER(C.c_ssize_t), cip, cip]
    L.x265amd_ipfilter.argtypes = [vp, ci, vp, C.c_ssize_t, vp, C.c_ssize_t, ci, ci, ci, ci, ci]
    L.x265amd_lookahead_status.argtypes = [vp]
    L.x265amd_field_me.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp]
    L.x265amd_field_me_batch.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp]
    L.x265amd_field_me_multi.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
    L.x265amd_field_me_multi_tmvp.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp]
    L.x265amd_field_results_device.argtypes = [vp]
    L.x265amd_field_results_device.restype = vp
    L.x265amd_plane_device_ptr.argtypes = [vp, ci]
    L.x265amd_plane_device_ptr.restype = vp
    L.x265amd_plane_release.argtypes = [vp, ci]
    L.x265amd_plane_create.argtypes = [vp, ci, ci, ci, ci, C.c_ssize_t, ci, ci, cip]
    L.x265amd_plane_update_rows.argtypes = [vp, ci, vp, C.c_ssize_t, ci, ci]
    L.x265amd_plane_download.argtypes = [vp, ci, ci, vp, C.c_size_t]
    L.x265amd_lowres_create.argtypes = [vp, ci, cip]
    L.x265amd_lowres_update.argtypes = [vp, ci, ci]
    L.x265amd_lowres_geometry.argtypes = [vp, ci, cip, cip, C.POINTER(C.c_ssize_t), cip, cip]
    L.x265amd_lookahead_me.argtypes = [vp, ci, cip, cip, ci, ci, ci, ci, vp]
    L.x265amd_lookahead_results_device.argtypes = [vp]
    L.x265amd_lookahead_results_device.restype = vp
    L.x265amd_cost_batch.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    L.x265amd_mc_cost_batch.argtypes = [vp, ci, ci, ci, ci, vp, vp]
    L.x265amd_bipred_source_batch.argtypes = [vp, ci, ci, ci, ci, vp]
    L.x265amd_me_search_batch.argtypes = [vp, ci, cip, ci, vp, ci, vp]
    L.x265amd_me_search_batch_device.argtypes = [vp, ci, cip, ci, vp, ci, vp]
    L.x265amd_me_search_batch_device_noalloc.argtypes = [vp, ci, cip, ci, vp, ci, vp]
    L.x265amd_joblist_create.argtypes = [vp, vp, ci, ci, ci, C.POINTER(vp)]
    L.x265amd_joblist_run.argtypes = [vp, vp, ci, cip, ci, vp]
    L.x265amd_joblist_run_overlapped.argtypes = [vp, vp, ci, cip, ci, ci]
    L.x265amd_joblist_results_device.argtypes = [vp]
    L.x265amd_joblist_results_device.restype = vp
    L.x265amd_joblist_results.argtypes = [vp, vp, vp]
    L.x265amd_joblist_results_async.argtypes = [vp, vp, vp]
    L.x265amd_joblist_order_by_field.argtypes = [vp, vp, vp, ci, ci]
    L.x265amd_joblist_eval_counts.argtypes = [vp, vp, vp]
    L.x265amd_joblist_order_by_length.argtypes = [vp, vp, vp]
    L.x265amd_plane_refill_async.argtypes = [vp, C.c_int, vp]
    L.x265amd_build_info.argtypes = []
    L.x265amd_build_info.restype = C.c_char_p
    L.x265amd_joblist_destroy.argtypes = [vp, vp]
    L.x265amd_joblist_rebalance.argtypes = [vp, vp]
    L.x265amd_eval_count.argtypes = [vp, ci, C.POINTER(C.c_uint64)]
    L.x265amd_set_timing.argtypes = [vp, ci]
    L.x265amd_kernel_times.argtypes = [vp, C.POINTER(C.c_float)]
    L.x265amd_mvcost_table.argtypes = [vp, ci, vp]
    L.x265amd_bitcost.argtypes = [vp, ci, ci, ci, ci]
    L.x265amd_setup_primitives.argtypes = [C.POINTER(Primitives), ci, ci]
    L.x265amd_partition_from_sizes.argtypes = [ci, ci]
    _lib = L
    return L


class Engine:
    """One x265amd context (one HIP stream on one device)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.x265amd_create(device, C.byref(h))
        if rc != 0:
            raise X265AmdError("x265amd_create(device=%d) failed with %d (no MI355X visible or kernels not loadable)" % (device, rc))
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.x265amd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise X265AmdError("x265amd error %d: %s" % (rc, self.lib.x265amd_last_error(self.h).decode()))

    @property
    def stream(self) -> int:
        return self.lib.x265amd_stream(self.h)

    def torch_stream(self):
        """The engine's HIP stream as a torch stream object (for GPU-side ordering against torch work)."""
        import torch
        if getattr(self, "_tstream", None) is None:
            self._tstream = torch.cuda.ExternalStream(self.stream, device=torch.device("cuda", self.device))
        return self._tstream

    def sync(self):
        self._chk(self.lib.x265amd_sync(self.h))

    def debug_state(self) -> str:
        buf = C.create_string_buffer(512)
        self.lib.x265amd_debug_state(self.h, buf, 512)
        return buf.value.decode()

    def wait_event(self, event, stream_set: int = 0):
        """work queued on the engine (on stream set `stream_set`) from now on starts after `event` (a torch.cuda.Event
        recorded on another stream)"""
        self._chk(self.lib.x265amd_wait_event_on(self.h, event.cuda_event, stream_set))

    def record_event(self, event, stream_set: int = 0):
        """record `event` (a torch.cuda.Event) behind everything queued on the engine's stream set so far"""
        if stream_set == 0:
            event.record(self.torch_stream())
            return
        if not event.cuda_event:            # (torch creates the HIP event at its first record)
            event.record(self.torch_stream())
        self._chk(self.lib.x265amd_record_event_on(self.h, event.cuda_event, stream_set))

    def plane_geometry(self, pid: int):
        """(width, height, margin_x, margin_y, stride, bytes_per_px, bit_depth)"""
        v = [C.c_int() for _ in range(6)]
        st = C.c_ssize_t()
        self._chk(self.lib.x265amd_plane_geometry(self.h, pid, C.byref(v[0]), C.byref(v[1]), C.byref(v[2]), C.byref(v[3]),
                                                  C.byref(st), C.byref(v[4]), C.byref(v[5])))
        return v[0].value, v[1].value, v[2].value, v[3].value, st.value, v[4].value, v[5].value

    def ipfilter(self, kind: int, src: np.ndarray, x0: int, y0: int, w: int, h: int, frac: int, depth: int,
                 bytes_per_px: int) -> np.ndarray:
        """one 8-tap luma filter call (kind 0 pp-H, 1 pp-V, 2 ps-H, 3 sp-V) on the w x h block at (x0, y0) of the
        2-D array `src` (pixels; int16 for kind 3); pixels are `bytes_per_px` wide"""
        assert src.flags["C_CONTIGUOUS"] and src.ndim == 2
        pix = np.uint8 if bytes_per_px == 1 else np.uint16
        assert src.dtype == (np.int16 if kind == 3 else pix)
        dst = np.zeros((h, w), dtype=np.int16 if kind == 2 else pix)
        ss = src.shape[1]
        self._chk(self.lib.x265amd_ipfilter(self.h, kind, src.ctypes.data + (y0 * ss + x0) * src.itemsize, ss,
                                            dst.ctypes.data, w, w, h, frac, bytes_per_px, depth))
        return dst

    def lookahead_status(self):
        self._chk(self.lib.x265amd_lookahead_status(self.h))

    # -- planes ------------------------------------------------------------------------------
    def upload_plane(self, padded: np.ndarray, bit_depth=None, margin: int = MARGIN) -> int:
        """padded: (H + 2*margin, W + 2*margin) uint8/uint16 array with replicated borders"""
        assert padded.flags["C_CONTIGUOUS"] and padded.dtype in (np.uint8, np.uint16)
        hh, ww = padded.shape
        pid = C.c_int(-1)
        depth = bit_depth if bit_depth is not None else (8 if padded.itemsize == 1 else 10)
        self._chk(self.lib.x265amd_plane_upload(self.h, padded.ctypes.data, ww - 2 * margin, hh - 2 * margin,
                                                margin, margin, ww, padded.itemsize, depth, C.byref(pid)))
        return pid.value

    def wrap_device_plane(self, dev_ptr: int, width: int, height: int, stride: int, bytes_per_px: int,
                          bit_depth: int, margin: int = MARGIN) -> int:
        pid = C.c_int(-1)
        self._chk(self.lib.x265amd_plane_wrap_device(self.h, dev_ptr, width, height, margin, margin, stride,
                                                     bytes_per_px, bit_depth, C.byref(pid)))
        return pid.value

    def create_plane(self, width: int, height: int, bytes_per_px: int = 1, bit_depth=None, margin: int = MARGIN) -> int:
        """an empty device plane of the usual geometry (stride = width + 2 * margin), to be filled row by row"""
        pid = C.c_int(-1)
        depth = bit_depth if bit_depth is not None else (8 if bytes_per_px == 1 else 10)
        self._chk(self.lib.x265amd_plane_create(self.h, width, height, margin, margin, width + 2 * margin, bytes_per_px,
                                                depth, C.byref(pid)))
        return pid.value

    def update_plane_rows(self, pid: int, picture: np.ndarray, y0: int, n_rows: int):
        """copy rows [y0, y0 + n_rows) of an UNPADDED picture array into the plane and extend their borders"""
        assert picture.flags["C_CONTIGUOUS"]
        self._chk(self.lib.x265amd_plane_update_rows(self.h, pid, picture.ctypes.data, picture.shape[1], y0, n_rows))

    def refill_plane_async(self, pid: int, host_ptr: int):
        """queue the upload of a whole padded plane (page-locked host memory at `host_ptr`) into plane `pid`; no wait"""
        self._chk(self.lib.x265amd_plane_refill_async(self.h, pid, host_ptr))

    def release_plane(self, pid: int):
        self._chk(self.lib.x265amd_plane_release(self.h, pid))

    # -- lookahead: half-resolution planes and whole-frame search ------------------------------
    def lowres_create(self, src_plane: int) -> int:
        """half-resolution F/H/V/C plane set of a registered full-resolution plane; returns its plane id"""
        pid = C.c_int(-1)
        self._chk(self.lib.x265amd_lowres_create(self.h, src_plane, C.byref(pid)))
        return pid.value

    def lowres_update(self, lowres_plane: int, src_plane: int):
        self._chk(self.lib.x265amd_lowres_update(self.h, lowres_plane, src_plane))

    def lowres_geometry(self, lowres_plane: int):
        """(width, lines, stride, cu_w, cu_h)"""
        w, l, cw, ch = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        st = C.c_ssize_t()
        self._chk(self.lib.x265amd_lowres_geometry(self.h, lowres_plane, C.byref(w), C.byref(l), C.byref(st),
                                                   C.byref(cw), C.byref(ch)))
        return w.value, l.value, st.value, cw.value, ch.value

    def download_plane(self, plane: int, shape, dtype, which: int = 0) -> np.ndarray:
        """the padded allocation of a plane (or plane `which` of a half-resolution set) as an array"""
        out = np.zeros(shape, dtype=dtype)
        self._chk(self.lib.x265amd_plane_download(self.h, plane, which, out.ctypes.data, out.nbytes))
        return out

    def lookahead_me(self, enc_lowres, ref_lowres, merange: int = 16, method: int = 1, subme: int = 1, qp: int = 12,
                     fetch: bool = True):
        """whole-frame lookahead searches of (enc, ref) half-resolution pairs; returns
        results[n_pairs, cu_h * cu_w] (raster order), or None with fetch=False (asynchronous)"""
        enc_lowres, ref_lowres = list(enc_lowres), list(ref_lowres)
        n = len(enc_lowres)
        assert n == len(ref_lowres) and n > 0
        _, _, _, cw, ch = self.lowres_geometry(enc_lowres[0])
        e = (C.c_int * n)(*enc_lowres)
        r = (C.c_int * n)(*ref_lowres)
        out = np.zeros((n, cw * ch), dtype=RESULT_DTYPE) if fetch else None
        self._chk(self.lib.x265amd_lookahead_me(self.h, n, e, r, merange, method, subme, qp,
                                                out.ctypes.data if fetch else None))
        return out

    def field_me(self, enc_plane: int, ref_plane: int, pu_w: int, pu_h: int, merange: int, method: int, subme: int, qp: int,
                 fetch: bool = True):
        """whole-frame field of pu_w x pu_h PUs with the neighbour-derived predictor chain; returns
        (results[gh, gw], mvp[gh, gw, 2]) or None with fetch=False (asynchronous)"""
        w, h = self.plane_geometry(enc_plane)[:2]
        gw, gh = w // pu_w, h // pu_h
        out = np.zeros((gh, gw), dtype=RESULT_DTYPE) if fetch else None
        mvp = np.zeros((gh, gw, 2), dtype=np.int16) if fetch else None
        self._chk(self.lib.x265amd_field_me(self.h, enc_plane, ref_plane, pu_w, pu_h, merange, method, subme, qp,
                                            out.ctypes.data if fetch else None, mvp.ctypes.data if fetch else None))
        return (out, mvp) if fetch else None

    def field_me_batch(self, enc_planes, ref_planes, pu_w: int, pu_h: int, merange: int, method: int, subme: int, qp: int,
                       fetch: bool = True):
        """the same for several (enc, ref) plane pairs of one picture size in one launch; returns
        (results[n, gh, gw], mvp[n, gh, gw, 2]) or None with fetch=False"""
        enc = np.ascontiguousarray(enc_planes, dtype=np.int32)
        ref = np.ascontiguousarray(ref_planes, dtype=np.int32)
        assert len(enc) == len(ref) and len(enc) > 0
        w, h = self.plane_geometry(int(enc[0]))[:2]
        gw, gh = w // pu_w, h // pu_h
        out = np.zeros((len(enc), gh, gw), dtype=RESULT_DTYPE) if fetch else None
        mvp = np.zeros((len(enc), gh, gw, 2), dtype=np.int16) if fetch else None
        self._chk(self.lib.x265amd_field_me_batch(self.h, len(enc), enc.ctypes.data, ref.ctypes.data, pu_w, pu_h, merange, method,
                                                  subme, qp, out.ctypes.data if fetch else None, mvp.ctypes.data if fetch else None))
        return (out, mvp) if fetch else None

    @staticmethod
    def field_grid(w: int, h: int, pu_w: int, pu_h: int, part_mode: int = 0):
        """(PUs per row, PU rows) of a field: whole CUs only (part_mode 1 = 2NxN: CUs of pu_w x 2 pu_h, 2 = Nx2N: 2 pu_w x pu_h;
        4 / 5 = 2NxnU / 2NxnD and 6 / 7 = nLx2N / nRx2N: CUs of pu_w x pu_h, two PU rows / two PU columns each)"""
        gw, gh = w // pu_w, h // pu_h
        if part_mode in (4, 5):
            return gw, 2 * gh
        if part_mode in (6, 7):
            return 2 * gw, gh
        return (gw & ~1 if part_mode == 2 else gw), (gh & ~1 if part_mode == 1 else gh)

    def field_me_multi(self, enc_planes, ref_planes, pu_w, pu_h, part_mode, merange: int, method: int, subme: int, qp: int,
                       fetch: bool = True, col=None):
        """fields of DIFFERENT PU sizes / partition modes in one call (one per quadtree depth, say), run side by side;
        returns a list of (results[gh, gw], mvp[gh, gw, 2]) per field, or None with fetch=False.
        col: the co-located motion for the temporal AMVP candidate, RESULT_DTYPE[n_fields, 16x16 blocks of the picture in raster
        order], scaled already, cost < 0 = none (x265amd_field_me_multi_tmvp)"""
        enc = np.ascontiguousarray(enc_planes, dtype=np.int32)
        ref = np.ascontiguousarray(ref_planes, dtype=np.int32)
        pw = np.ascontiguousarray(pu_w, dtype=np.int32)
        ph = np.ascontiguousarray(pu_h, dtype=np.int32)
        pm = np.ascontiguousarray(part_mode, dtype=np.int32)
        assert len(enc) == len(ref) == len(pw) == len(ph) == len(pm) and len(enc) > 0
        w, h = self.plane_geometry(int(enc[0]))[:2]
        grids = [self.field_grid(w, h, int(a), int(b), int(m)) for a, b, m in zip(pw, ph, pm)]
        n = sum(gw * gh for gw, gh in grids)
        out = np.zeros(n, dtype=RESULT_DTYPE) if fetch else None
        mvp = np.zeros((n, 2), dtype=np.int16) if fetch else None
        if col is not None:
            col = np.ascontiguousarray(col, dtype=RESULT_DTYPE)
            if col.size != len(enc) * ((w + 15) // 16) * ((h + 15) // 16):
                raise ValueError("col needs one entry per 16x16 block of the picture and field")
        self._chk(self.lib.x265amd_field_me_multi_tmvp(self.h, len(enc), enc.ctypes.data, ref.ctypes.data, pw.ctypes.data, ph.ctypes.data,
                                                       pm.ctypes.data, col.ctypes.data if col is not None else None,
                                                       merange, method, subme, qp,
                                                       out.ctypes.data if fetch else None, mvp.ctypes.data if fetch else None))
        if not fetch:
            return None
        res, at = [], 0
        for gw, gh in grids:
            res.append((out[at:at + gw * gh].reshape(gh, gw), mvp[at:at + gw * gh].reshape(gh, gw, 2)))
            at += gw * gh
        return res

    # -- costs --------------------------------------------------------------------------------
    def cost_batch(self, kind: int, w: int, h: int, pairs: np.ndarray) -> np.ndarray:
        pairs = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
        out = np.zeros(len(pairs), dtype=np.int32)
        self._chk(self.lib.x265amd_cost_batch(self.h, kind, w, h, len(pairs), pairs.ctypes.data, out.ctypes.data))
        return out

    def mc_cost_batch(self, kind: int, w: int, h: int, pairs: np.ndarray) -> np.ndarray:
        """costs of PUs against motion-compensated (interpolated) blocks at q-pel MVs (MC_PAIR_DTYPE)"""
        pairs = np.ascontiguousarray(pairs, dtype=MC_PAIR_DTYPE)
        out = np.zeros(len(pairs), dtype=np.int32)
        self._chk(self.lib.x265amd_mc_cost_batch(self.h, kind, w, h, len(pairs), pairs.ctypes.data, out.ctypes.data))
        return out

    # -- search -------------------------------------------------------------------------------
    def bipred_source_batch(self, dst_plane: int, w: int, h: int, pairs: np.ndarray):
        """dst_plane[PU] = clip(2 * enc[PU] - motion-compensated prediction of the PU from ref at the q-pel MV); asynchronous"""
        pairs = np.ascontiguousarray(pairs, dtype=MC_PAIR_DTYPE)
        self._chk(self.lib.x265amd_bipred_source_batch(self.h, dst_plane, w, h, len(pairs), pairs.ctypes.data))

    def search(self, enc_plane: int, ref_planes, jobs: np.ndarray) -> np.ndarray:
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        refs = (C.c_int * len(ref_planes))(*ref_planes)
        out = np.zeros(len(jobs), dtype=RESULT_DTYPE)
        self._chk(self.lib.x265amd_me_search_batch(self.h, enc_plane, refs, len(ref_planes), jobs.ctypes.data,
                                                   len(jobs), out.ctypes.data))
        return out

    def search_device(self, enc_plane: int, ref_planes, d_jobs_ptr: int, n: int, d_out_ptr: int):
        """asynchronous: jobs/results are device pointers; call sync() before reading results"""
        refs = (C.c_int * len(ref_planes))(*ref_planes)
        self._chk(self.lib.x265amd_me_search_batch_device(self.h, enc_plane, refs, len(ref_planes), d_jobs_ptr, n, d_out_ptr))

    def search_device_noalloc(self, enc_plane: int, ref_planes, d_jobs_ptr: int, n: int, d_out_ptr: int):
        """search_device for callers that must not allocate: fixed scratch inside the library, n <= 65536"""
        refs = (C.c_int * len(ref_planes))(*ref_planes)
        self._chk(self.lib.x265amd_me_search_batch_device_noalloc(self.h, enc_plane, refs, len(ref_planes), d_jobs_ptr, n, d_out_ptr))

    def make_joblist(self, jobs: np.ndarray, n_refs: int = 1, bytes_per_px: int = 1) -> "JobList":
        return JobList(self, jobs, n_refs, bytes_per_px)

    def eval_count(self, reset: bool = False) -> int:
        v = C.c_uint64(0)
        self._chk(self.lib.x265amd_eval_count(self.h, int(reset), C.byref(v)))
        return v.value

    def set_timing(self, enable: bool = True):
        self._chk(self.lib.x265amd_set_timing(self.h, int(enable)))

    def kernel_times(self):
        """(ms of the exhaustive kernel, ms of the general search kernel) of the last batch"""
        ms = (C.c_float * 2)()
        self._chk(self.lib.x265amd_kernel_times(self.h, ms))
        return float(ms[0]), float(ms[1])

    def mvcost_table(self, qp: int) -> np.ndarray:
        out = np.zeros(65536, dtype=np.uint16)
        self._chk(self.lib.x265amd_mvcost_table(self.h, qp, out.ctypes.data))
        return out

    def bitcost(self, mvp, mv) -> int:
        return self.lib.x265amd_bitcost(self.h, mvp[0], mvp[1], mv[0], mv[1])


class JobList:
    """x265amd_joblist: jobs resident in device memory + the exhaustive kernel's work-item plan."""

    def __init__(self, eng: Engine, jobs: np.ndarray, n_refs: int, bytes_per_px: int):
        self.eng = eng
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        self.n = len(jobs)
        h = C.c_void_p()
        eng._chk(eng.lib.x265amd_joblist_create(eng.h, jobs.ctypes.data, self.n, n_refs, bytes_per_px, C.byref(h)))
        self.h = h

    def run(self, enc_plane: int, ref_planes, fetch: bool = True):
        """fetch=True: blocks and returns the results; False: asynchronous, results stay on the device"""
        refs = (C.c_int * len(ref_planes))(*ref_planes)
        out = np.zeros(self.n, dtype=RESULT_DTYPE) if fetch else None
        self.eng._chk(self.eng.lib.x265amd_joblist_run(self.eng.h, self.h, enc_plane, refs, len(ref_planes),
                                                       out.ctypes.data if fetch else None))
        return out

    def run_overlapped(self, enc_plane: int, ref_planes, stream_set: int):
        """asynchronous run on stream set 0 or 1 of the engine: lists alternating between the sets keep two frames in flight"""
        refs = (C.c_int * len(ref_planes))(*ref_planes)
        self.eng._chk(self.eng.lib.x265amd_joblist_run_overlapped(self.eng.h, self.h, enc_plane, refs, len(ref_planes), stream_set))

    def results(self) -> np.ndarray:
        """the results of the list's last run (waits for it): for runs queued with fetch=False"""
        out = np.zeros(self.n, dtype=RESULT_DTYPE)
        self.eng._chk(self.eng.lib.x265amd_joblist_results(self.eng.h, self.h, out.ctypes.data))
        return out

    def results_async(self, out_ptr: int):
        """queue the copy of the last run's results to page-locked host memory at `out_ptr` (n * 8 bytes); no wait"""
        self.eng._chk(self.eng.lib.x265amd_joblist_results_async(self.eng.h, self.h, out_ptr))

    def order_by_field(self, field: np.ndarray, cu_w: int, cu_h: int):
        """start first the jobs the lookahead's half-resolution field (cu_w x cu_h results) predicts to be long"""
        field = np.ascontiguousarray(field, dtype=RESULT_DTYPE)
        assert field.size == cu_w * cu_h
        self.eng._chk(self.eng.lib.x265amd_joblist_order_by_field(self.eng.h, self.h, field.ctypes.data, cu_w, cu_h))

    def eval_counts(self) -> np.ndarray:
        """block-cost evaluations per job of the last run"""
        out = np.zeros(self.n, dtype=np.uint32)
        self.eng._chk(self.eng.lib.x265amd_joblist_eval_counts(self.eng.h, self.h, out.ctypes.data))
        return out

    def order_by_length(self, length: np.ndarray):
        """start first the jobs with the largest predicted length (one value per job)"""
        length = np.ascontiguousarray(length, dtype=np.uint32)
        assert length.size == self.n
        self.eng._chk(self.eng.lib.x265amd_joblist_order_by_length(self.eng.h, self.h, length.ctypes.data))

    def rebalance(self):
        """start the jobs that were longest in the last run first in the following runs"""
        self.eng._chk(self.eng.lib.x265amd_joblist_rebalance(self.eng.h, self.h))

    @property
    def results_device_ptr(self) -> int:
        return self.eng.lib.x265amd_joblist_results_device(self.h)

    def close(self):
        if self.h and self.eng.h:
            self.eng.lib.x265amd_joblist_destroy(self.eng.h, self.h)
        self.h = None


def build_info() -> str:
    """x265amd_build_info: names the sources the loaded library was built from"""
    return load_library().x265amd_build_info().decode()


def setup_primitives(device: int = 0, bytes_per_px: int = 1) -> Primitives:
    """x265amd_setup_primitives: table of per-call primitives with the reference's signatures."""
    lib = load_library()
    p = Primitives()
    rc = lib.x265amd_setup_primitives(C.byref(p), device, bytes_per_px)
    if rc != 0:
        raise X265AmdError("x265amd_setup_primitives failed with %d" % rc)
    return p
