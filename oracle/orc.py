"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. Nothing under hybvio_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def build(force: bool = False) -> str:
    so = os.path.join(_DIR, "liboracle.so")
    srcs = [os.path.join(_DIR, f) for f in ("pyrlk_oracle.c", "ekf_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _DIR, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_pyramid_build.restype = C.c_void_p
        L.orc_pyramid_build.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_pyramid_free.argtypes = [C.c_void_p]
        L.orc_pyramid_levels.argtypes = [C.c_void_p]
        L.orc_pyramid_level_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_pyramid_copy_gray.argtypes = [C.c_void_p, C.c_int, C.c_int, u8p]
        L.orc_pyramid_copy_deriv.argtypes = [C.c_void_p, C.c_int, C.c_int, i16p]
        L.orc_klt_track.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f32p, f32p, u8p, f32p, C.c_int, C.c_int,
                                    C.c_int, C.c_double, C.c_int, C.c_double, i32p]
        L.orc_optical_flow_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f32p, f32p, i32p, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


class Pyramid:
    """cv::buildOpticalFlowPyramid restatement (padded OpenCV layout)."""

    def __init__(self, img: np.ndarray, win: int = 31, max_level: int = 3):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        self.win = win
        self._h = lib().orc_pyramid_build(_p(img, u8p), w, h, w, win, max_level)
        if not self._h:
            raise RuntimeError("orc_pyramid_build failed")
        self.levels = lib().orc_pyramid_levels(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_pyramid_free(self._h)
            self._h = None

    def size(self, level: int):
        w, h = C.c_int(), C.c_int()
        assert lib().orc_pyramid_level_size(self._h, level, C.byref(w), C.byref(h)) == 0
        return w.value, h.value

    def gray(self, level: int, padded: bool = False) -> np.ndarray:
        w, h = self.size(level)
        p = self.win if padded else 0
        out = np.empty((h + 2 * p, w + 2 * p), np.uint8)
        assert lib().orc_pyramid_copy_gray(self._h, level, int(padded), _p(out, u8p)) == 0
        return out

    def deriv(self, level: int, padded: bool = False) -> np.ndarray:
        w, h = self.size(level)
        p = self.win if padded else 0
        out = np.empty((h + 2 * p, w + 2 * p, 2), np.int16)
        assert lib().orc_pyramid_copy_deriv(self._h, level, int(padded), _p(out, i16p)) == 0
        return out


USE_INITIAL_FLOW = 4


def klt_track(prev: Pyramid, nxt: Pyramid, prev_pts, next_pts=None, win=31, max_level=3, max_count=20,
              eps=0.03, min_eig=1e-3, want_iters=False):
    """cv::calcOpticalFlowPyrLK restatement. Returns (next_pts, status u8, err f32[, iters])."""
    prev_pts = np.ascontiguousarray(prev_pts, np.float32).reshape(-1, 2)
    n = prev_pts.shape[0]
    flags = 0
    if next_pts is not None:
        flags = USE_INITIAL_FLOW
        out = np.ascontiguousarray(next_pts, np.float32).reshape(-1, 2).copy()
    else:
        out = np.zeros_like(prev_pts)
    status = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    iters = np.zeros((max_level + 1, n), np.int32) if want_iters else None
    rc = lib().orc_klt_track(prev._h, nxt._h, n, _p(prev_pts, f32p), _p(out, f32p), _p(status, u8p),
                             _p(err, f32p), win, max_level, max_count, eps, flags, min_eig,
                             _p(iters, i32p) if want_iters else None)
    assert rc == 0
    return (out, status, err, iters) if want_iters else (out, status, err)


def optical_flow_compute(prev: Pyramid, cur: Pyramid, prev_corners, corners=None, win=31, max_level=3,
                         max_iter=20, eps=0.03, min_eig=1e-3):
    """tracker::OpticalFlow::compute restatement (optical_flow.cpp:10-59).
    Returns (corners, Feature::Status int32)."""
    prev_corners = np.ascontiguousarray(prev_corners, np.float32).reshape(-1, 2)
    n = prev_corners.shape[0]
    use_init = corners is not None
    out = (np.ascontiguousarray(corners, np.float32).reshape(-1, 2).copy() if use_init
           else np.zeros_like(prev_corners))
    st = np.full(n, 2, np.int32)
    rc = lib().orc_optical_flow_compute(prev._h, cur._h, n, _p(prev_corners, f32p), _p(out, f32p),
                                        _p(st, i32p), int(use_init), win, max_level, max_iter, eps, min_eig)
    assert rc == 0
    return out, st
