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
    if os.environ.get("ORC_NATIVE") == "1":
        # -O3 -march=native build for bench.py's second CPU column: compiled on the machine that runs it, outside the tree
        import tempfile
        so = os.path.join(tempfile.gettempdir(), f"liboracle_native_{os.getuid()}.so")
        subprocess.check_call(["make", "-C", _DIR, "-B", "native", f"NATIVE_OUT={so}"], stdout=subprocess.DEVNULL)
        return so
    so = os.path.join(_DIR, "liboracle.so")
    srcs = [os.path.join(_DIR, f) for f in ("pyrlk_oracle.c", "ekf_oracle.c", "gftt_oracle.c", "ingest_oracle.c", "triangulation_oracle.c", "rot_ransac_oracle.c", "Makefile")]
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
        L.orc_klt_track_mode.argtypes = L.orc_klt_track.argtypes + [C.c_int]
        L.orc_lk_acc_A.argtypes = [C.c_int, i16p, C.c_int, f32p]
        L.orc_lk_acc_b.argtypes = [C.c_int, i32p, i16p, C.c_int, f32p]
        L.orc_optical_flow_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f32p, f32p, i32p, C.c_int,
                                               C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(min(8, os.cpu_count() or 1))     # a 256-thread team on 480-row loops is far slower than 1 thread
        L.orc_corner_min_eigen_val.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_gftt_block_size.argtypes = [C.c_double]
        L.orc_gftt_collect_max.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, f32p]
        L.orc_apply_min_distance.argtypes = [f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int]
        L.orc_gftt_detect.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_float, f32p, C.c_int,
                                      C.c_int, C.c_int, f32p, C.c_int]
        L.orc_camera_create.restype = C.c_void_p
        L.orc_camera_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, f64p, f64p, C.c_double]
        L.orc_camera_destroy.argtypes = [C.c_void_p]
        L.orc_camera_pixel_to_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, f64p]
        L.orc_camera_ray_to_pixel.argtypes = [C.c_void_p, f64p, f64p]
        L.orc_undistort_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, f64p, u8p]
        L.orc_undistort_apply.argtypes = [u8p, C.c_int, C.c_int, f64p, u8p, u8p]
        L.orc_color_to_gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_tri_default_params.argtypes = [C.c_void_p]
        L.orc_extract_camera_pose_trail.argtypes = [f64p, i32p, C.c_int, f64p, f64p, C.c_void_p]
        L.orc_inverse_depth.argtypes = [f64p, f64p, f64p]
        L.orc_pinv32.argtypes = [f64p, f64p]
        L.orc_triangulate_with_two_cameras.argtypes = [C.c_void_p, C.c_void_p, f64p, f64p, f64p, f64p, C.c_int, C.c_int, C.c_int,
                                                       C.c_double, f64p, f64p]
        L.orc_triangulate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double,
                                      f64p, f64p, f64p, f64p]
        L.orc_prepare_visual_update.argtypes = [f64p, f64p, f64p, f64p, f64p, C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, C.c_int,
                                                C.c_int, C.c_int, C.c_int, C.c_double, f64p, f64p, i32p]
        L.orc_tri_last_diag.argtypes = [f64p]
        L.orc_visual_track_prepare.argtypes = [C.c_void_p, f64p, C.c_int, i32p, C.c_int, f64p, f64p, f64p, f64p, f64p, f64p, f64p, i32p]
        u32p = C.POINTER(C.c_uint32)
        L.orc_mt19937_draws.argtypes = [C.c_uint32, C.c_int, C.c_int, u32p]
        L.orc_kabsch_rotation.argtypes = [f32p, f32p]
        L.orc_solve_rotation.argtypes = [f32p, f32p, i32p, C.c_int, f32p]
        L.orc_rot_ransac_fit.argtypes = [f32p, f32p, C.c_int, C.c_void_p, C.c_void_p, u32p, C.c_float, i32p, f32p, i32p]
        _LIB = L
    return _LIB


def set_threads(n: int) -> int:
    """Threads for the row (pyramid, Scharr) and point (LK) loops: 1 = scalar port, 0 = all host cores.
    Returns the count in effect. Results do not depend on it (integer sums, independent points)."""
    return int(lib().orc_set_threads(int(n)))


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


# accumulator modes of the five LK sums (oracle/pyrlk_oracle.c header): INT64 is the kernel's; the F32 ones restate what
# OpenCV builds off-ARM compute and exist to MEASURE the sensitivity (scripts/lk_accumulator_study.py)
ACC_MODES = {"int64": 0, "f32_scalar": 1, "f32_simd128": 2, "f32_simd128_fma": 3, "f32_sse2_legacy": 4, "f32_wide8": 5}


def lk_acc_A(mode: str, dI, win=31):
    """(A11, A12, A22) float sums of an int16 [win][win][2] derivative window in the order of an F32 mode (unscaled)."""
    dI = np.ascontiguousarray(dI, np.int16)
    out = np.zeros(3, np.float32)
    lib().orc_lk_acc_A(ACC_MODES[mode], _p(dI, i16p), win, _p(out, f32p))
    return out


def lk_acc_b(mode: str, diff, dI, win=31):
    diff, dI = np.ascontiguousarray(diff, np.int32), np.ascontiguousarray(dI, np.int16)
    out = np.zeros(2, np.float32)
    lib().orc_lk_acc_b(ACC_MODES[mode], _p(diff, i32p), _p(dI, i16p), win, _p(out, f32p))
    return out


def klt_track(prev: Pyramid, nxt: Pyramid, prev_pts, next_pts=None, win=31, max_level=3, max_count=20,
              eps=0.03, min_eig=1e-3, want_iters=False, acc_mode="int64"):
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
    rc = lib().orc_klt_track_mode(prev._h, nxt._h, n, _p(prev_pts, f32p), _p(out, f32p), _p(status, u8p),
                                  _p(err, f32p), win, max_level, max_count, eps, flags, min_eig,
                                  _p(iters, i32p) if want_iters else None, ACC_MODES[acc_mode])
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


# ---------------------------------------------------------------------------------------------
# EKF oracle (oracle/ekf_oracle.c)
# ---------------------------------------------------------------------------------------------

class EkfParams(C.Structure):
    _fields_ = [("cameraTrailLength", C.c_int), ("hybridMapSize", C.c_int)] + [
        (k, C.c_double) for k in (
            "noiseScale", "gravity", "augmentR", "initZuptR", "rotationZuptR",
            "noiseInitialPos", "noiseInitialOri", "noiseInitialVel", "noiseInitialPosTrail", "noiseInitialOriTrail",
            "noiseInitialBGA", "noiseInitialBAA", "noiseInitialBAT", "noiseInitialSFT",
            "noiseProcessAcc", "noiseProcessGyro", "noiseProcessBAA", "noiseProcessBGA",
            "noiseProcessBAARev", "noiseProcessBGARev")]


_EKF_READY = False


def _ekf_lib():
    global _EKF_READY
    L = lib()
    if not _EKF_READY:
        vp, d, i = C.c_void_p, C.c_double, C.c_int
        sig = {
            "orc_ekf_default_params": (None, [C.POINTER(EkfParams)]),
            "orc_ekf_create": (vp, [C.POINTER(EkfParams)]), "orc_ekf_clone": (vp, [vp]), "orc_ekf_free": (None, [vp]),
            "orc_ekf_state_dim": (i, [vp]), "orc_ekf_state": (f64p, [vp]), "orc_ekf_cov": (f64p, [vp]),
            "orc_ekf_process_noise": (f64p, [vp]), "orc_ekf_dydx": (f64p, [vp]),
            "orc_ekf_platform_time": (d, [vp]), "orc_ekf_pose_count": (i, [vp]), "orc_ekf_was_stationary": (i, [vp]),
            "orc_ekf_history_time": (d, [vp, i]), "orc_ekf_set_first_sample_time": (None, [vp, d]),
            "orc_ekf_normalize_quaternions": (None, [vp, i]), "orc_ekf_maintain_psd": (None, [vp]),
            "orc_ekf_initialize_orientation": (None, [vp, f64p]), "orc_ekf_predict": (None, [vp, d, f64p, f64p]),
            "orc_ekf_update_zupt": (None, [vp, d]), "orc_ekf_update_zupt_initialization": (None, [vp]),
            "orc_ekf_update_zrupt": (None, [vp, f64p]), "orc_ekf_update_pseudo_velocity": (None, [vp, d, d]),
            "orc_ekf_update_position": (None, [vp, f64p, d]), "orc_ekf_update_zero_height": (None, [vp, d]),
            "orc_ekf_update_orientation": (None, [vp, f64p, d]),
            "orc_ekf_get_inertial_state": (None, [vp, f64p, f64p]), "orc_ekf_set_inertial_state": (None, [vp, f64p, f64p]),
            "orc_ekf_translate_to": (None, [vp, f64p]), "orc_ekf_transform_to": (None, [vp, f64p, f64p, i]),
            "orc_ekf_visual_track_outlier_check": (i, [vp, i, i, f64p, f64p, f64p, d, d, f64p]),
            "orc_ekf_update_visual_track": (None, [vp, i, i, f64p, f64p, f64p, d]),
            "orc_ekf_update_visual_pose_augmentation": (None, [vp, i]), "orc_ekf_update_undo_augmentation": (None, [vp]),
            "orc_ekf_condition_on_last_pose": (None, [vp]), "orc_ekf_lock_biases": (None, [vp]),
            "orc_ekf_map_point_state_index": (i, [vp, i]), "orc_ekf_insert_map_point": (None, [vp, i, f64p]),
            "orc_ldlt_quadratic_form": (d, [i, f64p, f64p]),
            "orc_quat2rmat": (None, [f64p, f64p]), "orc_quat2rmat_d": (None, [f64p, f64p, f64p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _EKF_READY = True
    return L


def _d(a):
    return np.ascontiguousarray(a, np.float64)


def ekf_default_params(**over):
    p = EkfParams()
    _ekf_lib().orc_ekf_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def ldlt_quadratic_form(M, v):
    M, v = np.asfortranarray(M, np.float64), _d(v)
    return _ekf_lib().orc_ldlt_quadratic_form(len(v), M.ctypes.data_as(f64p), _p(v, f64p))


def quat2rmat_d(q):
    q = _d(q)
    R, dR = np.zeros(9), np.zeros(36)
    _ekf_lib().orc_quat2rmat_d(_p(q, f64p), _p(R, f64p), _p(dR, f64p))
    return R.reshape(3, 3).T.copy(), [dR[9 * k:9 * k + 9].reshape(3, 3).T.copy() for k in range(4)]


class _Owned(np.ndarray):
    """ndarray view into C-owned storage that keeps its owner alive."""
    _owner = None


def _view(ptr, shape, owner):
    a = np.ctypeslib.as_array(ptr, shape).view(_Owned)
    a._owner = owner
    return a


class Ekf:
    """odometry::EKF restatement (src/odometry/ekf.hpp:62-174). Matrices are numpy (row, col)."""
    INLIER, NOT_COMPUTED, RMSE, CHI2 = 0, 1, 2, 3

    def __init__(self, params=None, _handle=None):
        L = _ekf_lib()
        self.params = params if params is not None else ekf_default_params()
        self._h = _handle if _handle is not None else L.orc_ekf_create(C.byref(self.params))
        self.n = L.orc_ekf_state_dim(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            _ekf_lib().orc_ekf_free(self._h)
            self._h = None

    def clone(self):
        return Ekf(self.params, _ekf_lib().orc_ekf_clone(self._h))

    # views into the C object's storage
    @property
    def m(self):
        return _view(_ekf_lib().orc_ekf_state(self._h), (self.n,), self)

    @property
    def P(self):   # column-major storage exposed as an (n, n) array with P[i, j] semantics
        return _view(_ekf_lib().orc_ekf_cov(self._h), (self.n, self.n), self).T

    @property
    def Q(self):
        return _view(_ekf_lib().orc_ekf_process_noise(self._h), (12, 12), self).T

    @property
    def dydx(self):
        return _view(_ekf_lib().orc_ekf_dydx(self._h), (20, 20), self).T

    def set_state(self, m):
        self.m[:] = m

    def set_cov(self, P):
        self.P[:, :] = P

    def __getattr__(self, name):
        # thin pass-through for the scalar-argument methods
        simple = {"update_zupt", "update_zupt_initialization", "update_zero_height", "update_pseudo_velocity",
                  "update_visual_pose_augmentation", "update_undo_augmentation", "condition_on_last_pose",
                  "lock_biases", "maintain_psd", "normalize_quaternions", "set_first_sample_time",
                  "platform_time", "pose_count", "was_stationary", "history_time", "map_point_state_index"}
        if name in simple:
            fn = getattr(_ekf_lib(), "orc_ekf_" + name)
            return lambda *a: fn(self._h, *a)
        raise AttributeError(name)

    def initialize_orientation(self, xa):
        _ekf_lib().orc_ekf_initialize_orientation(self._h, _p(_d(xa), f64p))

    def predict(self, t, xg, xa):
        _ekf_lib().orc_ekf_predict(self._h, t, _p(_d(xg), f64p), _p(_d(xa), f64p))

    def update_zrupt(self, xg):
        _ekf_lib().orc_ekf_update_zrupt(self._h, _p(_d(xg), f64p))

    def update_position(self, pos, r):
        _ekf_lib().orc_ekf_update_position(self._h, _p(_d(pos), f64p), r)

    def update_orientation(self, q, r):
        _ekf_lib().orc_ekf_update_orientation(self._h, _p(_d(q), f64p), r)

    def get_inertial_state(self):
        mean, cov = np.zeros(20), np.zeros(400)
        _ekf_lib().orc_ekf_get_inertial_state(self._h, _p(mean, f64p), _p(cov, f64p))
        return mean, cov.reshape(20, 20).T.copy()

    def set_inertial_state(self, mean, cov):
        _ekf_lib().orc_ekf_set_inertial_state(self._h, _p(_d(mean), f64p), np.asfortranarray(cov, np.float64).ctypes.data_as(f64p))

    def translate_to(self, pos):
        _ekf_lib().orc_ekf_translate_to(self._h, _p(_d(pos), f64p))

    def transform_to(self, pos, q, i=-1):
        _ekf_lib().orc_ekf_transform_to(self._h, _p(_d(pos), f64p), _p(_d(q), f64p), i)

    def visual_track_outlier_check(self, H, f, y, r, rmse_threshold=-1.0):
        Hf = np.asfortranarray(H, np.float64)
        chi2 = C.c_double()
        st = _ekf_lib().orc_ekf_visual_track_outlier_check(self._h, Hf.shape[0], Hf.shape[1], Hf.ctypes.data_as(f64p),
                                                           _p(_d(f), f64p), _p(_d(y), f64p), r, rmse_threshold, C.byref(chi2))
        return st, chi2.value

    def update_visual_track(self, H, f, y, r):
        Hf = np.asfortranarray(H, np.float64)
        _ekf_lib().orc_ekf_update_visual_track(self._h, Hf.shape[0], Hf.shape[1], Hf.ctypes.data_as(f64p),
                                               _p(_d(f), f64p), _p(_d(y), f64p), r)

    def insert_map_point(self, idx, pf):
        _ekf_lib().orc_ekf_insert_map_point(self._h, idx, _p(_d(pf), f64p))


# ---- GFTT feature detector (oracle/gftt_oracle.c) ----
def corner_min_eigen_val(img: np.ndarray, block_size: int = 3) -> np.ndarray:
    """cv::cornerMinEigenVal(img, blockSize, ksize=3) restatement, float32 response."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    rc = lib().orc_corner_min_eigen_val(_p(img, u8p), w, h, w, block_size, _p(out, f32p))
    if rc != 0:
        raise RuntimeError(f"orc_corner_min_eigen_val failed: {rc}")
    return out


def gftt_block_size(min_distance: float = 50.0) -> int:
    return int(lib().orc_gftt_block_size(float(min_distance)))


def gftt_collect_max(response: np.ndarray, bs: int, min_response: float = 1e-3) -> np.ndarray:
    """CollectMax::cpuImplementation: one (x, y, response) row per bs x bs block."""
    response = np.ascontiguousarray(response, np.float32)
    h, w = response.shape
    kp = np.zeros(((w // bs) * (h // bs), 3), np.float32)
    n = lib().orc_gftt_collect_max(_p(response, f32p), w, h, bs, min_response, _p(kp, f32p))
    assert n == len(kp)
    return kp


def apply_min_distance(corners, prev, r: int, max_tracks: int = 200) -> np.ndarray:
    c = np.ascontiguousarray(corners, np.float32).reshape(-1, 2).copy()
    pv = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
    n = lib().orc_apply_min_distance(_p(c, f32p), len(c), _p(pv, f32p), len(pv), int(r), int(max_tracks))
    return c[:n].copy()


def gftt_detect(img: np.ndarray, prev=(), mask_radius: int = 0, block_size: int = 3, min_distance: float = 50.0,
                min_response: float = 1e-3, max_tracks: int = 200) -> np.ndarray:
    """FeatureDetectorImplementation::detect on the CPU-fallback path (incl. the prepended zero points)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    bs = gftt_block_size(min_distance)
    cap = 2 * (w // bs) * (h // bs)
    out = np.zeros((max(cap, 1), 2), np.float32)
    pv = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
    n = lib().orc_gftt_detect(_p(img, u8p), w, h, w, block_size, float(min_distance), min_response, _p(pv, f32p), len(pv),
                              int(mask_radius), int(max_tracks), _p(out, f32p), cap)
    if n < 0:
        raise RuntimeError(f"orc_gftt_detect failed: {n}")
    return out[:n].copy()


# ---- image ingest: colour -> gray, undistort / rectify remap (oracle/ingest_oracle.c) ----
class Camera:
    """tracker::Camera (camera.cpp): kind 'pinhole' (radial k1..k3, optional rotation) or 'fisheye' (k1..k4)."""

    def __init__(self, kind, fx, fy, ppx, ppy, coeffs=(), rotation=None, max_valid_fov_deg=180.0):
        co = np.ascontiguousarray(coeffs, np.float64).reshape(-1)
        rot = None if rotation is None else np.ascontiguousarray(rotation, np.float64).reshape(9)
        self.h = C.c_void_p(lib().orc_camera_create({"pinhole": 0, "fisheye": 1}[kind], fx, fy, ppx, ppy, len(co),
                                                    _p(co, f64p) if len(co) else None,
                                                    _p(rot, f64p) if rot is not None else None, max_valid_fov_deg))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_camera_destroy(self.h)
            self.h = None

    def pixel_to_ray(self, x, y):
        ray = np.zeros(3)
        ok = lib().orc_camera_pixel_to_ray(self.h, float(x), float(y), _p(ray, f64p))
        return bool(ok), ray

    def ray_to_pixel(self, ray):
        r = np.ascontiguousarray(ray, np.float64)
        pix = np.zeros(2)
        ok = lib().orc_camera_ray_to_pixel(self.h, _p(r, f64p), _p(pix, f64p))
        return bool(ok), pix


def mono_rectified_camera(w, h, focal_length, zoom=1.0):
    """Undistorter::buildMono (undistorter.cpp:150-168): equal focal lengths, principal point at the image centre."""
    f = np.float32(focal_length) * np.float32(zoom)          # float focalLength * float rectificationZoom
    return Camera("pinhole", float(f), float(f), float(np.float32(w) * np.float32(0.5)), float(np.float32(h) * np.float32(0.5)))


def undistort_map(rect: Camera, orig: Camera, w: int, h: int):
    pix = np.zeros((h, w, 2), np.float64)
    valid = np.zeros((h, w), np.uint8)
    lib().orc_undistort_map(rect.h, orig.h, w, h, _p(pix, f64p), _p(valid, u8p))
    return pix, valid


def undistort_apply(img: np.ndarray, pix_orig: np.ndarray, valid: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_undistort_apply(_p(img, u8p), w, h, _p(np.ascontiguousarray(pix_orig, np.float64), f64p),
                              _p(np.ascontiguousarray(valid, np.uint8), u8p), _p(out, u8p))
    return out


def color_to_gray(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, np.uint8)
    h, w, ch = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_color_to_gray(_p(img, u8p), w * ch, w, h, ch, _p(out, u8p))
    return out


# ---- triangulation + prepareVisualUpdate (oracle/triangulation_oracle.c) ----
class TriParams(C.Structure):
    _fields_ = [("triangulationConvergenceThreshold", C.c_double), ("triangulationConvergenceR", C.c_double),
                ("triangulationRcondThreshold", C.c_double), ("triangulationGaussNewtonIterations", C.c_uint),
                ("triangulationMinDist", C.c_double), ("triangulationMaxDist", C.c_double),
                ("estimateImuCameraTimeShift", C.c_int), ("useLinearTriangulation", C.c_int)]


class CamPose(C.Structure):      # 3x3 matrices row-major
    _fields_ = [("p", C.c_double * 3), ("R", C.c_double * 9), ("dR", (C.c_double * 9) * 4), ("baseline", C.c_double * 3)]


TRI_STATUS = ("OK", "HYBRID", "BEHIND", "BAD_COND", "NO_CONVERGENCE", "BAD_DEPTH", "UNKNOWN_PROBLEM")


def tri_default_params(**over):
    p = TriParams()
    lib().orc_tri_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def _f64(a):
    return np.ascontiguousarray(a, np.float64)


def vec2matrix(v):
    """odometry::util::vec2matrix (util.hpp:92-110): 9 values = column-major rotation, 16 = column-major 4x4."""
    v = np.asarray(v, np.float64)
    m = np.eye(4)
    if v.size == 9:
        m[:3, :3] = v.reshape(3, 3).T
    elif v.size == 16:
        m = v.reshape(4, 4).T.copy()
    else:
        raise ValueError(v.size)
    return m


def make_pose(p, q, baseline=(0, 0, 0), imu_to_cam_rot=None):
    """CameraPose from a camera position and a quaternion (R = imuToCamRot * quat2rmat(q)), as the reference tests do."""
    T = np.eye(4)
    if imu_to_cam_rot is not None:
        T[:3, :3] = imu_to_cam_rot
    T[:3, 3] = baseline
    m = np.zeros(20)
    m[0:3] = np.asarray(p, np.float64)
    m[6:10] = q
    tr = extract_camera_pose_trail(m, [0], T)
    tr[0].p[:] = list(np.asarray(p, np.float64))
    return tr


def extract_camera_pose_trail(m, pose_trail_index, imu_to_cam, imu_to_cam2=None):
    idx = np.ascontiguousarray(pose_trail_index, np.int32)
    n = len(idx)
    trail = (CamPose * (n * (2 if imu_to_cam2 is not None else 1)))()
    a, b = _f64(imu_to_cam), (None if imu_to_cam2 is None else _f64(imu_to_cam2))
    lib().orc_extract_camera_pose_trail(_p(_f64(m), f64p), _p(idx, i32p), n, _p(a, f64p), None if b is None else _p(b, f64p), trail)
    return trail


def inverse_depth(p):
    ip, dip = np.zeros(3), np.zeros((3, 3))
    lib().orc_inverse_depth(_p(_f64(p), f64p), _p(ip, f64p), _p(dip, f64p))
    return ip, dip


def pinv32(A):
    out = np.zeros((2, 3))
    lib().orc_pinv32(_p(_f64(A), f64p), _p(out, f64p))
    return out


def triangulate_with_two_cameras(pose0, pose1, ip0, ip1, vel0=(0, 0), vel1=(0, 0), calc_derivatives=False,
                                 estimate_time_shift=False, derivative_test=False, time_shift=0.0):
    pf, dpf = np.zeros(3), np.zeros((3, 15))
    lib().orc_triangulate_with_two_cameras(C.byref(pose0), C.byref(pose1), _p(_f64(ip0), f64p), _p(_f64(ip1), f64p), _p(_f64(vel0), f64p),
                                           _p(_f64(vel1), f64p), int(calc_derivatives), int(estimate_time_shift), int(derivative_test),
                                           float(time_shift), _p(pf, f64p), _p(dpf, f64p))
    return pf, dpf


def triangulate(par, trail, image_features, feature_velocities, stereo=False, calc_derivatives=True, derivative_test=False,
                time_shift=0.0):
    n = len(trail)
    pf, dp, dq, dt = np.zeros(3), np.zeros((n, 3, 3)), np.zeros((n, 3, 4)), np.zeros(3)
    st = lib().orc_triangulate(C.byref(par), n, trail, _p(_f64(image_features), f64p), _p(_f64(feature_velocities), f64p), int(stereo),
                               int(calc_derivatives), int(derivative_test), float(time_shift), _p(pf, f64p), _p(dp, f64p), _p(dq, f64p),
                               _p(dt, f64p))
    return st, pf, dp, dq, dt


def prepare_visual_update(pf, dpfdp, dpfdq, dpfdt, feature_velocities, trail, pose_trail_index, state_dim, truncated=False,
                          map_point_offset=-1, estimate_time_shift=True, derivative_test=False, time_shift=0.0):
    idx = np.ascontiguousarray(pose_trail_index, np.int32)
    nt = len(trail)
    H = np.zeros((state_dim, 2 * nt))                       # column-major (2 nt) x end
    f = np.zeros(2 * nt)
    end = np.zeros(1, np.int32)
    dp = None if dpfdp is None else _f64(dpfdp)
    dq = None if dpfdq is None else _f64(dpfdq)
    st = lib().orc_prepare_visual_update(_p(_f64(pf), f64p), None if dp is None else _p(dp, f64p), None if dq is None else _p(dq, f64p),
                                         _p(_f64(dpfdt), f64p), _p(_f64(feature_velocities), f64p), trail, nt, _p(idx, i32p), len(idx),
                                         state_dim, int(truncated), map_point_offset, int(estimate_time_shift), int(derivative_test),
                                         float(time_shift), _p(H, f64p), _p(f, f64p), _p(end, i32p))
    e = int(end[0])
    return st, H.reshape(-1)[:2 * nt * e].reshape(e, 2 * nt).T.copy(), f


def visual_track_prepare(par, m, pose_trail_index, imu_to_cam, imu_to_cam2, image_features, feature_velocities):
    """backend.cpp:1063-1148 for one pose-trail track: (triangulation status, prepare status, pf, H, f)."""
    m = _f64(m)
    idx = np.ascontiguousarray(pose_trail_index, np.int32)
    nt = len(idx) * (2 if imu_to_cam2 is not None else 1)
    H = np.zeros((len(m), 2 * nt))
    f, pf, ps = np.zeros(2 * nt), np.zeros(3), np.zeros(1, np.int32)
    a, b = _f64(imu_to_cam), (None if imu_to_cam2 is None else _f64(imu_to_cam2))
    st = lib().orc_visual_track_prepare(C.byref(par), _p(m, f64p), len(m), _p(idx, i32p), len(idx), _p(a, f64p),
                                        None if b is None else _p(b, f64p), _p(_f64(image_features), f64p),
                                        _p(_f64(feature_velocities), f64p), _p(pf, f64p), _p(H, f64p), _p(f, f64p), _p(ps, i32p))
    return st, int(ps[0]), pf, H.T.copy(), f


def tri_last_diag():
    """(min rcond of E'E, min |h_z| / |h|, iterations, converged) of this thread's last triangulation: see orc_tri_last_diag."""
    out = np.zeros(4)
    lib().orc_tri_last_diag(_p(out, f64p))
    return out


# ---- 2-point rotation RANSAC (oracle/rot_ransac_oracle.c) ----
def mt19937_draws(seed: int, n: int, skip: int = 0) -> np.ndarray:
    """Raw outputs of std::mt19937(seed) after `skip` draws."""
    out = np.zeros(n, np.uint32)
    lib().orc_mt19937_draws(int(seed), int(skip), int(n), out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def solve_rotation(p1, p2, inds) -> np.ndarray:
    a, b = np.ascontiguousarray(p1, np.float32), np.ascontiguousarray(p2, np.float32)
    ii = np.ascontiguousarray(inds, np.int32)
    R = np.zeros((3, 3), np.float32)
    lib().orc_solve_rotation(_p(a, f32p), _p(b, f32p), _p(ii, i32p), len(ii), _p(R, f32p))
    return R


def rot_ransac_fit(c1, c2, cam1: "Camera", cam2: "Camera", draws, threshold_pow2: float):
    """RotRansac::fit: returns (status [n] 0 TRACKED / 3 RANSAC_OUTLIER, R 3x3 f32, bestInlierCount, draws consumed)."""
    a, b = np.ascontiguousarray(c1, np.float32).reshape(-1, 2), np.ascontiguousarray(c2, np.float32).reshape(-1, 2)
    d = np.ascontiguousarray(draws, np.uint32)
    assert len(d) >= 200
    st, R, best = np.zeros(len(a), np.int32), np.zeros((3, 3), np.float32), np.zeros(1, np.int32)
    used = lib().orc_rot_ransac_fit(_p(a, f32p), _p(b, f32p), len(a), cam1.h, cam2.h, d.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    float(threshold_pow2), _p(st, i32p), _p(R, f32p), _p(best, i32p))
    return st, R, int(best[0]), int(used)
