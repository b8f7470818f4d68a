"""ctypes binding of include/hybvio_hip.h (libhybvio_hip.so).

Python is only the test / bench harness language here: the product is the C-ABI library and the
C++ adapters under hybvio_amd/host/. There is no CPU fallback: if the library cannot be built or
loaded, or no HIP device is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

# The torch wheel bundles its own libamdhip64/libhsa-runtime64. A process must initialise exactly
# ONE HIP/HSA runtime, so when torch is installed it is imported BEFORE libhybvio_hip.so is loaded:
# the dynamic linker then resolves our NEEDED libamdhip64.so.7 to the copy torch already mapped.
# (A C++ host such as the reference `main` has no torch and simply uses /opt/rocm.)
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

from . import build as _build

u8p = C.POINTER(C.c_uint8)
i16p = C.POINTER(C.c_int16)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)

HV_MAX_LEVELS = 6
K_PYR_L0, K_PYR_LN, K_KLT, K_EKF_PREDICT, K_EKF_UPDATE, K_EKF_AUGMENT, K_GFTT, K_INGEST, K_VU_PREPARE, K_ROT_RANSAC, K_EKF_GATE, K_VU_TRI = range(12)

# tracker::Feature::Status (src/tracker/track.hpp:9-21)
ST_TRACKED, ST_NEW, ST_FAILED_FLOW, ST_RANSAC_OUTLIER, ST_FLOW_OUT_OF_RANGE = 0, 1, 2, 3, 4


class Params(C.Structure):
    _fields_ = [("device", C.c_int), ("width", C.c_int), ("height", C.c_int), ("levels", C.c_int),
                ("win", C.c_int), ("max_iter", C.c_int), ("eps", C.c_double), ("min_eig", C.c_double),
                ("max_tracks", C.c_int), ("pool_size", C.c_int), ("max_pairs", C.c_int)]


class EkfParams(C.Structure):
    _fields_ = [("cameraTrailLength", C.c_int), ("hybridMapSize", C.c_int)] + [
        (k, C.c_double) for k in (
            "noiseScale", "gravity", "augmentR", "initZuptR", "rotationZuptR",
            "noiseInitialPos", "noiseInitialOri", "noiseInitialVel", "noiseInitialPosTrail", "noiseInitialOriTrail",
            "noiseInitialBGA", "noiseInitialBAA", "noiseInitialBAT", "noiseInitialSFT",
            "noiseProcessAcc", "noiseProcessGyro", "noiseProcessBAA", "noiseProcessBGA",
            "noiseProcessBAARev", "noiseProcessBGARev")]


class GfttParams(C.Structure):
    _fields_ = [("gfttBlockSize", C.c_int), ("gfttMinDistance", C.c_double), ("gfttMinResponse", C.c_float),
                ("maxTracks", C.c_int)]


class HvError(RuntimeError):
    pass


_LIB = None

# name -> (restype, argtypes); also the list of symbols tests check against the header.
PROTOTYPES = {
    "hv_default_params": (None, [C.POINTER(Params)]),
    "hv_abi_version": (C.c_int, []),
    "hv_status_string": (C.c_char_p, [C.c_int]),
    "hv_debug_set_knob": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "hv_debug_get_knob": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "hv_ekf_frame_error": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hv_create": (C.c_int, [C.POINTER(Params), C.POINTER(C.c_void_p)]),
    "hv_destroy": (None, [C.c_void_p]),
    "hv_last_error": (C.c_char_p, [C.c_void_p]),
    "hv_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "hv_synchronize": (C.c_int, [C.c_void_p]),
    "hv_get_stream": (C.c_void_p, [C.c_void_p]),
    "hv_lanes_create": (C.c_int, [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]),
    "hv_lanes_count": (C.c_int, [C.c_void_p]),
    "hv_lanes_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "hv_lanes_destroy": (None, [C.c_void_p]),
    "hv_pyramid_acquire": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "hv_pyramid_release": (C.c_int, [C.c_void_p, C.c_int]),
    "hv_pyramid_level_size": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "hv_pyramid_build": (C.c_int, [C.c_void_p, C.c_int, u8p, C.c_int]),
    "hv_ingest_set_undistort_map": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "hv_ingest_build": (C.c_int, [C.c_void_p, C.c_int, u8p, C.c_int, C.c_int, C.c_int]),
    "hv_ingest_build_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int]),
    "hv_pyramid_build_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_longlong, C.c_int]),
    "hv_pyramid_download": (C.c_int, [C.c_void_p, C.c_int, C.c_int, u8p, i16p]),
    "hv_klt_track": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, f32p, u8p, f32p, C.c_int, C.c_int]),
    "hv_optical_flow_compute": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, f32p, i32p, C.c_int, C.c_int]),
    "hv_klt_track_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hv_klt_track_batch_ragged_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "hv_ekf_default_params": (None, [C.POINTER(EkfParams)]),
    "hv_ekf_create": (C.c_int, [C.c_void_p, C.POINTER(EkfParams), C.c_int, C.POINTER(C.c_void_p)]),
    "hv_ekf_destroy": (None, [C.c_void_p]),
    "hv_ekf_state_dim": (C.c_int, [C.c_void_p]),
    "hv_ekf_batch": (C.c_int, [C.c_void_p]),
    "hv_ekf_set_state": (C.c_int, [C.c_void_p, C.c_int, f64p, f64p]),
    "hv_ekf_get_state": (C.c_int, [C.c_void_p, C.c_int, f64p, f64p]),
    "hv_ekf_get_means": (C.c_int, [C.c_void_p, f64p]),
    "hv_ekf_set_process_noise": (C.c_int, [C.c_void_p, C.c_int, f64p]),
    "hv_ekf_get_process_noise": (C.c_int, [C.c_void_p, C.c_int, f64p]),
    "hv_ekf_get_dydx": (C.c_int, [C.c_void_p, C.c_int, f64p]),
    "hv_ekf_device_pointers": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "hv_ekf_predict": (C.c_int, [C.c_void_p, f64p, f64p, f64p]),
    "hv_ekf_predict_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hv_ekf_predict_n_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hv_ekf_predict_n": (C.c_int, [C.c_void_p, C.c_int, f64p, f64p, f64p]),
    "hv_ekf_update": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f64p, f64p, f64p, u8p, C.c_int]),
    "hv_ekf_visual_gate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f64p, f64p, C.c_double, f64p, i32p]),
    "hv_ekf_visual_update": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f64p, f64p, C.c_double, u8p]),
    "hv_camera_model_init": (C.c_int, [C.c_void_p]),
    "hv_rot_ransac": (C.c_int, [C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] + [C.c_void_p] * 4),
    "hv_rot_ransac_lk_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 3),
    "hv_rot_ransac_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3),
    "hv_vu_default_params": (None, [C.c_void_p]),
    "hv_ekf_visual_prepare_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 10),
    "hv_ekf_visual_track_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_double, C.c_double] + [C.c_void_p] * 4),
    "hv_ekf_visual_track_hybrid_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_double, C.c_double] + [C.c_void_p] * 4),
    "hv_ekf_visual_track_limited_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int]),
    "hv_ekf_visual_track": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_double, C.c_double] + [C.c_void_p] * 4),
    "hv_ekf_visual_frame_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int]),
    "hv_ekf_visual_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int]),
    "hv_ekf_visual_frame_ragged": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int]),
    "hv_ekf_visual_frame_ragged_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int]),
    "hv_ekf_visual_frame_batch_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int, C.c_int]),
    "hv_ekf_augment_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hv_ekf_symmetrize_augment_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    # ABI 4: host-pointer forms of the r04 frame entries + map points on the resident state
    "hv_ekf_symmetrize_augment": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "hv_ekf_visual_frame_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [C.c_double, C.c_double] + [C.c_void_p] * 5 + [C.c_int, C.c_int]),
    "hv_ekf_visual_track_hybrid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_double, C.c_double] + [C.c_void_p] * 4),
    "hv_ekf_insert_map_point": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hv_ekf_get_map_point": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "hv_ekf_visual_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    "hv_ekf_augment": (C.c_int, [C.c_void_p, i32p, u8p]),
    "hv_ekf_undo_augment": (C.c_int, [C.c_void_p, u8p]),
    "hv_ekf_symmetrize": (C.c_int, [C.c_void_p]),
    "hv_ekf_normalize_quaternions": (C.c_int, [C.c_void_p, C.c_int]),
    "hv_ekf_transform": (C.c_int, [C.c_void_p, C.c_int, f64p, f64p, f64p]),
    "hv_gftt_default_params": (None, [C.POINTER(GfttParams)]),
    "hv_gftt_block_size": (C.c_int, [C.POINTER(GfttParams)]),
    "hv_gftt_keypoint_count": (C.c_int, [C.c_void_p, C.POINTER(GfttParams)]),
    "hv_gftt_detect": (C.c_int, [C.c_void_p, C.POINTER(GfttParams), C.c_int, f32p, C.c_int, C.c_int, f32p, C.c_int,
                                 C.POINTER(C.c_int)]),
    "hv_gftt_keypoints_batch_dev": (C.c_int, [C.c_void_p, C.POINTER(GfttParams), C.c_int, C.c_void_p, C.c_void_p]),
    "hv_apply_min_distance": (None, [f32p, C.POINTER(C.c_int), f32p, C.c_int, C.c_int, C.c_int]),
    "hv_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "hv_profile_reset": (C.c_int, [C.c_void_p]),
    "hv_profile_read": (C.c_int, [C.c_void_p, C.c_int, f64p, C.POINTER(C.c_longlong)]),
}


def lib():
    """Load (building if needed) libhybvio_hip.so. Raises if it cannot be produced."""
    global _LIB
    if _LIB is None:
        path = _build.build_hip()
        # HV_LIB_OVERRIDE (developer A/B runs only): another build of the same library, e.g. the previous commit's, in the same process setup
        path = os.environ.get("HV_LIB_OVERRIDE", path)
        if not os.path.exists(path):
            raise HvError(f"{path} missing: the HIP extension is required, there is no fallback")
        L = C.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


class Context:
    """One hv_ctx: a HIP stream, a pool of pyramid slots and the tracker parameters."""

    def __init__(self, width=752, height=480, levels=4, win=31, max_iter=20, eps=0.03, min_eig=1e-3,
                 max_tracks=200, pool_size=16, max_pairs=1, device=0, _adopt=None):
        L = lib()
        p = Params()
        L.hv_default_params(C.byref(p))
        p.device, p.width, p.height, p.levels, p.win = device, width, height, levels, win
        p.max_iter, p.eps, p.min_eig, p.max_tracks = max_iter, eps, min_eig, max_tracks
        p.pool_size, p.max_pairs = pool_size, max_pairs
        self.params = p
        self._owned = _adopt is None
        if _adopt is not None:                       # a lane of a Lanes set: the set owns the hv_ctx
            self._h = C.c_void_p(_adopt)
        else:
            self._h = C.c_void_p()
            rc = L.hv_create(C.byref(p), C.byref(self._h))
            if rc != 0:
                self._h = None
                raise HvError(f"hv_create: {L.hv_status_string(rc).decode()}")
        self.levels = 0
        self.level_sizes = []
        w, h = C.c_int(), C.c_int()
        while self.levels < levels and L.hv_pyramid_level_size(self._h, self.levels, C.byref(w), C.byref(h)) == 0:
            self.level_sizes.append((w.value, h.value))
            self.levels += 1

    def close(self):
        if getattr(self, "_h", None):
            for child in list(getattr(self, "_children", [])):   # hv_ekf objects must die before their hv_ctx
                child.close()
            if self._owned:
                lib().hv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:                             # interpreter shutdown: module globals may be gone already
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc != 0:
            L = lib()
            msg = L.hv_status_string(rc).decode()
            if rc == -4:
                msg += ": " + L.hv_last_error(self._h).decode()
            raise HvError(f"{what}: {msg}")

    # -- plumbing --
    def set_knob(self, name: str, value: int):
        """Force a kernel variant (tests / measurements; include/hybvio_hip.h hv_debug_set_knob)."""
        self._chk(lib().hv_debug_set_knob(self._h, name.encode(), int(value)), f"hv_debug_set_knob({name})")

    def get_knob(self, name: str) -> int:
        v = C.c_int()
        self._chk(lib().hv_debug_get_knob(self._h, name.encode(), C.byref(v)), f"hv_debug_get_knob({name})")
        return v.value

    def set_stream(self, stream_ptr: int):
        self._chk(lib().hv_set_stream(self._h, C.c_void_p(stream_ptr)), "hv_set_stream")

    def get_stream(self) -> int:
        """The hipStream_t the context issues on (hv_get_stream), as an integer handle."""
        return int(lib().hv_get_stream(self._h) or 0)

    def synchronize(self):
        self._chk(lib().hv_synchronize(self._h), "hv_synchronize")

    # -- pyramid --
    def acquire(self) -> int:
        s = C.c_int()
        self._chk(lib().hv_pyramid_acquire(self._h, C.byref(s)), "hv_pyramid_acquire")
        return s.value

    def release(self, slot: int):
        self._chk(lib().hv_pyramid_release(self._h, slot), "hv_pyramid_release")

    def build(self, slot: int, gray: np.ndarray):
        gray = np.ascontiguousarray(gray, np.uint8)
        assert gray.shape == (self.params.height, self.params.width), gray.shape
        self._chk(lib().hv_pyramid_build(self._h, slot, _p(gray, u8p), gray.strides[0]), "hv_pyramid_build")
        self.synchronize()   # the numpy temporary may die after return

    def build_batch_dev(self, n: int, slots_dev: int, gray_dev: int, image_stride: int, row_stride: int):
        self._chk(lib().hv_pyramid_build_batch_dev(self._h, n, C.c_void_p(slots_dev), C.c_void_p(gray_dev),
                                                   image_stride, row_stride), "hv_pyramid_build_batch_dev")

    # ---- 2-point rotation RANSAC (f4) ----
    def rot_ransac(self, c1, c2, cam1: "CameraModel", cam2: "CameraModel", pairs, threshold_pow2: float):
        """hv_rot_ransac: returns (status [n], R 3x3 f32, bestInlierCount, hypotheses visited)."""
        a, b = np.ascontiguousarray(c1, np.float32).reshape(-1, 2), np.ascontiguousarray(c2, np.float32).reshape(-1, 2)
        pr = np.ascontiguousarray(pairs, np.int32).reshape(100, 2)
        st, R = np.zeros(len(a), np.int32), np.zeros((3, 3), np.float32)
        best, vis = C.c_int(0), C.c_int(0)
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        self._chk(lib().hv_rot_ransac(self._h, len(a), vp(a), vp(b), C.byref(cam1), C.byref(cam2), vp(pr), float(threshold_pow2),
                                      vp(st), vp(R), C.byref(best), C.byref(vis)), "hv_rot_ransac")
        return st, R, best.value, vis.value

    def rot_ransac_batch_dev(self, n_sets, max_points, n_points_dev, c1_dev, c2_dev, cam1, cam2, pairs_dev, threshold_pow2,
                             status_dev, R_dev, summary_dev):
        p = lambda x: C.c_void_p(x)
        self._chk(lib().hv_rot_ransac_batch_dev(self._h, n_sets, max_points, p(n_points_dev), p(c1_dev), p(c2_dev), C.byref(cam1),
                                                C.byref(cam2), p(pairs_dev), float(threshold_pow2), p(status_dev), p(R_dev),
                                                p(summary_dev)), "hv_rot_ransac_batch_dev")

    def rot_ransac_lk_batch_dev(self, n_sets, max_points, n_points_dev, c1_dev, c2_dev, lk_status_dev, lk_tracked_value, cam1, cam2,
                                draws_dev, threshold_pow2, status_dev, R_dev, summary_dev):
        p = lambda x: C.c_void_p(x)
        self._chk(lib().hv_rot_ransac_lk_batch_dev(self._h, n_sets, max_points, p(n_points_dev), p(c1_dev), p(c2_dev), p(lk_status_dev),
                                                   int(lk_tracked_value), C.byref(cam1), C.byref(cam2), p(draws_dev),
                                                   float(threshold_pow2), p(status_dev), p(R_dev), p(summary_dev)),
                  "hv_rot_ransac_lk_batch_dev")

    # ---- image ingest (f2): colour -> gray and the undistort / rectify remap in front of the pyramid ----
    def ingest_set_undistort_map(self, camera: int, pix_orig=None, valid=None):
        """pix_orig (h, w, 2) f64 = original-image position of every rectified pixel (None removes the table)."""
        if pix_orig is None:
            self._chk(lib().hv_ingest_set_undistort_map(self._h, camera, None, None), "hv_ingest_set_undistort_map")
            return
        pix = np.ascontiguousarray(pix_orig, np.float64)
        assert pix.shape == (self.params.height, self.params.width, 2), pix.shape
        v = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        self._chk(lib().hv_ingest_set_undistort_map(self._h, camera, pix.ctypes.data_as(C.c_void_p),
                                                    None if v is None else v.ctypes.data_as(C.c_void_p)),
                  "hv_ingest_set_undistort_map")

    def ingest_build(self, slot: int, image: np.ndarray, camera: int = -1):
        """image (h, w) gray or (h, w, 3|4) colour, u8."""
        image = np.ascontiguousarray(image, np.uint8)
        ch = 1 if image.ndim == 2 else image.shape[2]
        assert image.shape[:2] == (self.params.height, self.params.width), image.shape
        self._chk(lib().hv_ingest_build(self._h, slot, _p(image, u8p), image.strides[0], ch, camera), "hv_ingest_build")
        self.synchronize()

    def ingest_build_batch_dev(self, n: int, slots_dev: int, src_dev: int, image_stride: int, row_stride: int,
                               channels: int = 1, camera: int = -1):
        self._chk(lib().hv_ingest_build_batch_dev(self._h, n, C.c_void_p(slots_dev), C.c_void_p(src_dev), image_stride,
                                                  row_stride, channels, camera), "hv_ingest_build_batch_dev")

    def download(self, slot: int, level: int):
        w, h = self.level_sizes[level]
        g = np.empty((h, w), np.uint8)
        d = np.empty((h, w, 2), np.int16)
        self._chk(lib().hv_pyramid_download(self._h, slot, level, _p(g, u8p), _p(d, i16p)), "hv_pyramid_download")
        return g, d

    # -- Lucas-Kanade --
    def klt_track(self, prev_slot, next_slot, prev_xy, next_xy=None, max_iter_override=-1):
        prev_xy = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
        n = prev_xy.shape[0]
        use_init = next_xy is not None
        out = (np.ascontiguousarray(next_xy, np.float32).reshape(-1, 2).copy() if use_init
               else np.zeros_like(prev_xy))
        st = np.zeros(n, np.uint8)
        err = np.zeros(n, np.float32)
        self._chk(lib().hv_klt_track(self._h, prev_slot, next_slot, n, _p(prev_xy, f32p), _p(out, f32p),
                                     _p(st, u8p), _p(err, f32p), int(use_init), max_iter_override), "hv_klt_track")
        return out, st, err

    def optical_flow_compute(self, prev_slot, cur_slot, prev_corners, corners=None, override_max_iterations=-1):
        prev_corners = np.ascontiguousarray(prev_corners, np.float32).reshape(-1, 2)
        n = prev_corners.shape[0]
        use_init = corners is not None
        out = (np.ascontiguousarray(corners, np.float32).reshape(-1, 2).copy() if use_init
               else np.zeros_like(prev_corners))
        st = np.full(n, ST_FAILED_FLOW, np.int32)
        self._chk(lib().hv_optical_flow_compute(self._h, prev_slot, cur_slot, n, _p(prev_corners, f32p),
                                                _p(out, f32p), _p(st, i32p), int(use_init),
                                                override_max_iterations), "hv_optical_flow_compute")
        return out, st

    def klt_track_batch_dev(self, n_pairs, prev_slots_dev, next_slots_dev, pts_per_pair, prev_xy_dev,
                            next_xy_dev, status_dev, err_dev, use_initial_flow=True, max_iter_override=-1):
        self._chk(lib().hv_klt_track_batch_dev(self._h, n_pairs, C.c_void_p(prev_slots_dev),
                                               C.c_void_p(next_slots_dev), pts_per_pair, C.c_void_p(prev_xy_dev),
                                               C.c_void_p(next_xy_dev), C.c_void_p(status_dev),
                                               C.c_void_p(err_dev), int(use_initial_flow), max_iter_override),
                  "hv_klt_track_batch_dev")

    def klt_track_batch_ragged_dev(self, n_pairs, prev_slots_dev, next_slots_dev, pts_per_pair, pts_in_pair_dev, prev_xy_dev,
                                   next_xy_dev, status_dev, err_dev, use_initial_flow=True, max_iter_override=-1):
        self._chk(lib().hv_klt_track_batch_ragged_dev(self._h, n_pairs, C.c_void_p(prev_slots_dev), C.c_void_p(next_slots_dev),
                                                      pts_per_pair, C.c_void_p(pts_in_pair_dev), C.c_void_p(prev_xy_dev),
                                                      C.c_void_p(next_xy_dev), C.c_void_p(status_dev), C.c_void_p(err_dev),
                                                      int(use_initial_flow), max_iter_override), "hv_klt_track_batch_ragged_dev")

    # -- GFTT feature detector --
    def gftt_detect(self, slot: int, prev=(), mask_radius: int = 0, params: "GfttParams" = None):
        """FeatureDetector::detect on the level-0 image of a built pyramid slot -> corners [n, 2]."""
        gp = params if params is not None else gftt_default_params()
        nk = lib().hv_gftt_keypoint_count(self._h, C.byref(gp))
        out = np.zeros((max(2 * nk, 1), 2), np.float32)
        pv = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
        n = C.c_int(0)
        self._chk(lib().hv_gftt_detect(self._h, C.byref(gp), slot, _p(pv, f32p) if len(pv) else None, len(pv),
                                       int(mask_radius), _p(out, f32p), 2 * nk, C.byref(n)), "hv_gftt_detect")
        return out[:n.value].copy()

    def gftt_keypoint_count(self, params: "GfttParams" = None) -> int:
        gp = params if params is not None else gftt_default_params()
        return int(lib().hv_gftt_keypoint_count(self._h, C.byref(gp)))

    def gftt_keypoints_batch_dev(self, n_images: int, slots_dev: int, kp_dev: int, params: "GfttParams" = None):
        gp = params if params is not None else gftt_default_params()
        self._chk(lib().hv_gftt_keypoints_batch_dev(self._h, C.byref(gp), n_images, C.c_void_p(slots_dev),
                                                    C.c_void_p(kp_dev)), "hv_gftt_keypoints_batch_dev")

    # -- timers --
    def profile_enable(self, on=True):
        self._chk(lib().hv_profile_enable(self._h, int(on)), "hv_profile_enable")

    def profile_reset(self):
        self._chk(lib().hv_profile_reset(self._h), "hv_profile_reset")

    def profile_read(self, kernel_id: int):
        ms, n = C.c_double(), C.c_longlong()
        self._chk(lib().hv_profile_read(self._h, kernel_id, C.byref(ms), C.byref(n)), "hv_profile_read")
        return ms.value, n.value


def gftt_default_params(**over) -> GfttParams:
    p = GfttParams()
    lib().hv_gftt_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def apply_min_distance(corners, prev, r: int, max_tracks: int = 200) -> np.ndarray:
    """FeatureDetector::applyMinDistance (host code of the library)."""
    c = np.ascontiguousarray(corners, np.float32).reshape(-1, 2).copy()
    pv = np.ascontiguousarray(prev, np.float32).reshape(-1, 2)
    n = C.c_int(len(c))
    lib().hv_apply_min_distance(_p(c, f32p), C.byref(n), _p(pv, f32p) if len(pv) else None, len(pv), int(r), int(max_tracks))
    return c[:n.value].copy()


def ekf_default_params(**over) -> EkfParams:
    p = EkfParams()
    lib().hv_ekf_default_params(C.byref(p))
    for k, v in over.items():
        setattr(p, k, v)
    return p


def _f(a):
    return np.ascontiguousarray(a, np.float64)


class CameraModel(C.Structure):
    """hv_camera_model (tracker::Camera, camera.cpp)."""
    _fields_ = [("kind", C.c_int), ("fx", C.c_double), ("fy", C.c_double), ("ppx", C.c_double), ("ppy", C.c_double),
                ("n_coeffs", C.c_int), ("coeffs", C.c_double * 4), ("rotation_enabled", C.c_int), ("rotation", C.c_double * 9),
                ("max_valid_fov_deg", C.c_double), ("distortion_enabled", C.c_int), ("kinv", C.c_double * 9),
                ("max_theta", C.c_double), ("max_r", C.c_double), ("n_table", C.c_int), ("table", C.c_double * 50)]


def camera_model(kind, fx, fy, ppx, ppy, coeffs=(), rotation=None, max_valid_fov_deg=180.0) -> CameraModel:
    m = CameraModel()
    m.kind = {"pinhole": 0, "fisheye": 1}[kind]
    m.fx, m.fy, m.ppx, m.ppy = fx, fy, ppx, ppy
    m.n_coeffs = len(coeffs)
    for i, c in enumerate(coeffs):
        m.coeffs[i] = c
    if rotation is not None:
        m.rotation_enabled = 1
        m.rotation[:] = list(np.asarray(rotation, np.float64).reshape(9))
    m.max_valid_fov_deg = max_valid_fov_deg
    rc = lib().hv_camera_model_init(C.byref(m))
    if rc != 0:
        raise HvError(f"hv_camera_model_init: {lib().hv_status_string(rc).decode()}")
    return m


class VuParams(C.Structure):
    """hv_vu_params (field names = the reference's parameters)."""
    _fields_ = [("triangulationConvergenceThreshold", C.c_double), ("triangulationConvergenceR", C.c_double),
                ("triangulationRcondThreshold", C.c_double), ("triangulationGaussNewtonIterations", C.c_uint),
                ("triangulationMinDist", C.c_double), ("triangulationMaxDist", C.c_double),
                ("estimateImuCameraTimeShift", C.c_int), ("useStereo", C.c_int),
                ("imuToCamera", C.c_double * 16), ("secondImuToCamera", C.c_double * 16),
                ("useLinearTriangulation", C.c_int),
                ("trackRmseThreshold", C.c_double), ("trackOutlierThresholdGrowthFactor", C.c_double)]


def vu_default_params(imu_to_camera=None, second_imu_to_camera=None, **over) -> VuParams:
    p = VuParams()
    lib().hv_vu_default_params(C.byref(p))
    if imu_to_camera is not None:
        p.imuToCamera[:] = list(np.asarray(imu_to_camera, np.float64).reshape(16))
    if second_imu_to_camera is not None:
        p.secondImuToCamera[:] = list(np.asarray(second_imu_to_camera, np.float64).reshape(16))
        p.useStereo = 1
    for k, v in over.items():
        setattr(p, k, v)
    return p


class Lanes:
    """hv_lanes: n batched contexts of one GPU whose streams come from the device's high-priority queue pool, so that their launch
    chains run beside each other whatever the process did before (include/hybvio_hip.h). ctx[i] are Context objects the set owns."""

    def __init__(self, n_lanes, **kw):
        L = lib()
        p = Params()
        L.hv_default_params(C.byref(p))
        vals = dict(width=752, height=480, levels=4, win=31, max_iter=20, eps=0.03, min_eig=1e-3, max_tracks=200, pool_size=16,
                    max_pairs=1, device=0)
        vals.update(kw)
        for k, v in vals.items():
            setattr(p, k, v)
        self._g = C.c_void_p()
        rc = L.hv_lanes_create(C.byref(p), int(n_lanes), C.byref(self._g))
        if rc != 0:
            self._g = None
            raise HvError(f"hv_lanes_create: {L.hv_status_string(rc).decode()}")
        n = L.hv_lanes_count(self._g)
        kw2 = {k: getattr(p, k) for k in ("width", "height", "levels", "win", "max_iter", "eps", "min_eig", "max_tracks", "pool_size", "max_pairs", "device")}
        self.ctx = [Context(_adopt=L.hv_lanes_ctx(self._g, i), **kw2) for i in range(n)]

    def close(self):
        if getattr(self, "_g", None):
            for c in self.ctx:
                c.close()                             # (closes the lanes' hv_ekf children; the contexts themselves die with the set)
            lib().hv_lanes_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:                             # interpreter shutdown: module globals may be gone already
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class EkfBatch:
    """hv_ekf: a batch of independent filters on one Context (batch = 1: the reference's EKF)."""

    def __init__(self, ctx: Context, params: EkfParams | None = None, batch: int = 1):
        self.ctx, self.batch = ctx, batch
        self.params = params if params is not None else ekf_default_params()
        self._h = C.c_void_p()
        rc = lib().hv_ekf_create(ctx._h, C.byref(self.params), batch, C.byref(self._h))
        if rc != 0:
            self._h = None
            raise HvError(f"hv_ekf_create: {lib().hv_status_string(rc).decode()}")
        self.n = lib().hv_ekf_state_dim(self._h)
        if not hasattr(ctx, "_children"):
            ctx._children = []
        ctx._children.append(self)

    def close(self):
        if getattr(self, "_h", None):
            lib().hv_ekf_destroy(self._h)
            self._h = None
            if self in getattr(self.ctx, "_children", []):
                self.ctx._children.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:                             # interpreter shutdown: module globals may be gone already
            pass

    def _chk(self, rc, what):
        self.ctx._chk(rc, what)

    def frame_error(self) -> int:
        """Reads and clears the device error word of the batch (hv_ekf_frame_error)."""
        v = C.c_int()
        self._chk(lib().hv_ekf_frame_error(self._h, C.byref(v)), "hv_ekf_frame_error")
        return v.value

    def set_state(self, b, m=None, P=None):
        mm = _f(m) if m is not None else None
        PP = np.asfortranarray(P, np.float64) if P is not None else None
        self._chk(lib().hv_ekf_set_state(self._h, b, _p(mm, f64p), PP.ctypes.data_as(f64p) if PP is not None else None),
                  "hv_ekf_set_state")

    def get_state(self, b):
        m, P = np.zeros(self.n), np.zeros((self.n, self.n))
        self._chk(lib().hv_ekf_get_state(self._h, b, _p(m, f64p), _p(P, f64p)), "hv_ekf_get_state")
        return m, P.T.copy()          # column-major buffer -> P[i, j]

    def get_means(self):
        m = np.zeros((self.batch, self.n))
        self._chk(lib().hv_ekf_get_means(self._h, _p(m, f64p)), "hv_ekf_get_means")
        return m

    def set_process_noise(self, b, Q):
        self._chk(lib().hv_ekf_set_process_noise(self._h, b, np.asfortranarray(Q, np.float64).ctypes.data_as(f64p)),
                  "hv_ekf_set_process_noise")

    def get_dydx(self, b):
        F = np.zeros((20, 20))
        self._chk(lib().hv_ekf_get_dydx(self._h, b, _p(F, f64p)), "hv_ekf_get_dydx")
        return F.T.copy()

    def device_pointers(self):
        m, P = C.c_void_p(), C.c_void_p()
        self._chk(lib().hv_ekf_device_pointers(self._h, C.byref(m), C.byref(P)), "hv_ekf_device_pointers")
        return m.value, P.value

    def predict(self, dt, gyro, acc):
        dt, gyro, acc = _f(np.broadcast_to(dt, (self.batch,))), _f(np.broadcast_to(gyro, (self.batch, 3))), _f(np.broadcast_to(acc, (self.batch, 3)))
        self._chk(lib().hv_ekf_predict(self._h, _p(dt, f64p), _p(gyro, f64p), _p(acc, f64p)), "hv_ekf_predict")
        self.ctx.synchronize()

    def predict_n_dev(self, n_samples, dt_dev, gyro_dev, acc_dev):
        """n_samples predicts in one launch; device arrays [n][batch], [n][batch][3], [n][batch][3]."""
        self._chk(lib().hv_ekf_predict_n_dev(self._h, int(n_samples), C.c_void_p(dt_dev), C.c_void_p(gyro_dev),
                                             C.c_void_p(acc_dev)), "hv_ekf_predict_n_dev")

    def predict_dev(self, dt_dev, gyro_dev, acc_dev):
        self._chk(lib().hv_ekf_predict_dev(self._h, C.c_void_p(dt_dev), C.c_void_p(gyro_dev), C.c_void_p(acc_dev)),
                  "hv_ekf_predict_dev")

    @staticmethod
    def _pack_H(H, batch):
        H = np.asarray(H, np.float64)
        if H.ndim == 2:
            H = np.broadcast_to(H, (batch,) + H.shape)
        nr, l = H.shape[1:]
        return np.ascontiguousarray(np.transpose(H, (0, 2, 1))), nr, l     # [b][col][row] == column-major per filter

    def update(self, H, y, r_diag, active=None, normalize_all=False):
        Hc, nr, l = self._pack_H(H, self.batch)
        y = _f(np.broadcast_to(y, (self.batch, nr)))
        rd = _f(np.broadcast_to(r_diag, (self.batch,)))
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        self._chk(lib().hv_ekf_update(self._h, nr, l, _p(Hc, f64p), _p(y, f64p), _p(rd, f64p), _p(act, u8p),
                                      int(normalize_all)), "hv_ekf_update")
        self.ctx.synchronize()

    def visual_gate(self, H, v, r):
        Hc, nr, l = self._pack_H(H, self.batch)
        v = _f(np.broadcast_to(v, (self.batch, nr)))
        chi2, st = np.zeros(self.batch), np.zeros(self.batch, np.int32)
        self._chk(lib().hv_ekf_visual_gate(self._h, nr, l, _p(Hc, f64p), _p(v, f64p), r, _p(chi2, f64p), _p(st, i32p)),
                  "hv_ekf_visual_gate")
        return chi2, st

    def visual_update(self, H, v, r, active=None):
        Hc, nr, l = self._pack_H(H, self.batch)
        v = _f(np.broadcast_to(v, (self.batch, nr)))
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        self._chk(lib().hv_ekf_visual_update(self._h, nr, l, _p(Hc, f64p), _p(v, f64p), r, _p(act, u8p)),
                  "hv_ekf_visual_update")
        self.ctx.synchronize()

    def visual_prepare_dev(self, params: VuParams, n_poses, pose_index_dev, features_dev, velocities_dev, y_dev, H_dev, v_dev,
                           f_dev, pf_dev, status_dev, active_dev=0):
        """hv_ekf_visual_prepare_dev: device pointers (ints); y_dev / f_dev / active_dev may be 0."""
        p = [C.c_void_p(x) for x in (pose_index_dev, features_dev, velocities_dev, y_dev, H_dev, v_dev, f_dev, pf_dev, status_dev, active_dev)]
        self._chk(lib().hv_ekf_visual_prepare_dev(self._h, C.byref(params), n_poses, *p), "hv_ekf_visual_prepare_dev")

    def visual_track_dev(self, params: VuParams, n_poses, pose_index_dev, features_dev, velocities_dev, y_dev, r_gate, r_update,
                         status_dev, gate_status_dev, chi2_dev=0, pf_dev=0):
        a = [C.c_void_p(x) for x in (pose_index_dev, features_dev, velocities_dev, y_dev)]
        b = [C.c_void_p(x) for x in (status_dev, gate_status_dev, chi2_dev, pf_dev)]
        self._chk(lib().hv_ekf_visual_track_dev(self._h, C.byref(params), n_poses, *a, float(r_gate), float(r_update), *b),
                  "hv_ekf_visual_track_dev")

    def visual_track_hybrid_dev(self, params: VuParams, n_poses, pose_index_dev, features_dev, velocities_dev, y_dev, map_update_dev, map_offer_dev,
                                r_gate, r_update, status_dev, gate_status_dev, chi2_dev=0, pf_dev=0):
        """hv_ekf_visual_track_hybrid_dev: a track visit with hybrid-map tracks (map_update_dev) and map-point offers (map_offer_dev)."""
        a = [C.c_void_p(x) for x in (pose_index_dev, features_dev, velocities_dev, y_dev, map_update_dev, map_offer_dev)]
        b = [C.c_void_p(x) for x in (status_dev, gate_status_dev, chi2_dev, pf_dev)]
        self._chk(lib().hv_ekf_visual_track_hybrid_dev(self._h, C.byref(params), n_poses, *a, float(r_gate), float(r_update), *b),
                  "hv_ekf_visual_track_hybrid_dev")

    def visual_track_limited_dev(self, params: VuParams, n_poses, pose_index_dev, features_dev, velocities_dev, y_dev, r_gate, r_update,
                                 status_dev, gate_status_dev, success_counter_dev, max_successful, chi2_dev=0, pf_dev=0):
        a = [C.c_void_p(x) for x in (pose_index_dev, features_dev, velocities_dev, y_dev)]
        b = [C.c_void_p(x) for x in (status_dev, gate_status_dev, chi2_dev, pf_dev, success_counter_dev)]
        self._chk(lib().hv_ekf_visual_track_limited_dev(self._h, C.byref(params), n_poses, *a, float(r_gate), float(r_update), *b,
                                                        int(max_successful)), "hv_ekf_visual_track_limited_dev")

    def visual_track(self, params: VuParams, pose_index, features, velocities, y, r_gate, r_update):
        """hv_ekf_visual_track with numpy arrays [batch][...]: returns (status [batch][2], gate_status, chi2, pf)."""
        idx = np.ascontiguousarray(pose_index, np.int32).reshape(self.batch, -1)
        ft, vl, yy = _f(features), _f(velocities), _f(y)
        st, gs = np.zeros((self.batch, 2), np.int32), np.zeros(self.batch, np.int32)
        chi, pf = np.zeros(self.batch), np.zeros((self.batch, 3))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(lib().hv_ekf_visual_track(self._h, C.byref(params), idx.shape[1], vp(idx), vp(ft), vp(vl), vp(yy), float(r_gate),
                                            float(r_update), vp(st), vp(gs), vp(chi), vp(pf)), "hv_ekf_visual_track")
        return st, gs, chi, pf

    def visual_dev(self, nr, l, H_dev, v_dev, r, mode, chi2_dev=0, status_dev=0):
        self._chk(lib().hv_ekf_visual_dev(self._h, nr, l, C.c_void_p(H_dev), C.c_void_p(v_dev), r, mode,
                                          C.c_void_p(chi2_dev), C.c_void_p(status_dev)), "hv_ekf_visual_dev")

    def augment(self, discarded=None, active=None):
        d = np.ascontiguousarray(np.broadcast_to(discarded, (self.batch,)), np.int32) if discarded is not None else None
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        self._chk(lib().hv_ekf_augment(self._h, _p(d, i32p), _p(act, u8p)), "hv_ekf_augment")
        self.ctx.synchronize()

    def augment_dev(self, discarded_dev=0, active_dev=0):
        """hv_ekf_augment_dev: device arrays (or 0), asynchronous."""
        self._chk(lib().hv_ekf_augment_dev(self._h, C.c_void_p(discarded_dev), C.c_void_p(active_dev)), "hv_ekf_augment_dev")

    def symmetrize_augment_dev(self, discarded_dev=0, active_dev=0):
        """hv_ekf_symmetrize_augment_dev: symmetrize() + augment_dev() in one pass over P, asynchronous."""
        self._chk(lib().hv_ekf_symmetrize_augment_dev(self._h, C.c_void_p(discarded_dev), C.c_void_p(active_dev)), "hv_ekf_symmetrize_augment_dev")

    def visual_frame_dev(self, params: VuParams, n_tracks, n_poses, pose_index_dev, features_dev, velocities_dev, y_dev, r_gate, r_update,
                         status_dev, gate_status_dev, success_counter_dev, max_successful, chi2_dev=0, pf_dev=0):
        """hv_ekf_visual_frame_dev: the whole visual-update loop of a frame, track-major device arrays."""
        a = [C.c_void_p(x) for x in (pose_index_dev, features_dev, velocities_dev, y_dev)]
        b = [C.c_void_p(x) for x in (status_dev, gate_status_dev, chi2_dev, pf_dev, success_counter_dev)]
        self._chk(lib().hv_ekf_visual_frame_dev(self._h, C.byref(params), int(n_tracks), int(n_poses), *a, float(r_gate), float(r_update), *b,
                                                int(max_successful)), "hv_ekf_visual_frame_dev")

    def visual_frame_ragged_dev(self, params: VuParams, n_tracks, n_poses_max, n_poses_dev, pose_index_dev, features_dev, velocities_dev, y_dev,
                                r_gate, r_update, status_dev, gate_status_dev, success_counter_dev, max_successful, chi2_dev=0, pf_dev=0):
        """hv_ekf_visual_frame_ragged_dev: per (visit, filter) track lengths n_poses_dev [n_tracks][batch] (< 2: no track)."""
        a = [C.c_void_p(x) for x in (n_poses_dev, pose_index_dev, features_dev, velocities_dev, y_dev)]
        b = [C.c_void_p(x) for x in (status_dev, gate_status_dev, chi2_dev, pf_dev, success_counter_dev)]
        self._chk(lib().hv_ekf_visual_frame_ragged_dev(self._h, C.byref(params), int(n_tracks), int(n_poses_max), *a, float(r_gate),
                                                       float(r_update), *b, int(max_successful)), "hv_ekf_visual_frame_ragged_dev")

    def visual_frame_batch_dev(self, params: VuParams, n_tracks, n_poses_max, n_poses_dev, pose_index_dev, features_dev, velocities_dev, y_dev,
                               r_gate, r_update, status_dev, gate_status_dev, success_counter_dev, max_successful, max_update_rows=0, chi2_dev=0, pf_dev=0):
        """hv_ekf_visual_frame_batch_dev: the frame loop with batchVisualUpdate (inlier blocks applied as one update per batch)."""
        a = [C.c_void_p(x) for x in (n_poses_dev, pose_index_dev, features_dev, velocities_dev, y_dev)]
        b = [C.c_void_p(x) for x in (status_dev, gate_status_dev, chi2_dev, pf_dev, success_counter_dev)]
        self._chk(lib().hv_ekf_visual_frame_batch_dev(self._h, C.byref(params), int(n_tracks), int(n_poses_max), *a, float(r_gate),
                                                      float(r_update), *b, int(max_successful), int(max_update_rows)), "hv_ekf_visual_frame_batch_dev")

    def visual_frame(self, params: VuParams, pose_index, features, velocities, y, r_gate, r_update, max_successful):
        """hv_ekf_visual_frame with numpy arrays [n_tracks][batch][...]: returns (status [K][B][2], gate_status [K][B], chi2, pf, applied [B])."""
        idx = np.ascontiguousarray(pose_index, np.int32)
        K = idx.shape[0]
        idx = idx.reshape(K, self.batch, -1)
        ft, vl, yy = _f(features), _f(velocities), _f(y)
        st, gs = np.zeros((K, self.batch, 2), np.int32), np.zeros((K, self.batch), np.int32)
        chi, pf, cnt = np.zeros((K, self.batch)), np.zeros((K, self.batch, 3)), np.zeros(self.batch, np.int32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._chk(lib().hv_ekf_visual_frame(self._h, C.byref(params), K, idx.shape[2], vp(idx), vp(ft), vp(vl), vp(yy), float(r_gate),
                                            float(r_update), vp(st), vp(gs), vp(chi), vp(pf), vp(cnt), int(max_successful)), "hv_ekf_visual_frame")
        return st, gs, chi, pf, cnt

    def undo_augment(self, active=None):
        act = np.ascontiguousarray(active, np.uint8) if active is not None else None
        self._chk(lib().hv_ekf_undo_augment(self._h, _p(act, u8p)), "hv_ekf_undo_augment")
        self.ctx.synchronize()

    def symmetrize(self):
        self._chk(lib().hv_ekf_symmetrize(self._h), "hv_ekf_symmetrize")

    def normalize_quaternions(self, only_current=False):
        self._chk(lib().hv_ekf_normalize_quaternions(self._h, int(only_current)), "hv_ekf_normalize_quaternions")

    def transform(self, b, pC, qC, tr):
        self._chk(lib().hv_ekf_transform(self._h, b, _p(_f(pC), f64p), _p(_f(qC), f64p), _p(_f(tr), f64p)),
                  "hv_ekf_transform")
        self.ctx.synchronize()
