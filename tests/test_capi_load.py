"""CPU-only checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol
that include/hybvio_hip.h declares (no compute is attempted without a GPU)."""
import ctypes as C
import os
import re

import pytest

from hybvio_amd import capi

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hybvio_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    L = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in hybvio_hip.h but not exported"
        assert s in capi.PROTOTYPES, f"{s} has no ctypes prototype"
    for s in capi.PROTOTYPES:
        assert s in syms, f"{s} bound in capi.py but not declared in the header"


def test_defaults_and_status_strings():
    L = capi.lib()
    assert L.hv_abi_version() == 4
    p = capi.Params()
    L.hv_default_params(C.byref(p))
    # codegen/parameter_definitions.c:262,336-344
    assert (p.width, p.height, p.levels, p.win, p.max_iter, p.max_tracks) == (752, 480, 4, 31, 20, 200)
    assert p.eps == 0.03 and p.min_eig == 1e-3
    assert L.hv_status_string(0) == b"ok"
    assert b"device" in L.hv_status_string(-3)


def test_create_rejects_bad_arguments_without_touching_the_device():
    L = capi.lib()
    h = C.c_void_p()
    assert L.hv_create(None, C.byref(h)) == -1
    p = capi.Params()
    L.hv_default_params(C.byref(p))
    p.win = 21
    assert L.hv_create(C.byref(p), C.byref(h)) == -2      # only the 31x31 window is implemented
    p.win, p.levels = 31, 9
    assert L.hv_create(C.byref(p), C.byref(h)) == -1


def test_lanes_reject_bad_arguments_and_a_missing_device():
    """hv_lanes_create (r04): argument checks come before any device work; without a GPU it reports HV_ERR_NO_DEVICE like hv_create."""
    import torch
    L = capi.lib()
    p = capi.Params()
    L.hv_default_params(C.byref(p))
    h = C.c_void_p()
    assert L.hv_lanes_create(C.byref(p), 0, C.byref(h)) == -1
    assert L.hv_lanes_create(C.byref(p), 9, C.byref(h)) == -1
    assert L.hv_lanes_create(None, 2, C.byref(h)) == -1
    assert L.hv_lanes_count(None) == -1 and not L.hv_lanes_ctx(None, 0) and not L.hv_get_stream(None)
    L.hv_lanes_destroy(None)
    if not torch.cuda.is_available():
        assert L.hv_lanes_create(C.byref(p), 2, C.byref(h)) == -3 and not h.value
        with pytest.raises(capi.HvError, match="no HIP device"):
            capi.Lanes(2)


def test_vu_params_defaults_keep_the_adaptive_thresholds_off():
    """ABI 3 fields of hv_vu_params: trackRmseThreshold -1 (no RMSE test), trackOutlierThresholdGrowthFactor 1 (parameter_definitions.c:21,27)."""
    vp = capi.vu_default_params()
    assert vp.trackRmseThreshold == -1.0 and vp.trackOutlierThresholdGrowthFactor == 1.0
    assert vp.triangulationGaussNewtonIterations == 10 and vp.useLinearTriangulation == 0


def test_r04_ekf_entries_reject_a_null_filter_before_any_device_work():
    """hv_ekf_symmetrize_augment_dev, hv_ekf_visual_frame_batch_dev, hv_ekf_visual_track_hybrid_dev (ABI 3, r04): a null filter handle is
    HV_ERR_INVALID, decided on the host."""
    L = capi.lib()
    vp = capi.vu_default_params()
    assert L.hv_ekf_symmetrize_augment_dev(None, None, None) == -1
    assert L.hv_ekf_visual_frame_batch_dev(None, C.byref(vp), 4, 6, None, None, None, None, None, 1.5, 0.05, None, None, None, None, None, 5, 0) == -1
    assert L.hv_ekf_visual_track_hybrid_dev(None, C.byref(vp), 6, None, None, None, None, None, None, 1.5, 0.05, None, None, None, None) == -1


def test_no_silent_cpu_fallback():
    """Without a GPU the product path must fail loudly (HV_ERR_NO_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.HvError, match="no HIP device"):
        capi.Context()
