"""CPU checks of the ingest oracle (oracle/ingest_oracle.c). The reference holds no test
or golden vector for this stage ("parity unpinned"), so the restatement is checked against the properties the
reference code implies and against an independent numpy evaluation of the same arithmetic."""
import os

import numpy as np
import pytest

EUROC = dict(fx=458.654, fy=457.296, ppx=367.215, ppy=248.375)
RADIAL = [-0.28340811, 0.07395907, 0.0]
FISH = [0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182]


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
            @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))


def cameras(oracle):
    return {
        "pinhole": oracle.Camera("pinhole", **EUROC),
        "radial": oracle.Camera("pinhole", **EUROC, coeffs=RADIAL),
        "radial_rot": oracle.Camera("pinhole", **EUROC, coeffs=RADIAL, rotation=rot(0.01, -0.02, 0.005)),
        "fisheye": oracle.Camera("fisheye", 190.97, 190.97, 254.93, 256.89, coeffs=FISH, max_valid_fov_deg=175.0),
        "fisheye_plain": oracle.Camera("fisheye", 190.97, 190.97, 254.93, 256.89),
    }


@pytest.mark.parametrize("name", ["pinhole", "radial", "radial_rot", "fisheye", "fisheye_plain"])
def test_pixel_ray_round_trip(oracle, name):
    """camera.cpp: rayToPixel(pixelToRay(p)) == p (Newton tolerance: 1e-5 normalised / 0.01 px)."""
    cam = cameras(oracle)[name]
    rng = np.random.default_rng(5)
    lo, hi = ([60, 60], [450, 450]) if name.startswith("fisheye") else ([40, 40], [700, 440])   # inside the valid field of view
    for p in rng.uniform(lo, hi, (200, 2)):
        ok, ray = cam.pixel_to_ray(*p)
        assert ok and abs(np.linalg.norm(ray) - 1) < 1e-12
        ok2, q = cam.ray_to_pixel(ray)
        assert ok2 and np.abs(q - p).max() < (0.02 if name.startswith("fisheye") else 5e-3), (p, q)


def test_rays_behind_the_camera_and_outside_the_valid_fov_fail(oracle):
    cams = cameras(oracle)
    assert not cams["pinhole"].ray_to_pixel([0.1, 0.1, -1.0])[0]           # camera.cpp:184
    assert not cams["fisheye"].ray_to_pixel([0.1, 0.1, 0.0])[0]            # camera.cpp:383
    narrow = oracle.Camera("fisheye", 190.97, 190.97, 254.93, 256.89, coeffs=FISH, max_valid_fov_deg=90.0)
    assert narrow.ray_to_pixel([0.0, 0.5, 1.0])[0] and not narrow.ray_to_pixel([0.0, 1.2, 1.0])[0]   # :388
    assert not narrow.pixel_to_ray(254.93 + 400.0, 256.89)[0]              # r > maxValidR: camera.cpp:361-364


def test_pinhole_model_against_direct_formulas(oracle):
    cam = cameras(oracle)["radial"]
    ray = np.array([0.21, -0.13, 1.0])
    x, y = ray[0] / ray[2], ray[1] / ray[2]
    r2 = x * x + y * y
    th = 1 + r2 * (RADIAL[0] + r2 * (RADIAL[1] + r2 * RADIAL[2]))
    ok, pix = cam.ray_to_pixel(ray)
    assert ok and np.allclose(pix, [EUROC["fx"] * x * th + EUROC["ppx"], EUROC["fy"] * y * th + EUROC["ppy"]], rtol=0, atol=1e-12)


def numpy_remap(img, pix, valid):
    """Independent evaluation of undistorter.cpp:84-110 (float32 taps and weights, the flat-memory rule for the tap
    one past a row, round through double)."""
    h, w = img.shape
    flat = np.concatenate([img.reshape(-1).astype(np.float32), np.zeros(w + 2, np.float32)])
    px, py = pix[..., 0], pix[..., 1]
    ok = (valid != 0) & (px >= 0) & (px < w) & (py >= 0) & (py < h)
    x0 = np.floor(np.where(ok, px, 0)).astype(np.int64); y0 = np.floor(np.where(ok, py, 0)).astype(np.int64)
    xf = (np.where(ok, px, 0) - x0).astype(np.float32); yf = (np.where(ok, py, 0) - y0).astype(np.float32)
    one = np.float32(1)
    out = np.zeros((h, w), np.float32)
    for iy in range(2):
        wy = yf if iy else one - yf
        for ix in range(2):
            wx = xf if ix else one - xf
            out = out + flat[(y0 + iy) * w + x0 + ix] * wx * wy
    return np.where(ok, (out.astype(np.float64) + 0.5).astype(np.int64), 0).astype(np.uint8)


@pytest.mark.parametrize("orig", ["radial", "fisheye", "radial_rot"])
def test_remap_matches_independent_numpy_evaluation(oracle, orig):
    w, h = 376, 240
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    cams = cameras(oracle)
    src = {"radial": oracle.Camera("pinhole", 229.3, 228.6, 183.6, 124.2, coeffs=RADIAL),
           "radial_rot": oracle.Camera("pinhole", 229.3, 228.6, 183.6, 124.2, coeffs=RADIAL, rotation=rot(0.02, 0.01, -0.03)),
           "fisheye": oracle.Camera("fisheye", 95.5, 95.5, 187.5, 120.4, coeffs=FISH, max_valid_fov_deg=170.0)}[orig]
    del cams
    rect = oracle.mono_rectified_camera(w, h, 180.0, zoom=0.8)
    pix, valid = oracle.undistort_map(rect, src, w, h)
    out = oracle.undistort_apply(img, pix, valid)
    assert np.array_equal(out, numpy_remap(img, pix, valid))
    inside = (pix[..., 0] >= 0) & (pix[..., 0] < w) & (pix[..., 1] >= 0) & (pix[..., 1] < h)
    assert inside.mean() > 0.3 and (out[~inside] == 0).all()               # where zoom 0.8 looks past the original frame: zeros
    assert orig == "fisheye" or inside.mean() < 1.0


def test_identity_integer_shift_and_half_pixel_maps(oracle):
    w, h = 64, 48
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    ones = np.ones((h, w), np.uint8)
    assert np.array_equal(oracle.undistort_apply(img, np.dstack([xx, yy]), ones), img)
    out = oracle.undistort_apply(img, np.dstack([xx + 3, yy - 2]), ones)
    assert np.array_equal(out[2:, :-3], img[:-2, 3:]) and (out[:2] == 0).all() and (out[:, -3:] == 0).all()
    half = oracle.undistort_apply(img, np.dstack([xx + 0.5, yy]), ones)
    a, b = img[:, :-1].astype(np.float32), img[:, 1:].astype(np.float32)
    assert np.array_equal(half[:, :-1], ((a * np.float32(.5) + b * np.float32(.5)).astype(np.float64) + 0.5).astype(np.uint8))
    # the tap one past the end of a row is the first pixel of the next row (continuous cv::Mat), 0 after the last row
    nxt = np.concatenate([img[1:, 0], [0]]).astype(np.float32)
    assert np.array_equal(half[:, -1], ((img[:, -1].astype(np.float32) * np.float32(.5) + nxt * np.float32(.5)).astype(np.float64) + 0.5).astype(np.uint8))
    assert (oracle.undistort_apply(img, np.dstack([xx, yy]), np.zeros((h, w), np.uint8)) == 0).all()


def test_color_to_gray_values(oracle):
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert oracle.color_to_gray(px).tolist() == [[255, 0, 76, 150, 29, 18]]
    rng = np.random.default_rng(9)
    rgba = rng.integers(0, 256, (31, 45, 4), dtype=np.uint8)
    g = oracle.color_to_gray(rgba)
    assert np.array_equal(g, oracle.color_to_gray(rgba[..., :3]))          # the 4th coefficient is 0 (image.cpp:358-360)
    f = rgba[..., :3].astype(np.float32)
    want = (((np.float32(.299) * f[..., 0] + np.float32(.587) * f[..., 1]) + np.float32(.114) * f[..., 2]) + np.float32(.5)).astype(np.int32)
    assert np.array_equal(g, want.astype(np.uint8))
    assert np.abs(g.astype(np.float64) - (0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2])).max() <= 0.5 + 1e-4


def test_golden_fixture(oracle):
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ingest_golden.npz"))
    rect = oracle.mono_rectified_camera(int(gold["w"]), int(gold["h"]), float(gold["rect_focal"]), zoom=float(gold["zoom"]))
    cam = oracle.Camera("pinhole", *gold["intrinsics"], coeffs=gold["coeffs"])
    pix, valid = oracle.undistort_map(rect, cam, int(gold["w"]), int(gold["h"]))
    assert np.array_equal(pix, gold["pix"]) and np.array_equal(valid, gold["valid"])
    gray = oracle.color_to_gray(gold["rgb"])
    assert np.array_equal(gray, gold["gray"])
    assert np.array_equal(oracle.undistort_apply(gray, pix, valid), gold["rectified"])


# ---- the camera models are pinned by the reference's own tests (test/camera.cpp), values copied from there ----
@pytest.mark.parametrize("distorted", [False, True])
def test_reference_fisheye_camera_case(oracle, distorted):
    """test/camera.cpp:7-81 "fisheye camera" (coefficients "from a RealSense camera")."""
    coeff = [-0.00200599804520607, 0.03895416110754013, -0.03715667128562927, 0.0061612860299646854] if distorted else []
    cam = oracle.Camera("fisheye", 10, 11, 5, 5.5, coeffs=coeff)
    ok, axis = cam.pixel_to_ray(5, 5.5)
    assert np.linalg.norm(axis[:2]) < 1e-6 and abs(axis[2] - 1) < 1e-6
    v = np.array([1.0, -2.0, 3.0])
    ok, p = cam.ray_to_pixel(v)
    assert ok and p[0] > 5 and p[1] < 5.5
    assert cam.pixel_to_ray(*p)[0] and not cam.pixel_to_ray(1000, 10000)[0]               # isValidPixel (camera.cpp:400-403)
    ok, proj = cam.pixel_to_ray(*p)
    assert abs(v[2] / np.linalg.norm(v) - proj[2]) < 1e-6


def test_reference_pinhole_matlab_projection(oracle):
    """test/camera.cpp:83-114: ip = (235, 695) "given by our matlab function"."""
    cam = oracle.Camera("pinhole", 1000, 1000, 360, 640)
    ok, ip = cam.ray_to_pixel([-0.25, 0.11, 2])
    assert ok and np.abs(ip - [235, 695]).sum() < 1e-5


def test_reference_distorted_pinhole_case(oracle):
    """test/camera.cpp:116-168: ray0 <-> pixel0 through k1 k2 k3, both directions."""
    cam = oracle.Camera("pinhole", 1.31841527e+03, 1.31745365e+03, 9.49043714e+02, 5.31894317e+02,
                        coeffs=[0.20740335, -0.28361953, -0.10090323])
    ray0, pixel0 = np.array([0.26726124, 0.53452248, 0.80178373]), np.array([1393.07961912, 1419.31839027])
    ok, pixelp = cam.ray_to_pixel(ray0)
    ok2, rayp = cam.pixel_to_ray(*pixel0)
    assert ok and ok2
    assert np.abs(pixelp - pixel0).sum() < 1e-3 and np.abs(rayp - ray0).sum() < 1e-4
