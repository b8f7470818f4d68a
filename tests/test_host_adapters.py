"""C++ host adapters (hybvio_amd/host): tracker::ImagePyramid::Factory / OpticalFlow / odometry::EKF
re-implemented on the C ABI with the reference's interfaces.

CPU part: the adapter library and its test program compile and link with plain g++ against
libhybvio_hip.so (no HIP headers: they are a pure C-ABI client).
GPU part: tests/cpp/test_host_adapters.cpp runs the reference's EKF tests (test/ekf.cpp) through
EKF::buildHip and drives the tracker adapters like image.cpp:87-106; its tracker output is compared
with the CPU oracle here.
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from hybvio_amd import build, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ekf_reference_fixtures.npz")


def test_host_library_and_test_program_build():
    lib, exe = build.build_host()
    assert os.path.exists(lib) and os.access(exe, os.X_OK)
    syms = subprocess.check_output(["nm", "-D", "--defined-only", "-C", lib], text=True)
    for name in ("hybvio::tracker::ImagePyramid::Factory::buildHip", "hybvio::tracker::OpticalFlow::buildHip",
                 "hybvio::odometry::EKF::buildHip", "hybvio::tracker::FeatureDetector::buildHip",
                 "hybvio::tracker::Undistorter::buildRectifiedHip", "hybvio::tracker::rot_ransac::RotRansac::buildHip"):
        assert name in syms, name
    needed = subprocess.check_output(["readelf", "-d", lib], text=True)
    assert "libhybvio_hip.so" in needed and "amdhip64" not in needed      # only the C ABI is linked


@pytest.mark.gpu
def test_reference_tests_through_the_cpp_adapters(oracle):
    _, exe = build.build_host()
    fx = np.load(GOLD)
    w, h, n = 320, 240, 96
    tex = synth.Texture.make(33)
    img0 = synth.render(tex, w, h, synth.Warp.make(0, 0, 0, w / 2, h / 2))
    img1 = synth.render(tex, w, h, synth.Warp.make(-0.4, 1.1, 2.3, w / 2, h / 2), noise_seed=5, noise_sigma=1.5)
    rng = np.random.default_rng(6)
    pts = np.concatenate([synth.grid_points(w, h, n - 16, margin=8, seed=9),
                          rng.uniform([-20, -20], [w + 20, h + 20], (16, 2)).astype(np.float32)])
    with tempfile.TemporaryDirectory() as d:
        np.savetxt(os.path.join(d, "dims.txt"), [w, h])
        for key, name in (("chi2_M", "chi2_M"), ("chi2_v", "chi2_v"), ("poses", "poses"), ("gyro", "gyro"),
                          ("acc", "acc"), ("P55", "P55"), ("m55", "m55")):
            np.savetxt(os.path.join(d, name + ".txt"), np.asarray(fx[key]).reshape(-1), fmt="%.17g")
        img0.tofile(os.path.join(d, "img0.raw"))
        img1.tofile(os.path.join(d, "img1.raw"))
        np.savetxt(os.path.join(d, "pts.txt"), pts.reshape(-1), fmt="%.9g")
        rgb = np.clip(img0[..., None].astype(np.int32) + rng.integers(-50, 51, (h, w, 3)), 0, 255).astype(np.uint8)
        rgb.tofile(os.path.join(d, "rgb0.raw"))
        cams = [195.2, 194.6, 156.3, 121.7, -0.28340811, 0.07395907, 0.0, 170.0, w * 0.5, h * 0.5]
        np.savetxt(os.path.join(d, "cameras.txt"), cams, fmt="%.17g")
        # f4: three frames of RANSAC input sharing one generator; frame 1 has no outliers (early exit of the reference loop)
        rcam = [458.654 * w / 752, 457.296 * w / 752, 367.215 * w / 752, 248.375 * w / 752, -0.28340811, 0.07395907, 0.0]
        thr = float(np.float32((4.0 * min(w, h) / 720.0) ** 2))
        np.savetxt(os.path.join(d, "ransac_camera.txt"), rcam + [thr], fmt="%.17g")
        ocam = oracle.Camera("pinhole", *rcam[:4], coeffs=rcam[4:])
        rsets = []
        for frame, (npts, nbad) in enumerate(((120, 30), (60, 0), (90, 20))):
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            Rt = np.eye(3) + np.sin(0.02) * K + (1 - np.cos(0.02)) * K @ K
            a = rng.uniform([30, 30], [w - 30, h - 30], (npts, 2)).astype(np.float32)
            b = np.array([ocam.ray_to_pixel(Rt @ ocam.pixel_to_ray(*p)[1])[1] for p in a]) + rng.normal(size=(npts, 2)) * (0.02 if nbad == 0 else 0.2)
            b = b.astype(np.float32)
            bad = rng.choice(npts, nbad, replace=False)
            b[bad] += (rng.uniform(6, 25, (nbad, 2)) * rng.choice([-1, 1], (nbad, 2))).astype(np.float32)
            rsets.append((a, b))
            np.savetxt(os.path.join(d, f"ransac_pts{frame}.txt"), np.hstack([a, b]).reshape(-1), fmt="%.9g")
        tri = np.load(os.path.join(os.path.dirname(__file__), "golden", "triangulation_reference_fixtures.npz"))
        for key, name in (("visual_poses", "visual_poses"), ("visual_uv", "visual_uv"), ("visual_pf_matlab", "visual_pf")):
            np.savetxt(os.path.join(d, name + ".txt"), np.asarray(tri[key]).reshape(-1), fmt="%.17g")
        r = subprocess.run([exe, d], capture_output=True, text=True, timeout=300)
        print(r.stdout, r.stderr)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "all host adapter tests passed" in r.stdout
        out = np.loadtxt(os.path.join(d, "flow_out.txt"))
        out2 = np.loadtxt(os.path.join(d, "flow_out_init2.txt"))
        g1 = np.fromfile(os.path.join(d, "gray1.raw"), np.uint8).reshape((h + 1) // 2, (w + 1) // 2)
        det = [np.loadtxt(os.path.join(d, f), ndmin=2).astype(np.float32)
               for f in ("detect_raw.txt", "detect_masked.txt", "detect_raw_then_mask.txt")]
        ing = [np.fromfile(os.path.join(d, f), np.uint8).reshape(h, w)
               for f in ("ingest_gray.raw", "ingest_rect.raw", "ingest_rect_again.raw")]
        rout = open(os.path.join(d, "ransac_out.txt")).read().split("\n")
    # RotRansac::buildHip: three frames on one std::mt19937 vs the oracle consuming the same stream
    skip = 0
    for frame, (a, b) in enumerate(rsets):
        st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(a, b, ocam, ocam, oracle.mt19937_draws(4649, 200, skip=skip), thr)
        skip += used_o
        vals = rout[frame].split()
        assert int(vals[0]) == best_o
        np.testing.assert_allclose(np.array(vals[1:10], np.float64), R_o.reshape(-1), rtol=0, atol=1e-7)
        np.testing.assert_array_equal(np.array(vals[10:], np.int32), st_o)
    assert skip < 600 and int(rout[3]) == int(oracle.mt19937_draws(4649, 1, skip=skip)[0])    # frame 1 stopped early; the stream stays in step
    # Image::Factory::build on a colour frame and Undistorter::buildRectifiedHip vs the oracle
    gray = oracle.color_to_gray(rgb)
    np.testing.assert_array_equal(ing[0], gray)
    pix, valid = oracle.undistort_map(oracle.Camera("pinhole", cams[7], cams[7], cams[8], cams[9]),
                                      oracle.Camera("pinhole", *cams[:4], coeffs=cams[4:7]), w, h)
    np.testing.assert_array_equal(ing[1], oracle.undistort_apply(gray, pix, valid))
    np.testing.assert_array_equal(ing[2], ing[1])
    p0, p1 = oracle.Pyramid(img0), oracle.Pyramid(img1)
    o_xy, o_st = oracle.optical_flow_compute(p0, p1, pts)
    np.testing.assert_array_equal(out[:, 2].astype(int), o_st)
    assert np.abs(out[:, :2] - o_xy).max() <= 1e-3
    o_xy2, o_st2 = oracle.optical_flow_compute(p0, p1, pts, corners=o_xy, max_iter=2)
    np.testing.assert_array_equal(out2[:, 2].astype(int), o_st2)
    assert np.abs(out2[:, :2] - o_xy2).max() <= 1e-3
    np.testing.assert_array_equal(g1, p1.gray(1))
    # FeatureDetector::buildHip: detect / applyMinDistance vs the oracle's restatement of the reference flow
    pts32 = pts.astype(np.float32)
    np.testing.assert_array_equal(det[0], oracle.gftt_detect(img0, mask_radius=0, min_distance=20))
    np.testing.assert_array_equal(det[1], oracle.gftt_detect(img1, prev=pts32, mask_radius=20, min_distance=20, max_tracks=60))
    np.testing.assert_array_equal(det[2], oracle.apply_min_distance(det[0], pts32, 20, 60))
