"""GPU parity tests of image ingest through the C ABI against oracle/ingest_oracle.c:
the gray image that lands in level 0 of the slot, and every pyramid level built from it, must be bit-identical."""
import os

import numpy as np
import pytest

from hybvio_amd import capi, synth

pytestmark = pytest.mark.gpu
RADIAL = [-0.28340811, 0.07395907, 0.0]
FISH = [0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182]


def _colour(gray, ch, seed):
    rng = np.random.default_rng(seed)
    h, w = gray.shape
    img = np.clip(gray[..., None].astype(np.int32) + rng.integers(-60, 61, (h, w, ch)), 0, 255).astype(np.uint8)
    return img


def _check_pyramid(ctx, oracle, slot, gray):
    ref = oracle.Pyramid(gray)
    for l in range(ctx.levels):
        g, d = ctx.download(slot, l)
        assert np.array_equal(g, ref.gray(l)), f"gray level {l}"
        assert np.array_equal(d, ref.deriv(l)), f"gradient level {l}"


@pytest.mark.parametrize("shape", [(480, 752), (250, 330), (241, 323), (97, 131)])
@pytest.mark.parametrize("ch", [1, 3, 4])
def test_colour_to_gray_and_copy_bit_exact(oracle, shape, ch):
    h, w = shape
    base = synth.stereo_sequence(11, w, h, 1)[0][0]
    img = base if ch == 1 else _colour(base, ch, 7 + ch)
    want = img if ch == 1 else oracle.color_to_gray(img)
    with capi.Context(width=w, height=h) as ctx:
        s = ctx.acquire()
        ctx.ingest_build(s, img)
        _check_pyramid(ctx, oracle, s, want)


def _cams(oracle, w, h, kind):
    sc = w / 752.0
    if kind == "radial":
        cam = oracle.Camera("pinhole", 458.654 * sc, 457.296 * sc, 367.215 * sc, 248.375 * sc, coeffs=RADIAL)
        rect = oracle.mono_rectified_camera(w, h, 458.0 * sc, zoom=0.85)
    elif kind == "fisheye":
        cam = oracle.Camera("fisheye", 190.97 * sc * 1.4, 190.97 * sc * 1.4, 0.49 * w, 0.52 * h, coeffs=FISH, max_valid_fov_deg=150.0)
        rect = oracle.mono_rectified_camera(w, h, 150.0 * sc, zoom=1.0)
    else:   # stereo-rectified: undistorted pinhole with a rotation (StereoRectifier output, image.cpp:322-328)
        c, s_ = np.cos(0.02), np.sin(0.02)
        R = np.array([[c, 0, s_], [0, 1, 0], [-s_, 0, c]]) @ np.array([[1, 0, 0], [0, np.cos(-.01), -np.sin(-.01)], [0, np.sin(-.01), np.cos(-.01)]])
        cam = oracle.Camera("pinhole", 458.654 * sc, 457.296 * sc, 367.215 * sc, 248.375 * sc, coeffs=RADIAL)
        rect = oracle.Camera("pinhole", 430.0 * sc, 430.0 * sc, 0.5 * w, 0.5 * h, rotation=R)
    return rect, cam


@pytest.mark.parametrize("shape", [(480, 752), (241, 323)])
@pytest.mark.parametrize("kind", ["radial", "fisheye", "rotated"])
@pytest.mark.parametrize("ch", [1, 3])
def test_undistort_remap_bit_exact(oracle, shape, kind, ch):
    h, w = shape
    base = synth.stereo_sequence(13, w, h, 1)[0][0]
    img = base if ch == 1 else _colour(base, ch, 3)
    rect, cam = _cams(oracle, w, h, kind)
    pix, valid = oracle.undistort_map(rect, cam, w, h)
    gray = img if ch == 1 else oracle.color_to_gray(img)           # image.cpp:282-285: gray first, then the undistorter
    want = oracle.undistort_apply(gray, pix, valid)
    assert (want == 0).mean() < 0.9
    with capi.Context(width=w, height=h) as ctx:
        ctx.ingest_set_undistort_map(1, pix, valid)
        s = ctx.acquire()
        ctx.ingest_build(s, img, camera=1)
        _check_pyramid(ctx, oracle, s, want)
        # removing the table turns rectification off again; a camera without a table is refused
        ctx.ingest_set_undistort_map(1, None)
        with pytest.raises(capi.HvError):
            ctx.ingest_build(s, img, camera=1)
        with pytest.raises(capi.HvError):
            ctx.ingest_build(s, img, camera=0)


def test_edge_taps_and_invalid_pixels(oracle):
    """Source positions in the last column / row (the unchecked at() of undistorter.cpp:99) and failed camera calls."""
    h, w = 120, 172
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    pix = np.dstack([xx * (w - 0.25) / (w - 1), yy * (h - 0.25) / (h - 1)])       # reaches x0 = w-1 and y0 = h-1 with weight on the next tap
    pix[5, 7] = (-0.5, 3.0); pix[6, 7] = (w, 3.0); pix[7, 7] = (3.0, h); pix[8, 7] = (np.nan, 1.0)
    valid = np.ones((h, w), np.uint8); valid[20:30, 40:50] = 0
    want = oracle.undistort_apply(img, pix, valid)
    assert want[5, 7] == want[6, 7] == want[7, 7] == want[8, 7] == 0 and (want[20:30, 40:50] == 0).all()
    with capi.Context(width=w, height=h) as ctx:
        ctx.ingest_set_undistort_map(0, pix, valid)
        s = ctx.acquire()
        ctx.ingest_build(s, img, camera=0)
        g, _ = ctx.download(s, 0)
        assert np.array_equal(g, want)


@pytest.mark.parametrize("ch", [1, 3])
def test_maps_whose_tiles_do_not_fit_the_staging_buffer(oracle, ch):
    """A 4x minification and a transposing map: the source rectangle of a 64 x 16 output tile exceeds the LDS staging
    buffer, so the library must take its gather kernel -- same results."""
    h, w = 256, 320
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (h, w) if ch == 1 else (h, w, ch), dtype=np.uint8)
    gray = img if ch == 1 else oracle.color_to_gray(img)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    maps = [np.dstack([np.minimum(xx * 4.03, w + 5.0), np.minimum(yy * 4.01, h + 5.0)]),      # also runs past the frame
            np.dstack([yy * (w - 1.5) / (h - 1), xx * (h - 1.5) / (w - 1)])]
    with capi.Context(width=w, height=h) as ctx:
        s = ctx.acquire()
        for pix in maps:
            ctx.ingest_set_undistort_map(0, pix)
            ctx.ingest_build(s, img, camera=0)
            assert np.array_equal(ctx.download(s, 0)[0], oracle.undistort_apply(gray, pix, np.ones((h, w), np.uint8)))


@pytest.mark.parametrize("ch,camera", [(1, -1), (3, -1), (4, 0), (1, 1)])
def test_batch_dev_with_padded_rows(oracle, ch, camera):
    import torch
    h, w, n = 250, 330, 3
    frames = synth.stereo_sequence(17, w, h, n)[0]
    imgs = [f if ch == 1 else _colour(f, ch, i) for i, f in enumerate(frames)]
    row = (w * ch + 63) // 64 * 64 + 64
    host = np.full((n, h + 2, row), 0xAB, np.uint8)                               # padded rows, 2 spare rows per image
    for i, im in enumerate(imgs):
        host[i, :h, :w * ch] = im.reshape(h, w * ch)
    rect, cam = _cams(oracle, w, h, "radial")
    pix, valid = oracle.undistort_map(rect, cam, w, h)
    with capi.Context(width=w, height=h, pool_size=4) as ctx:
        if camera >= 0:
            ctx.ingest_set_undistort_map(camera, pix, valid)
        slots = [ctx.acquire() for _ in range(n)]
        dev = torch.from_numpy(host).cuda()
        sl = torch.tensor(slots[::-1], dtype=torch.int32, device="cuda")          # image i -> slots[n-1-i]
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.ingest_build_batch_dev(n, sl.data_ptr(), dev.data_ptr(), (h + 2) * row, row, ch, camera)
        dev.fill_(0)                                                               # the frames may be reused immediately
        torch.cuda.synchronize()
        for i, im in enumerate(imgs):
            gray = im if ch == 1 else oracle.color_to_gray(im)
            want = gray if camera < 0 else oracle.undistort_apply(gray, pix, valid)
            _check_pyramid(ctx, oracle, slots[n - 1 - i], want)
        with pytest.raises(capi.HvError):                                          # 4-byte alignment of base and strides
            ctx.ingest_build_batch_dev(n, sl.data_ptr(), dev.data_ptr() + 1, (h + 2) * row, row, ch, camera)
        with pytest.raises(capi.HvError):
            ctx.ingest_build_batch_dev(n, sl.data_ptr(), dev.data_ptr(), (h + 2) * row, w * ch - 4, ch, camera)
        with pytest.raises(capi.HvError):
            ctx.ingest_build_batch_dev(n, sl.data_ptr(), dev.data_ptr(), (h + 2) * row, row, 2, camera)


def test_golden_fixture(oracle):
    """Input/output pair frozen in tests/golden/ingest_golden.npz (made by make_ingest_golden.py)."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ingest_golden.npz"))
    w, h = int(gold["w"]), int(gold["h"])
    with capi.Context(width=w, height=h) as ctx:
        ctx.ingest_set_undistort_map(0, gold["pix"], gold["valid"])
        s = ctx.acquire()
        ctx.ingest_build(s, gold["rgb"])
        assert np.array_equal(ctx.download(s, 0)[0], gold["gray"])
        ctx.ingest_build(s, gold["rgb"], camera=0)
        assert np.array_equal(ctx.download(s, 0)[0], gold["rectified"])


def test_tracker_runs_on_ingested_frames(oracle):
    """The tracker downstream of ingest: LK between two rectified colour frames equals the oracle's LK on the
    oracle-rectified frames (status bit-exact, positions <= 1e-3 px)."""
    w, h = 376, 240
    left = synth.stereo_sequence(19, w, h, 2)[0]
    rect, cam = _cams(oracle, w, h, "radial")
    pix, valid = oracle.undistort_map(rect, cam, w, h)
    cols = [_colour(f, 3, i) for i, f in enumerate(left)]
    refs = [oracle.undistort_apply(oracle.color_to_gray(c), pix, valid) for c in cols]
    pts = synth.grid_points(w, h, 80, margin=30, seed=2)
    with capi.Context(width=w, height=h) as ctx:
        ctx.ingest_set_undistort_map(0, pix, valid)
        a, b = ctx.acquire(), ctx.acquire()
        ctx.ingest_build(a, cols[0], camera=0); ctx.ingest_build(b, cols[1], camera=0)
        xy, st = ctx.optical_flow_compute(a, b, pts)
    oxy, ost = oracle.optical_flow_compute(oracle.Pyramid(refs[0]), oracle.Pyramid(refs[1]), pts)
    assert np.array_equal(st, ost) and (st == 0).sum() > 40
    assert np.abs(xy - oxy)[st == 0].max() <= 1e-3
