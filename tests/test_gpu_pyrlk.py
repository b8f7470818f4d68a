"""GPU parity tests (run with -m gpu on an MI355X): HIP pyramid + LK through the C ABI vs the CPU
oracle on the same seeded inputs. Bars (BASELINE.json north_star / SURVEY.md 8d):
  pyramid gray + gradients ... bit-exact (integer)
  LK status / Feature::Status . bit-exact
  LK positions ............... |dxy| <= 1e-3 px (the integer-accumulator design makes them equal)
"""
import os

import numpy as np
import pytest

from hybvio_amd import capi, synth

pytestmark = pytest.mark.gpu
POS_TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "pyrlk_golden.npz")


def _ctx(w, h, **kw):
    return capi.Context(width=w, height=h, **kw)


def _check_pyramid(ctx, oracle, img, slot):
    ref = oracle.Pyramid(img)
    assert ref.levels == ctx.levels
    for l in range(ctx.levels):
        g, d = ctx.download(slot, l)
        np.testing.assert_array_equal(g, ref.gray(l), err_msg=f"gray level {l}")
        np.testing.assert_array_equal(d, ref.deriv(l), err_msg=f"gradient level {l}")
    return ref


@pytest.mark.parametrize("tail", ["fused", "per_level"])
@pytest.mark.parametrize("shape", [(480, 752), (720, 1280), (479, 641), (70, 101), (33, 65), (97, 64), (256, 260)])
def test_pyramid_bit_exact(oracle, shape, tail, monkeypatch):
    # levels >= 2 are built by one LDS-resident launch for large batches and by per-level launches otherwise: force each
    monkeypatch.setenv("HV_PYR_TAIL", "1" if tail == "fused" else "0")
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    with _ctx(shape[1], shape[0]) as ctx:
        s = ctx.acquire()
        ctx.build(s, img)
        _check_pyramid(ctx, oracle, img, s)
        # strided host input (ROI of a wider buffer)
        wide = np.zeros((shape[0], shape[1] + 13), np.uint8)
        wide[:, :shape[1]] = img[::-1]
        s2 = ctx.acquire()
        rc = capi.lib().hv_pyramid_build(ctx._h, s2, wide.ctypes.data_as(capi.u8p), wide.strides[0])
        assert rc == 0
        ctx.synchronize()
        _check_pyramid(ctx, oracle, np.ascontiguousarray(img[::-1]), s2)


def test_pool_grows_on_demand_like_the_reference_allocator(oracle, seq752):
    """util::Allocator (allocator.hpp:55-67) hands out a new pyramid when every pooled one is referenced; the device pool
    doubles its slab instead of failing. Pyramids built before the growth (own level-0 copy AND caller-owned level 0) stay
    valid and keep their slot numbers; LK across the growth equals the oracle."""
    import torch
    left, right, _ = seq752
    pts = synth.grid_points(752, 480, 200, margin=4, seed=3)
    with _ctx(752, 480, pool_size=2) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        s0 = ctx.acquire(); ctx.build(s0, left[0])                      # own copy of level 0 inside the slab
        s1 = ctx.acquire()
        d_img = torch.from_numpy(left[1]).cuda()                        # level 0 stays in the caller's buffer
        ctx.build_batch_dev(1, torch.tensor([s1], dtype=torch.int32, device="cuda").data_ptr(), d_img.data_ptr(), 752 * 480, 752)
        more = [ctx.acquire() for _ in range(5)]                        # 2 -> 4 -> 8 slots
        assert len(set([s0, s1] + more)) == 7
        ctx.build(more[-1], right[0])
        r0 = _check_pyramid(ctx, oracle, left[0], s0)
        r1 = _check_pyramid(ctx, oracle, left[1], s1)
        _check_pyramid(ctx, oracle, right[0], more[-1])
        xy, st, _ = ctx.klt_track(s0, s1, pts)
        oxy, ost, _ = oracle.klt_track(r0, r1, pts)
        np.testing.assert_array_equal(st, ost)
        np.testing.assert_array_equal(xy[st > 0], oxy[ost > 0])


def test_pyramid_extreme_images(oracle):
    for img in (np.zeros((480, 752), np.uint8), np.full((480, 752), 255, np.uint8),
                (np.indices((480, 752)).sum(0) % 2 * 255).astype(np.uint8)):    # checkerboard: max |gradient|
        with _ctx(752, 480) as ctx:
            s = ctx.acquire()
            ctx.build(s, img)
            _check_pyramid(ctx, oracle, img, s)


def test_pyramid_batch_dev_path_and_pool(oracle, seq752):
    """Images resident in HBM, used in place as level 0; slots recycled like util::Allocator."""
    import torch
    left, right, _ = seq752
    imgs = np.concatenate([left, right])                       # 6 images
    with _ctx(752, 480, pool_size=8) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        slots = [ctx.acquire() for _ in range(8)]
        for s in slots[:2]:
            ctx.release(s)
        use = slots[2:]                                        # non-contiguous, non-zero-based
        d_imgs = torch.from_numpy(imgs).cuda()
        d_slots = torch.tensor(use, dtype=torch.int32, device="cuda")
        ctx.build_batch_dev(len(use), d_slots.data_ptr(), d_imgs.data_ptr(), 752 * 480, 752)
        ctx.synchronize()
        for i, s in enumerate(use):
            _check_pyramid(ctx, oracle, imgs[i], s)
        # size-independent property at full size: pyramid of a flipped image = flipped pyramid (gray),
        # and gradients negate along the flipped axis
        flipped = torch.flip(d_imgs[:1], dims=[2]).contiguous()
        s = ctx.acquire()
        ctx.build_batch_dev(1, torch.tensor([s], dtype=torch.int32, device="cuda").data_ptr(),
                            flipped.data_ptr(), 752 * 480, 752)
        ctx.synchronize()
        g0, d0 = ctx.download(use[0], 0)
        g1, d1 = ctx.download(s, 0)
        np.testing.assert_array_equal(g1, g0[:, ::-1])
        np.testing.assert_array_equal(d1[..., 0], -d0[:, ::-1, 0])
        np.testing.assert_array_equal(d1[..., 1], d0[:, ::-1, 1])


def _compare_klt(ctx, oracle, refp, refn, sp, sn, pts, guess=None, **kw):
    o_xy, o_st, o_err = oracle.klt_track(refp, refn, pts, next_pts=guess,
                                         **({"max_count": kw["max_iter_override"]} if "max_iter_override" in kw else {}))
    g_xy, g_st, g_err = ctx.klt_track(sp, sn, pts, next_xy=guess, **kw)
    np.testing.assert_array_equal(g_st, o_st)
    assert np.abs(g_xy - o_xy).max() <= POS_TOL, np.abs(g_xy - o_xy).max()
    np.testing.assert_array_equal(g_xy, o_xy)          # stronger than required: identical floats
    np.testing.assert_allclose(g_err, o_err, rtol=0, atol=0)
    return g_xy, g_st


def test_klt_parity_752_temporal_and_stereo(oracle, seq752):
    left, right, warps = seq752
    pts = synth.grid_points(752, 480, 200)
    with _ctx(752, 480) as ctx:
        s0, s1, sr = ctx.acquire(), ctx.acquire(), ctx.acquire()
        ctx.build(s0, left[0]); ctx.build(s1, left[1]); ctx.build(sr, right[1])
        r0, r1, rr = oracle.Pyramid(left[0]), oracle.Pyramid(left[1]), oracle.Pyramid(right[1])
        xy, st = _compare_klt(ctx, oracle, r0, r1, s0, s1, pts)
        gt = synth.true_flow(pts.astype(np.float64), warps[0], warps[1])
        assert st.mean() > 0.98 and np.median(np.linalg.norm(xy - gt, axis=1)[st > 0]) < 0.05
        _compare_klt(ctx, oracle, r1, rr, s1, sr, xy)                         # stereo: 20 px disparity, no guess
        _compare_klt(ctx, oracle, r0, r1, s0, s1, pts, guess=gt.astype(np.float32))   # predicted flow
        _compare_klt(ctx, oracle, r0, r1, s0, s1, pts, max_iter_override=1)
        _compare_klt(ctx, oracle, r0, r1, s0, s1, pts, max_iter_override=3)


def test_klt_parity_borders_failures_and_ragged_counts(oracle, seq752):
    left, _, _ = seq752
    rng = np.random.default_rng(11)
    with _ctx(752, 480) as ctx:
        s0, s2, sf = ctx.acquire(), ctx.acquire(), ctx.acquire()
        flat = np.full((480, 752), 128, np.uint8)
        ctx.build(s0, left[0]); ctx.build(s2, left[2]); ctx.build(sf, flat)
        r0, r2, rf = oracle.Pyramid(left[0]), oracle.Pyramid(left[2]), oracle.Pyramid(flat)
        # points on / beyond every border, with windows that leave the padded image
        pts = rng.uniform([-45, -45], [800, 530], (300, 2)).astype(np.float32)
        edge = np.array([[0, 0], [751, 479], [751.9, 0.1], [-15.5, 240], [766.4, 240], [376, -15.9],
                         [376, 494.9], [15, 15], [736, 464], [-31, -31], [782, 510]], np.float32)
        pts = np.concatenate([pts, edge])
        xy, st = _compare_klt(ctx, oracle, r0, r2, s0, s2, pts)
        assert 0 < st.sum() < len(pts)
        guess = (pts + rng.normal(0, 6.0, pts.shape)).astype(np.float32)       # bad guesses: tile re-staging path
        _compare_klt(ctx, oracle, r0, r2, s0, s2, pts, guess=guess)
        _compare_klt(ctx, oracle, rf, rf, sf, sf, pts[:64])                     # no texture: all lost
        _compare_klt(ctx, oracle, r0, rf, s0, sf, pts[:64])                     # texture -> flat
        for n in (1, 2, 63, 64, 65, 257):                                       # ragged point counts
            _compare_klt(ctx, oracle, r0, r2, s0, s2, pts[:n])
        # Feature::Status mapping incl. FLOW_OUT_OF_RANGE and the empty call
        o_xy, o_fs = oracle.optical_flow_compute(r0, r2, pts, corners=guess)
        g_xy, g_fs = ctx.optical_flow_compute(s0, s2, pts, corners=guess)
        np.testing.assert_array_equal(g_fs, o_fs)
        np.testing.assert_array_equal(g_xy, o_xy)
        assert {0, 2, 4} >= set(np.unique(g_fs)) and (g_fs == 4).any()
        e_xy, e_fs = ctx.optical_flow_compute(s0, s2, np.zeros((0, 2), np.float32))
        assert e_xy.shape == (0, 2) and e_fs.shape == (0,)


@pytest.mark.parametrize("shape", [(720, 1280), (150, 200), (70, 101)])
def test_klt_parity_other_sizes(oracle, shape):
    h, w = shape
    tex = synth.Texture.make(5)
    a = synth.render(tex, w, h, synth.Warp.make(0, 0, 0, w / 2, h / 2))
    b = synth.render(tex, w, h, synth.Warp.make(0.5, 2.2, -1.4, w / 2, h / 2), noise_seed=3, noise_sigma=2.0)
    n = 400 if w == 1280 else 64
    pts = synth.grid_points(w, h, n, margin=6, seed=2)
    with _ctx(w, h, max_tracks=n) as ctx:
        sa, sb = ctx.acquire(), ctx.acquire()
        ctx.build(sa, a); ctx.build(sb, b)
        _, st = _compare_klt(ctx, oracle, oracle.Pyramid(a), oracle.Pyramid(b), sa, sb, pts)
        assert st.mean() > 0.8


def test_klt_batch_dev_equals_single_calls(oracle, seq752):
    """Throughput path: several (prev,next) pairs x 200 points in ONE launch, all arrays in HBM."""
    import torch
    left, right, _ = seq752
    pts = synth.grid_points(752, 480, 200)
    with _ctx(752, 480, pool_size=8) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        imgs = np.concatenate([left, right])
        slots = [ctx.acquire() for _ in range(6)]
        d_imgs = torch.from_numpy(imgs).cuda()
        d_slots = torch.tensor(slots, dtype=torch.int32, device="cuda")
        ctx.build_batch_dev(6, d_slots.data_ptr(), d_imgs.data_ptr(), 752 * 480, 752)
        pairs = [(0, 1), (1, 2), (1, 4), (2, 5), (0, 2)]           # temporal and stereo pairs
        prev = torch.tensor([slots[a] for a, _ in pairs], dtype=torch.int32, device="cuda")
        nxt = torch.tensor([slots[b] for _, b in pairs], dtype=torch.int32, device="cuda")
        P = torch.from_numpy(np.tile(pts, (len(pairs), 1))).cuda()
        N = P.clone()
        S = torch.zeros(len(pairs) * 200, dtype=torch.uint8, device="cuda")
        E = torch.zeros(len(pairs) * 200, dtype=torch.float32, device="cuda")
        ctx.klt_track_batch_dev(len(pairs), prev.data_ptr(), nxt.data_ptr(), 200, P.data_ptr(), N.data_ptr(),
                                S.data_ptr(), E.data_ptr(), use_initial_flow=True)
        torch.cuda.synchronize()
        N, S, E = N.cpu().numpy().reshape(len(pairs), 200, 2), S.cpu().numpy().reshape(-1, 200), E.cpu().numpy().reshape(-1, 200)
        refs = [oracle.Pyramid(im) for im in imgs]
        for k, (a, b) in enumerate(pairs):
            o_xy, o_st, o_err = oracle.klt_track(refs[a], refs[b], pts, next_pts=pts)
            np.testing.assert_array_equal(S[k], o_st)
            np.testing.assert_array_equal(N[k], o_xy)
            np.testing.assert_array_equal(E[k], o_err)
        # ragged: the pairs of a batch do not hold the same number of points; the padding behind a pair's points is skipped
        counts = [200, 0, 37, 64, 1]
        d_counts = torch.tensor(counts, dtype=torch.int32, device="cuda")
        N2 = torch.full_like(P, -5.0)
        S2 = torch.full((len(pairs) * 200,), 9, dtype=torch.uint8, device="cuda")
        ctx.klt_track_batch_ragged_dev(len(pairs), prev.data_ptr(), nxt.data_ptr(), 200, d_counts.data_ptr(), P.data_ptr(), N2.data_ptr(),
                                       S2.data_ptr(), 0, use_initial_flow=False)
        torch.cuda.synchronize()
        N2, S2 = N2.cpu().numpy().reshape(len(pairs), 200, 2), S2.cpu().numpy().reshape(-1, 200)
        for k, (a, b) in enumerate(pairs):
            n = counts[k]
            if n:
                o_xy, o_st, _ = oracle.klt_track(refs[a], refs[b], pts[:n])
                np.testing.assert_array_equal(S2[k, :n], o_st)
                np.testing.assert_array_equal(N2[k, :n], o_xy)
            assert not S2[k, n:].any()                                      # padding: status 0 (positions unspecified)


def test_golden_fixture_on_gpu():
    """Committed fixture (tests/golden/pyrlk_golden.npz): needs no oracle build on the GPU box."""
    g = np.load(GOLDEN)
    h, w = g["img0"].shape
    with _ctx(w, h) as ctx:
        assert ctx.levels == int(g["levels"])
        s0, s1 = ctx.acquire(), ctx.acquire()
        ctx.build(s0, g["img0"]); ctx.build(s1, g["img1"])
        for l in range(ctx.levels):
            gg, dd = ctx.download(s0, l)
            np.testing.assert_array_equal(gg, g[f"gray{l}"])
            np.testing.assert_array_equal(dd, g[f"deriv{l}"])
        xy, st, err = ctx.klt_track(s0, s1, g["pts"])
        np.testing.assert_array_equal(st, g["status"])
        assert np.abs(xy - g["next"]).max() <= POS_TOL
        np.testing.assert_array_equal(err, g["err"])
        fxy, fs = ctx.optical_flow_compute(s0, s1, g["pts"], corners=g["guess"])
        np.testing.assert_array_equal(fs, g["flow_status"])
        assert np.abs(fxy - g["flow_corners"]).max() <= POS_TOL


def test_full_size_property_identity_tracking(oracle, seq752):
    """Size-independent property: tracking an image onto itself returns the input points exactly
    with status 1 and err 0 wherever the window has texture."""
    left, _, _ = seq752
    pts = synth.grid_points(752, 480, 200)
    with _ctx(752, 480) as ctx:
        s = ctx.acquire()
        ctx.build(s, left[0])
        xy, st, err = ctx.klt_track(s, s, pts)
        assert st.all() and not err.any()
        assert np.abs(xy - pts).max() < 1e-4


@pytest.mark.parametrize("seed", range(12))
def test_randomised_shapes_motions_and_points(oracle, seed):
    """Seeded sweep over image sizes (odd / tiny / non-multiple-of-4 strides -> every border and unaligned path),
    textures from noise to smooth, motions up to several pixels, points anywhere incl. far outside, with and
    without an initial guess and with iteration caps: pyramid bit-exact, LK status and positions identical."""
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(48, 300)), int(rng.integers(48, 400))
    kind = seed % 3
    if kind == 0:
        a = rng.integers(0, 256, (h, w), dtype=np.uint8); b = np.roll(a, (int(rng.integers(-3, 4)), int(rng.integers(-3, 4))), (0, 1))
    else:
        tex = synth.Texture.make(seed)
        a = synth.render(tex, w, h, synth.Warp.make(0, 0, 0, w / 2, h / 2))
        b = synth.render(tex, w, h, synth.Warp.make(*rng.uniform(-1, 1, 1), *rng.uniform(-4, 4, 2), w / 2, h / 2),
                         noise_seed=seed, noise_sigma=float(rng.uniform(0, 3)))
    n = int(rng.integers(1, 150))
    pts = rng.uniform([-40, -40], [w + 40, h + 40], (n, 2)).astype(np.float32)
    pts[: n // 2] = rng.uniform([0, 0], [w, h], (n // 2, 2)).astype(np.float32)
    with _ctx(w, h) as ctx:
        sa, sb = ctx.acquire(), ctx.acquire()
        ctx.build(sa, a); ctx.build(sb, b)
        ra = _check_pyramid(ctx, oracle, a, sa)
        rb = _check_pyramid(ctx, oracle, b, sb)
        _compare_klt(ctx, oracle, ra, rb, sa, sb, pts)
        guess = (pts + rng.normal(0, 3.0, pts.shape)).astype(np.float32)
        _compare_klt(ctx, oracle, ra, rb, sa, sb, pts, guess=guess)
        _compare_klt(ctx, oracle, rb, ra, sb, sa, pts, max_iter_override=int(rng.integers(1, 6)))


@pytest.mark.parametrize("levels,max_iter,eps,min_eig", [(1, 20, 0.03, 1e-3), (2, 5, 0.1, 1e-2), (3, 30, 0.01, 1e-4),
                                                           (4, 1, 0.03, 1e-3), (4, 100, 0.001, 1e-5)])
def test_tracker_parameters_other_than_the_defaults(oracle, seq752, levels, max_iter, eps, min_eig):
    """pyrLKMaxLevel / pyrLKMaxIter / pyrLKEpsilon / pyrLKMinEigThreshold (parameter_definitions.c) other than
    HybVIO's defaults reach the kernel through hv_params."""
    left, _, _ = seq752
    rng = np.random.default_rng(levels * 10 + max_iter)
    pts = np.concatenate([synth.grid_points(752, 480, 120), rng.uniform([-20, -20], [770, 500], (40, 2)).astype(np.float32)])
    with _ctx(752, 480, levels=levels, max_iter=max_iter, eps=eps, min_eig=min_eig) as ctx:
        assert ctx.levels == levels
        s0, s1 = ctx.acquire(), ctx.acquire()
        ctx.build(s0, left[0]); ctx.build(s1, left[1])
        r0 = oracle.Pyramid(left[0], max_level=levels - 1); r1 = oracle.Pyramid(left[1], max_level=levels - 1)
        o_xy, o_st, o_err = oracle.klt_track(r0, r1, pts, max_level=levels - 1, max_count=max_iter, eps=eps, min_eig=min_eig)
        g_xy, g_st, g_err = ctx.klt_track(s0, s1, pts)
        np.testing.assert_array_equal(g_st, o_st)
        np.testing.assert_array_equal(g_xy, o_xy)
        np.testing.assert_array_equal(g_err, o_err)


def test_level0_used_in_place_with_a_padded_row_stride(oracle, seq752):
    """hv_pyramid_build_batch_dev uses the caller's images as level 0 without copying them: row stride > width
    (and an image stride that is not the packed size) must reach the pyramid, LK and GFTT kernels alike."""
    import torch
    left, right, _ = seq752
    imgs = np.stack([left[0], left[1], right[1]])
    stride, pitch = 800, 800 * 480 + 4096
    buf = np.full((3, pitch), 213, np.uint8)
    for i in range(3):
        buf[i, :800 * 480].reshape(480, 800)[:, :752] = imgs[i]
    with _ctx(752, 480, pool_size=4) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        d_buf = torch.from_numpy(buf).cuda()
        slots = [ctx.acquire() for _ in range(3)]
        d_slots = torch.tensor(slots, dtype=torch.int32, device="cuda")
        ctx.build_batch_dev(3, d_slots.data_ptr(), d_buf.data_ptr(), pitch, stride)
        ctx.synchronize()
        refs = [_check_pyramid(ctx, oracle, imgs[i], slots[i]) for i in range(3)]
        pts = synth.grid_points(752, 480, 150)
        xy, st = _compare_klt(ctx, oracle, refs[0], refs[1], slots[0], slots[1], pts)
        _compare_klt(ctx, oracle, refs[1], refs[2], slots[1], slots[2], xy)
        assert np.array_equal(ctx.gftt_detect(slots[1], prev=xy, mask_radius=30), oracle.gftt_detect(imgs[1], prev=xy, mask_radius=30))


def test_two_contexts_interleaved_are_independent(oracle):
    """Two sessions in one process (different resolutions, own streams and pools), their calls interleaved: each equals the
    oracle as if it ran alone (the library keeps no cross-context state)."""
    from hybvio_amd import synth
    sizes = [(376, 240), (330, 250)]
    seqs = [synth.stereo_sequence(60 + i, w, h, 2)[0] for i, (w, h) in enumerate(sizes)]
    pts = [synth.grid_points(w, h, 50, margin=10, seed=i) for i, (w, h) in enumerate(sizes)]
    ctxs = [capi.Context(width=w, height=h) for (w, h) in sizes]
    try:
        slots = [[c.acquire(), c.acquire()] for c in ctxs]
        for k in range(2):                       # frame k of context 0, then of context 1, ...
            for c, s, q in zip(ctxs, slots, seqs):
                c.build(s[k], q[k])
        out = [c.klt_track(s[0], s[1], p) for c, s, p in zip(ctxs, slots, pts)]
        det = [c.gftt_detect(s[1], mask_radius=20) for c, s in zip(ctxs, slots)]
    finally:
        for c in ctxs:
            c.close()
    for q, p, (xy, st, err), d in zip(seqs, pts, out, det):
        oxy, ost, _ = oracle.klt_track(oracle.Pyramid(q[0]), oracle.Pyramid(q[1]), p)
        assert np.array_equal(st, ost) and np.abs(xy - oxy)[ost == 1].max() <= 1e-3
        assert np.array_equal(d, oracle.gftt_detect(q[1], mask_radius=20))


def test_level0_gradients_formed_in_the_lk_kernel_equal_the_stored_plane(oracle, seq752, monkeypatch):
    """Default: the pyramid build does not write the gradient planes of levels 0 and 1, the LK kernel forms the Scharr gradients
    of its template window from the gray rows (klt.hip scharr_row). HV_GRAD_FROM_LEVEL=0 stores every plane as
    cv::buildOpticalFlowPyramid does (image_pyramid.cpp:40-48), =1 all but level 0. All equal the oracle bit for bit -- windows
    inside the image, on every border, beyond it -- and the read-back of the gradients (getGradientLevel) is served in every
    layout."""
    left, _, _ = seq752
    rng = np.random.default_rng(5)
    pts = np.concatenate([synth.grid_points(752, 480, 200, margin=2, seed=9),
                          rng.uniform([-40, -40], [790, 520], (200, 2)).astype(np.float32),
                          np.array([[0.5, 0.5], [1, 1], [16, 16], [735, 463], [736.2, 464.7], [750.5, 478.5], [17.3, 240], [734.9, 3]], np.float32)])
    r0, r2 = oracle.Pyramid(left[0]), oracle.Pyramid(left[2])
    out = {}
    for stored in ("0", "1", "2"):
        monkeypatch.setenv("HV_GRAD_FROM_LEVEL", stored)
        with _ctx(752, 480) as ctx:
            s0, s2 = ctx.acquire(), ctx.acquire()
            ctx.build(s0, left[0]); ctx.build(s2, left[2])
            _check_pyramid(ctx, oracle, left[0], s0)
            out[stored] = _compare_klt(ctx, oracle, r0, r2, s0, s2, pts)
    for k in ("1", "2"):
        np.testing.assert_array_equal(out["0"][0], out[k][0])
        np.testing.assert_array_equal(out["0"][1], out[k][1])
