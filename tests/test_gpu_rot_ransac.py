"""GPU parity tests of the 2-point rotation RANSAC through the C ABI against
oracle/rot_ransac_oracle.c: with the pinhole model feature statuses, bestInlierCount, the number of hypotheses the
reference loop visits and R (binary32) must be bit-identical; with the fisheye model (sin / cos / acos of two maths
libraries) statuses and counts must agree and R within 1e-6."""
import numpy as np
import pytest

from hybvio_amd import capi

pytestmark = pytest.mark.gpu
RADIAL = [-0.28340811, 0.07395907, 0.0]
FISH = [0.0035, 0.0007, -0.002, 0.0002]
THR = float(np.float32((4.0 * 480 / 720.0) ** 2))                          # ransac_pipeline.cpp:91-93 at 752x480


def _rot(rng, angle):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def _cams(oracle, kind):
    if kind == "pinhole":
        args = ("pinhole", 458.654, 457.296, 367.215, 248.375)
        return oracle.Camera(*args, coeffs=RADIAL), capi.camera_model(*args, coeffs=RADIAL)
    if kind == "plain":
        args = ("pinhole", 458.654, 457.296, 367.215, 248.375)
        return oracle.Camera(*args), capi.camera_model(*args)
    if kind == "rotated":
        args = ("pinhole", 430.0, 430.0, 376.0, 240.0)
        R = _rot(np.random.default_rng(0), 0.03)
        return oracle.Camera(*args, rotation=R), capi.camera_model(*args, rotation=R)
    args = ("fisheye", 280.0, 280.0, 376.0, 240.0)
    return oracle.Camera(*args, coeffs=FISH, max_valid_fov_deg=170.0), capi.camera_model(*args, coeffs=FISH, max_valid_fov_deg=170.0)


def _scene(ocam, rng, n, outliers, angle=0.03, noise=0.3):
    w, h = 752, 480
    R = _rot(rng, angle)
    c1 = rng.uniform([60, 60], [w - 60, h - 60], (n, 2)).astype(np.float32)
    c2 = np.zeros_like(c1)
    for i in range(n):
        _, ray = ocam.pixel_to_ray(*c1[i])
        _, pix = ocam.ray_to_pixel(R @ ray)
        c2[i] = pix + rng.normal(size=2) * noise
    bad = rng.choice(n, outliers, replace=False)
    c2[bad] += rng.uniform(8, 40, (outliers, 2)).astype(np.float32) * rng.choice([-1, 1], (outliers, 2))
    return c1, c2


def _pairs(draws, n):
    return (draws[:200].astype(np.uint64) % np.uint64(n)).astype(np.int32).reshape(100, 2)


def test_camera_model_init_matches_the_oracle_models(oracle):
    """The derived fields (fisheye Newton table, inverse camera matrix) equal what the oracle's constructor builds."""
    ocam, gcam = _cams(oracle, "fisheye")
    assert gcam.n_table == 50 and gcam.distortion_enabled == 1
    # the oracle exposes the table only through its behaviour: identical rays for pixels across the image on the device
    # are checked by the statuses below; here the closed-form parts
    assert abs(gcam.max_theta - 0.5 * 170.0 / 180.0 * np.pi) < 1e-15
    assert np.allclose(np.array(gcam.kinv).reshape(3, 3) @ np.array([[280.0, 0, 376.0], [0, 280.0, 240.0], [0, 0, 1]]), np.eye(3), atol=1e-15)
    with pytest.raises(capi.HvError):
        capi.camera_model("pinhole", 400.0, 400.0, 300.0, 200.0, coeffs=[0.1, 0.2])          # camera.cpp:163: 3 coefficients or none


# knob rot_ransac_threads: 0 = auto (a single set: the split form, 25 workgroups per set, r05), 1024 / 256 = one workgroup per set
@pytest.mark.parametrize("threads", [0, 1024, 256])
@pytest.mark.parametrize("kind", ["pinhole", "plain", "rotated"])
@pytest.mark.parametrize("n,outliers", [(200, 40), (400, 150), (37, 5), (2, 0), (3, 1), (1000, 300)])
def test_fit_bit_exact_pinhole(oracle, kind, n, outliers, threads):
    ocam, gcam = _cams(oracle, kind)
    rng = np.random.default_rng(n + outliers)
    c1, c2 = _scene(ocam, rng, n, outliers)
    draws = oracle.mt19937_draws(4649, 200, skip=7 * n)
    st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(c1, c2, ocam, ocam, draws, THR)
    with capi.Context(width=752, height=480) as ctx:
        ctx.set_knob("rot_ransac_threads", threads)
        st, R, best, visited = ctx.rot_ransac(c1, c2, gcam, gcam, _pairs(draws, n), THR)
        if threads == 0:                                     # the split form's ticket counters are back at zero: a second call gives the same
            st2, R2, best2, visited2 = ctx.rot_ransac(c1, c2, gcam, gcam, _pairs(draws, n), THR)
            assert np.array_equal(st2, st) and np.array_equal(R2.view(np.uint32), R.view(np.uint32)) and (best2, visited2) == (best, visited)
    assert np.array_equal(st, st_o) and best == best_o and 2 * visited == used_o
    assert np.array_equal(R.view(np.uint32), R_o.view(np.uint32))
    if n > 10:
        assert (st == 3).sum() >= outliers * 0.8 and (st == 0).sum() >= (n - outliers) * 0.9


def test_fit_fisheye(oracle):
    ocam, gcam = _cams(oracle, "fisheye")
    rng = np.random.default_rng(9)
    c1, c2 = _scene(ocam, rng, 300, 60)
    draws = oracle.mt19937_draws(4649, 200)
    st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(c1, c2, ocam, ocam, draws, THR)
    with capi.Context(width=752, height=480) as ctx:
        st, R, best, visited = ctx.rot_ransac(c1, c2, gcam, gcam, _pairs(draws, 300), THR)
    assert np.array_equal(st, st_o) and best == best_o and 2 * visited == used_o
    assert np.abs(R - R_o).max() < 1e-6


def test_control_flow_early_exit_repeated_indices_and_different_cameras(oracle):
    ocam, gcam = _cams(oracle, "pinhole")
    ocam2, gcam2 = _cams(oracle, "plain")                                    # camera1 != camera2 (per-frame intrinsics)
    rng = np.random.default_rng(4)
    c1, c2 = _scene(ocam, rng, 50, 0, noise=0.05)
    with capi.Context(width=752, height=480) as ctx:
        # every point an inlier: the reference loop stops at the first evaluated hypothesis; a repeated index consumes its draws
        draws = np.array([7, 57] + [3, 9] + [1, 2] * 99, np.uint32)
        st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(c1, c2, ocam, ocam, draws, THR)
        st, R, best, visited = ctx.rot_ransac(c1, c2, gcam, gcam, _pairs(draws, 50), THR)
        assert (used_o, best_o) == (4, 50) and 2 * visited == used_o and best == best_o and np.array_equal(st, st_o)
        assert np.array_equal(R.view(np.uint32), R_o.view(np.uint32))
        # no hypothesis at all (100 repeated indices): bestInds stays {0, 1}, bestInlierCount 0
        same = np.repeat(np.arange(100, dtype=np.uint32), 2)
        st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(c1, c2, ocam, ocam, same, THR)
        st, R, best, visited = ctx.rot_ransac(c1, c2, gcam, gcam, _pairs(same, 50), THR)
        assert best == best_o == 0 and visited == 100 and np.array_equal(st, st_o) and np.array_equal(R.view(np.uint32), R_o.view(np.uint32))
        # two different camera models for the two frames
        d3 = oracle.mt19937_draws(11, 200)
        st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(c1, c2, ocam, ocam2, d3, THR)
        st, R, best, visited = ctx.rot_ransac(c1, c2, gcam, gcam2, _pairs(d3, 50), THR)
        assert np.array_equal(st, st_o) and best == best_o and np.array_equal(R.view(np.uint32), R_o.view(np.uint32))


@pytest.mark.parametrize("threads", [0, 25, 1024])          # 7 sets: auto = the split form (175 workgroups on 256 CUs)
def test_batch_dev_ragged_sets(oracle, threads):
    import torch
    ocam, gcam = _cams(oracle, "pinhole")
    rng = np.random.default_rng(21)
    sizes = [200, 131, 2, 1, 0, 400, 64]                                      # sets with < 2 points are skipped (ransac_pipeline.cpp:209)
    S, M = len(sizes), 400
    c1 = np.zeros((S, M, 2), np.float32); c2 = np.zeros((S, M, 2), np.float32); pairs = np.zeros((S, 100, 2), np.int32)
    ref = []
    for s, n in enumerate(sizes):
        if n >= 2:
            a, b = _scene(ocam, rng, n, n // 5)
            c1[s, :n], c2[s, :n] = a, b
            d = oracle.mt19937_draws(4649 + s, 200)
            pairs[s] = _pairs(d, n)
            ref.append(oracle.rot_ransac_fit(a, b, ocam, ocam, d, THR))
        else:
            ref.append(None)
    with capi.Context(width=752, height=480) as ctx:
        dev = lambda x: torch.from_numpy(x).cuda()
        d_n, d_c1, d_c2, d_pairs = dev(np.array(sizes, np.int32)), dev(c1), dev(c2), dev(pairs)
        st = torch.full((S, M), -5, dtype=torch.int32, device="cuda")
        R = torch.zeros((S, 9), dtype=torch.float32, device="cuda"); summ = torch.full((S, 2), -1, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_knob("rot_ransac_threads", threads)
        for rep in range(2):                                    # (twice: the split form leaves its ticket counters at zero)
            st.fill_(-5)
            ctx.rot_ransac_batch_dev(S, M, d_n.data_ptr(), d_c1.data_ptr(), d_c2.data_ptr(), gcam, gcam, d_pairs.data_ptr(), THR,
                                     st.data_ptr(), R.data_ptr(), summ.data_ptr())
            torch.cuda.synchronize()
        st, R, summ = st.cpu().numpy(), R.cpu().numpy(), summ.cpu().numpy()
    for s, n in enumerate(sizes):
        if ref[s] is None:
            assert summ[s].tolist() == [0, 0] and (st[s] == -5).all()
            continue
        st_o, R_o, best_o, used_o = ref[s]
        assert np.array_equal(st[s, :n], st_o) and (st[s, n:] == -5).all()
        assert summ[s].tolist() == [best_o, used_o // 2]
        assert np.array_equal(R[s].view(np.uint32), R_o.reshape(-1).view(np.uint32))


def test_lk_output_feeds_ransac_on_the_device(oracle):
    """LK -> RANSAC without the host in between (ransac_pipeline.cpp:106-119): the batched tracker's device outputs go
    straight into hv_rot_ransac_lk_batch_dev, which keeps the TRACKED features in order, forms rng() % n itself and writes
    the statuses back at the original feature numbers. Must equal oracle LK -> host compaction -> oracle RANSAC."""
    import torch
    from hybvio_amd import synth
    w, h, npts, S = 376, 240, 120, 3
    cam_args = ("pinhole", 229.3, 228.6, 183.6, 124.2)
    ocam, gcam = oracle.Camera(*cam_args, coeffs=RADIAL), capi.camera_model(*cam_args, coeffs=RADIAL)
    thr = float(np.float32((4.0 * min(w, h) / 720.0) ** 2))
    seqs = [synth.stereo_sequence(40 + s, w, h, 2)[0] for s in range(S)]
    rng = np.random.default_rng(6)
    pts = np.stack([np.concatenate([synth.grid_points(w, h, npts - 12, margin=12, seed=s),
                                    rng.uniform([-15, -15], [w + 15, h + 15], (6, 2)).astype(np.float32),
                                    rng.uniform([-70, -70], [-50, -50], (6, 2)).astype(np.float32)])[rng.permutation(npts)] for s in range(S)])   # 6 far outside: lost
    draws = np.stack([oracle.mt19937_draws(4649 + s, 200) for s in range(S)])
    with capi.Context(width=w, height=h, pool_size=2 * S, max_tracks=npts) as ctx:
        prev = [ctx.acquire() for _ in range(S)]; cur = [ctx.acquire() for _ in range(S)]
        for s in range(S):
            ctx.build(prev[s], seqs[s][0]); ctx.build(cur[s], seqs[s][1])
        dev = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x, dt)).cuda()
        d_prev, d_cur, d_pts = dev(prev, np.int32), dev(cur, np.int32), dev(pts, np.float32)
        d_next = torch.zeros_like(d_pts); d_lk = torch.zeros((S, npts), dtype=torch.uint8, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.klt_track_batch_dev(S, d_prev.data_ptr(), d_cur.data_ptr(), npts, d_pts.data_ptr(), d_next.data_ptr(), d_lk.data_ptr(), 0, use_initial_flow=False)
        d_n, d_draws = dev(np.full(S, npts, np.int32), np.int32), dev(draws, np.uint32)
        st = torch.full((S, npts), -7, dtype=torch.int32, device="cuda")
        R = torch.zeros((S, 9), dtype=torch.float32, device="cuda"); summ = torch.zeros((S, 2), dtype=torch.int32, device="cuda")
        ctx.rot_ransac_lk_batch_dev(S, npts, d_n.data_ptr(), d_pts.data_ptr(), d_next.data_ptr(), d_lk.data_ptr(), 1, gcam, gcam,
                                    d_draws.data_ptr(), thr, st.data_ptr(), R.data_ptr(), summ.data_ptr())
        torch.cuda.synchronize()
        st, R, summ, lk = st.cpu().numpy(), R.cpu().numpy(), summ.cpu().numpy(), d_lk.cpu().numpy()
        nxt = d_next.cpu().numpy()
    for s in range(S):
        oxy, ost, _ = oracle.klt_track(oracle.Pyramid(seqs[s][0]), oracle.Pyramid(seqs[s][1]), pts[s])
        assert np.array_equal(lk[s], ost) and 20 < ost.sum() < npts                    # some features are lost: compaction matters
        keep = np.nonzero(ost == 1)[0]
        assert np.abs(nxt[s][keep] - oxy[keep]).max() <= 1e-3
        # the RANSAC input is the DEVICE LK output (positions equal the oracle's to 1e-3 px; use the device's own bits)
        st_o, R_o, best_o, used_o = oracle.rot_ransac_fit(pts[s][keep], nxt[s][keep], ocam, ocam, draws[s], thr)
        assert np.array_equal(st[s][keep], st_o) and (st[s][ost == 0] == -7).all()
        assert summ[s].tolist() == [best_o, used_o // 2] and np.array_equal(R[s].view(np.uint32), R_o.reshape(-1).view(np.uint32))
