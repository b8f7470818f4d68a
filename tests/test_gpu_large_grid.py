"""Parity at grids larger than the chip: every batched entry point of the headline step at a batch that
(a) selects the throughput instantiations the benchmark runs (vu_*_kernel_2percu, rot_ransac_kernel<256>, pyr_tail_kernel,
ekf_gate_stream_kernel, several rounds of workgroups per CU) and (b) puts DISTINCT data into every workgroup, checked against the
oracle (all of it where the oracle is fast enough, otherwise every distinct input once + bit-equality of its replicas, which sit at
unrelated grid positions).

Reference chain: optical_flow.cpp:46-49 (LK), image_pyramid.cpp:40-48, rot_ransac.cpp:41-120, backend.cpp:1012-1252 (visit loop),
ekf.cpp:787-844 (gate, update), :848-885 (augmentation), :320-514 (predict)."""
import numpy as np
import pytest

from hybvio_amd import capi, synth

pytestmark = pytest.mark.gpu
W, H, NPTS = 752, 480, 200


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def test_pyramids_and_lk_over_640_slots_and_320_pairs(oracle):
    """640 pyramid slots built by ONE batch call (the level-0 down-sample at 2 x 94 x 640 workgroups, the tail kernel at 640) and
    320 (prev, next) pairs x 200 points in ONE LK launch (64 000 workgroups, xcd_remap over all of them). 16 distinct image pairs
    with their own points, each replicated 20 times at permuted batch positions: the oracle checks every distinct pair, the
    replicas must be bit-identical to it."""
    import torch
    D, R, M = 16, 20, 32
    left, _, _ = synth.stereo_sequence(2024, W + 2 * M, H + 2 * M, 2)
    rng = np.random.default_rng(5)
    offs = rng.integers(0, 2 * M + 1, (D, 2))
    prev = np.stack([left[0][oy:oy + H, ox:ox + W] for ox, oy in offs])
    nxt = np.stack([left[1][oy:oy + H, ox:ox + W] for ox, oy in offs])
    pts = np.stack([np.concatenate([synth.grid_points(W, H, NPTS - 16, margin=20, seed=d),
                                    rng.uniform([-20, -20], [W + 20, H + 20], (16, 2)).astype(np.float32)]) for d in range(D)])   # + border points
    perm = rng.permutation(D * R)
    which = (np.arange(D * R) % D)[perm]                                                     # pair p shows distinct input which[p]
    n_pairs = D * R
    with capi.Context(width=W, height=H, pool_size=2 * n_pairs, max_tracks=NPTS, max_pairs=n_pairs) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        slots = np.array([ctx.acquire() for _ in range(2 * n_pairs)], np.int32)
        rng.shuffle(slots)                                                                    # slot numbers unrelated to batch positions
        s_prev, s_next = slots[:n_pairs], slots[n_pairs:]
        imgs = torch.from_numpy(np.concatenate([prev[which], nxt[which]])).cuda()            # [640, H, W] used in place as level 0
        d_slots = torch.from_numpy(np.concatenate([s_prev, s_next])).cuda()
        ctx.build_batch_dev(2 * n_pairs, d_slots.data_ptr(), imgs.data_ptr(), W * H, W)
        d_pts = torch.from_numpy(np.ascontiguousarray(pts[which])).cuda()
        d_out = torch.zeros_like(d_pts); d_st = torch.zeros((n_pairs, NPTS), dtype=torch.uint8, device="cuda")
        d_err = torch.zeros((n_pairs, NPTS), dtype=torch.float32, device="cuda")
        d_sp, d_sn = torch.from_numpy(s_prev).cuda(), torch.from_numpy(s_next).cuda()
        ctx.klt_track_batch_dev(n_pairs, d_sp.data_ptr(), d_sn.data_ptr(), NPTS, d_pts.data_ptr(), d_out.data_ptr(), d_st.data_ptr(),
                                d_err.data_ptr(), use_initial_flow=False)
        torch.cuda.synchronize()
        xy, st, err = d_out.cpu().numpy(), d_st.cpu().numpy(), d_err.cpu().numpy()
        lost = 0
        for d in range(D):
            rp, rn = oracle.Pyramid(prev[d]), oracle.Pyramid(nxt[d])
            oxy, ost, oerr = oracle.klt_track(rp, rn, pts[d])
            lost += int((ost == 0).sum())
            keep = ost == 1
            for p in np.flatnonzero(which == d):
                assert np.array_equal(st[p], ost), (d, p)
                assert np.array_equal(xy[p][keep], oxy[keep]) and np.array_equal(err[p][keep], oerr[keep]), (d, p)
            # the pyramids themselves: every level of the first and the last replica of this input, both slots
            for p in (np.flatnonzero(which == d)[[0, -1]]):
                for lv in range(ctx.levels):
                    for slot, ref in ((int(s_prev[p]), rp), (int(s_next[p]), rn)):
                        g, gr = ctx.download(slot, lv)
                        assert np.array_equal(g, ref.gray(lv)) and np.array_equal(gr, ref.deriv(lv)), (d, p, lv)
        assert 0 < lost < D * NPTS // 4


@pytest.mark.parametrize("n_sets,threads", [(80, 0), (80, 1024), (7, 256), (300, 0), (80, 25)])       # 25: the split form, 2000 workgroups
def test_rot_ransac_both_instantiations(oracle, n_sets, threads):
    """rot_ransac_kernel<256> (what more than 64 sets -- the benchmark's 1024 -- run) and <1024>, each forced at both batch sizes."""
    import torch
    from test_gpu_rot_ransac import _cams, _scene, _pairs, THR
    ocam, gcam = _cams(oracle, "pinhole")
    rng = np.random.default_rng(1000 + n_sets)
    M = 400
    sizes = rng.integers(2, M + 1, n_sets)
    sizes[:4] = [200, 2, 400, 3]
    c1 = np.zeros((n_sets, M, 2), np.float32); c2 = np.zeros_like(c1); pairs = np.zeros((n_sets, 100, 2), np.int32)
    ref = []
    distinct = min(n_sets, 48)                                                 # scenes are generated point by point through the oracle camera
    for s in range(n_sets):
        n = int(sizes[s])
        if s < distinct:
            a, b = _scene(ocam, rng, n, n // 5)
        else:                                                                  # a different crop of an earlier scene, its own draws
            src = s % distinct
            n = min(n, int(sizes[src])); sizes[s] = n
            a, b = c1[src, :n].copy(), c2[src, :n].copy()
        c1[s, :n], c2[s, :n] = a, b
        d = oracle.mt19937_draws(4649 + s, 200)
        pairs[s] = _pairs(d, n)
        ref.append(oracle.rot_ransac_fit(a, b, ocam, ocam, d, THR))
    with capi.Context(width=W, height=H) as ctx:
        if threads:
            ctx.set_knob("rot_ransac_threads", threads)
        dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
        d_n, d_c1, d_c2, d_pairs = dev(sizes.astype(np.int32)), dev(c1), dev(c2), dev(pairs)
        st = torch.full((n_sets, M), -5, dtype=torch.int32, device="cuda")
        R = torch.zeros((n_sets, 9), dtype=torch.float32, device="cuda"); summ = torch.full((n_sets, 2), -1, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.rot_ransac_batch_dev(n_sets, M, d_n.data_ptr(), d_c1.data_ptr(), d_c2.data_ptr(), gcam, gcam, d_pairs.data_ptr(), THR,
                                 st.data_ptr(), R.data_ptr(), summ.data_ptr())
        torch.cuda.synchronize()
        st, R, summ = st.cpu().numpy(), R.cpu().numpy(), summ.cpu().numpy()
    for s in range(n_sets):
        n = int(sizes[s])
        st_o, R_o, best_o, used_o = ref[s]
        assert np.array_equal(st[s, :n], st_o) and (st[s, n:] == -5).all(), s
        assert summ[s].tolist() == [best_o, used_o // 2], s
        assert np.array_equal(R[s].view(np.uint32), R_o.reshape(-1).view(np.uint32)), s


def _filters(oracle, rng, B, trail_len, npose, stereo):
    from test_gpu_visual_prepare import _random_tracks
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, trail_len, npose, stereo, bad_fraction=0.0)
    return T1, T2, means


@pytest.mark.parametrize("variant", ["default", "dense", "gate_own_launch"])
def test_frame_of_320_distinct_filters(oracle, variant):
    """hv_ekf_visual_frame_dev over 320 DISTINCT filters (sequential visit loop: 320 x 8 > 256), 10-pose stereo tracks = the
    benchmark's 40-row shape, followed by symmetrise, the Joseph-form augmentation with per-filter discard slots and 10 predicts in
    one launch: every filter against the oracle's sequence. 320 workgroups = more than one round of every EKF kernel; `default`
    runs vu_gate_kernel_2percu + the compact-H update (what the benchmark's 1024 filters run), `gate_own_launch`
    vu_compact_kernel_2percu + ekf_sparse_gate_kernel, `dense` vu_prepare_kernel_2percu + ekf_update_kernel<2,3> mode 3."""
    import torch
    from test_gpu_visual_prepare import _random_tracks, VARIANTS
    rng = np.random.default_rng(4242)
    B, trail_len, npose, K, quota, NIMU = 320, 20, 10, 8, 3, 10
    T1, T2, means = _filters(oracle, rng, B, trail_len, npose, True)
    tracks = [_random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.2, given_means=means)[3:] for _ in range(K)]
    ys = [t[1].reshape(B, -1) + 2e-3 * rng.normal(size=(B, t[1].shape[1] * 2)) for t in tracks]
    for k in range(K):
        ys[k][rng.uniform(size=B) < 0.5] += 3.0                                 # per-filter different inlier patterns
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    hanoi = rng.choice([16, 17, 18, 19], B).astype(np.int32)
    gyro, acc = rng.normal(0, 0.02, (NIMU, B, 3)), rng.normal(0, 0.05, (NIMU, B, 3)) + [0.0, 0.0, 9.819]
    with capi.Context(width=64, height=64) as ctx:
        for k_, v_ in VARIANTS[variant].items():
            ctx.set_knob(k_, v_)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            A = rng.normal(size=(o.n, o.n)) * 0.01
            P = o.P.copy() * 1e-6 + A @ A.T * 1e-3 + np.eye(o.n) * 1e-4
            o.set_state(means[b]); o.set_cov(P); o.set_first_sample_time(0.0)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        d = [dev(np.stack([t[0] for t in tracks]), np.int32), dev(np.stack([t[1] for t in tracks]), np.float64),
             dev(np.stack([t[2] for t in tracks]), np.float64), dev(np.stack(ys), np.float64)]
        st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
        chi = torch.zeros((K, B), dtype=torch.float64, device="cuda")
        counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_frame_dev(vp, K, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), r_gate, r_update,
                           st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota, chi2_dev=chi.data_ptr())
        g.symmetrize()
        d_drop = dev(hanoi, np.int32)
        g.augment_dev(d_drop.data_ptr())
        d_dt, d_gy, d_ac = dev(np.full((NIMU, B), 0.005), np.float64), dev(gyro, np.float64), dev(acc, np.float64)
        g.predict_n_dev(NIMU, d_dt.data_ptr(), d_gy.data_ptr(), d_ac.data_ptr())
        torch.cuda.synchronize()
        assert g.frame_error() == 0
        st, gs, chi, counts = st.cpu().numpy(), gs.cpu().numpy(), chi.cpu().numpy(), counter.cpu().numpy()
        applied = rejected = 0
        for b, o in enumerate(filters):
            done = 0
            for k in range(K):
                idx, feat, vel = tracks[k]
                if done >= quota:
                    assert st[k, b].tolist() == [-1, -1] and gs[k, b] == 1, (b, k)
                    continue
                ost, ops, opf, oH, of = oracle.visual_track_prepare(par, o.m.copy(), idx[b], T1, T2, feat[b], vel[b])
                assert st[k, b].tolist() == [ost, ops], (b, k)
                if (ost, ops) != (0, 0):
                    assert gs[k, b] == 1
                    continue
                status, chi2 = o.visual_track_outlier_check(oH, of, ys[k][b], r_gate)
                assert gs[k, b] == status, (b, k)
                assert abs(chi[k, b] - chi2) <= 1e-7 * max(1.0, abs(chi2)), (b, k, chi[k, b], chi2)
                if status == 0:
                    o.update_visual_track(oH, of, ys[k][b], r_update); done += 1
                else:
                    rejected += 1
            assert counts[b] == done
            applied += done
            o.maintain_psd()
            o.update_visual_pose_augmentation(int(hanoi[b]))
            for j in range(NIMU):
                o.predict(0.005 * (j + 1), gyro[j, b], acc[j, b])
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
        assert applied > B and rejected > B
        g.close()


@pytest.mark.parametrize("rows", [40, 16, 8])
def test_gate_only_and_gated_update_over_320_filters(oracle, rows):
    """hv_ekf_visual_dev at a batch that selects ekf_gate_stream_kernel (mode 0, > 256 filters) and runs ekf_update_kernel in more than
    one round (mode 2: update where the gate passes), dense random Jacobians, distinct filters."""
    import torch
    rng = np.random.default_rng(77 + rows)
    B = 320
    with capi.Context(width=64, height=64) as ctx:
        g = capi.EkfBatch(ctx, capi.ekf_default_params(), B)
        n = g.n
        Hs = rng.normal(size=(B, rows, n)); Hs[:, :, 100:] *= (rng.uniform(size=(B, 1, 1)) < 0.5)       # some truncated Jacobians
        vs = rng.normal(size=(B, rows)) * np.where(rng.uniform(size=(B, 1)) < 0.5, 0.05, 2.0)
        filters = []
        for b in range(B):
            o = oracle.Ekf()
            A = rng.normal(size=(n, n)) * 0.02
            P = o.P.copy() * 1e-6 + A @ A.T * 1e-3 + np.eye(n) * 1e-4
            m = o.m.copy(); m[:20] += 0.01 * rng.normal(size=20)
            o.set_state(m); o.set_cov(P); g.set_state(b, m, P)
            filters.append(o)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        d_H = dev(Hs.transpose(0, 2, 1))                                        # [filter][col][row] = column-major rows x n
        d_v = dev(vs)
        chi = torch.zeros((B,), dtype=torch.float64, device="cuda"); stt = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_dev(rows, n, d_H.data_ptr(), d_v.data_ptr(), 0.05, 0, chi.data_ptr(), stt.data_ptr())
        torch.cuda.synchronize()
        chi0, st0 = chi.cpu().numpy(), stt.cpu().numpy()
        g.visual_dev(rows, n, d_H.data_ptr(), d_v.data_ptr(), 0.05, 2, chi.data_ptr(), stt.data_ptr())
        torch.cuda.synchronize()
        chi2_, st2 = chi.cpu().numpy(), stt.cpu().numpy()
        seen = set()
        for b, o in enumerate(filters):
            status, c = o.visual_track_outlier_check(Hs[b], np.zeros(rows), vs[b], 0.05)
            assert st0[b] == status == st2[b], b
            assert abs(chi0[b] - c) <= 1e-9 * max(1.0, abs(c)) and abs(chi2_[b] - c) <= 1e-9 * max(1.0, abs(c)), (b, chi0[b], c)
            m0, P0 = o.m.copy(), o.P.copy()
            mg, Pg = g.get_state(b)
            if status == 0:
                o.update_visual_track(Hs[b], np.zeros(rows), vs[b], 0.05)
                assert _rel(mg, o.m) < 1e-9 and _rel(Pg, o.P) < 1e-8, (b, _rel(mg, o.m), _rel(Pg, o.P))
            else:
                assert np.array_equal(mg, m0) and np.array_equal(Pg, P0), b
            seen.add(int(status))
        assert seen == {0, 3}
        g.close()
