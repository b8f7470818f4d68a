"""GPU parity tests of per-track triangulation + prepareVisualUpdate through the C ABI against
oracle/triangulation_oracle.c, which the reference's own tests pin (tests/test_oracle_triangulation.py):
TriangulatorStatus / PrepareVuStatus bit-exact, the triangulated point, H and f within 1e-9 relative (f64 with a
different summation order: parallel derivative columns, FMA contraction)."""
import os

import numpy as np
import pytest

from hybvio_amd import capi, synth

pytestmark = pytest.mark.gpu

# Kernel variants the library selects by shape / batch size, forced here so that EVERY instantiation the benchmark runs is
# parity-tested at test-sized batches too: the two-per-CU / latency builds of the prepare kernel
# (vu_prepare_kernel_2percu is what B = 1024 runs), the fused column-sparse gate (r03 default) against the dense
# gate + update kernels, and the three forms of a speculative pass.
VARIANTS = {
    "default": {},
    "vu384": {"vu_threads": 384},
    "vu768": {"vu_threads": 768},
    "dense": {"ekf_fused_gate": 0},
    "gate_in_prepare": {"ekf_fused_gate": 1, "vu_threads": 384},     # vu_gate_kernel_2percu (what more than 256 filters run by default)
    "gate_own_launch": {"ekf_fused_gate": 2},                        # vu_compact_kernel + ekf_sparse_gate_kernel
    "gate_own_launch_vu384": {"ekf_fused_gate": 2, "vu_threads": 384},
    "dense_vu384": {"ekf_fused_gate": 0, "vu_threads": 384},
    "spec3": {"ekf_spec_mode": 3},
    "split": {"ekf_spec_split": 1},
    "spec2_vu384": {"vu_threads": 384},
    "updates_one_by_one": {"ekf_dual_update": 0},                    # ragged two-class visits without ekf_update_dual_kernel
    "filter_order": {"ekf_visit_order": 0},                          # ... and without the longest-track-first permutation of the fused launches
    "one_stream": {"ekf_side_stream": 0},                            # ... with the long class's prepare + gate launches behind the fused launch
    # (ekf_visit_order 2: the per-frame sort -- and with it the second-stream forms -- also below one filter per CU, where the default skips them)
    "sorted_small_batch": {"ekf_visit_order": 2},                    # the many-filter default (gates on the second stream, enqueued first) on a small batch
    # r04: the long class's prepare + gate is ONE launch (vu_gate_long_kernel); r03's two launches stay behind knob ekf_long_fused = 0
    "long_two_launches": {"ekf_long_fused": 0},
    "long_two_launches_sorted": {"ekf_long_fused": 0, "ekf_visit_order": 2},
    "long_two_launches_one_stream": {"ekf_long_fused": 0, "ekf_side_stream": 0, "ekf_visit_order": 2},
    "sorted_one_stream": {"ekf_side_stream": 0, "ekf_visit_order": 2},
    "fork_r03_arrangement": {"ekf_side_stream": 3, "ekf_visit_order": 2},   # long class on the second stream (r04 default: on the context stream)
    # r05: the long build stores the compact Jacobian behind its gate, for inliers only; 0 = for every prepared track, in front of the gate (r04)
    "jacobian_in_front_of_the_gate": {"ekf_defer_jacobian": 0},
    # r06: the triangulation front as its own launch (vu_tri_kernel: a wavefront per short track, four per long track) + record-fed gates;
    # 2 = at every batch size (default 1: where the two-per-CU build would run, i.e. above one filter per CU); 0 = r05's fused kernels
    "split_tri": {"ekf_split_tri": 2, "ekf_visit_order": 2},
    "split_tri_one_stream": {"ekf_split_tri": 2, "ekf_visit_order": 2, "ekf_side_stream": 0},
    "split_tri_unsorted": {"ekf_split_tri": 2, "ekf_visit_order": 0},
    "fused_front": {"ekf_split_tri": 0},
    # (the split form's short class ends at 12 stereo poses = 48 rows; 11 = the fused builds' boundary)
    "split_tri_short11": {"ekf_split_tri": 2, "ekf_visit_order": 2, "ekf_short_np": 11},
}


def _well_conditioned(diag):
    """oracle.tri_last_diag() -> the track's triangulation status is reproducible to the bit (see the ragged frame-loop test)."""
    min_rcond, min_hz, iters, converged = diag
    return bool(converged == 1 and iters <= 5 and min_rcond > 1e-8 and min_hz > 1e-3)


def _apply(ctx, variant):
    for k, v in VARIANTS[variant].items():
        ctx.set_knob(k, v)
        assert ctx.get_knob(k) == v


FX = os.path.join(os.path.dirname(__file__), "golden", "triangulation_reference_fixtures.npz")
POS, ORI, SFT, CAM = 0, 6, 19, 20
TOL = 1e-9


def _state_from_poses(poses, cam_pose_count):
    m = np.zeros(20 + 7 * cam_pose_count)
    m[POS:POS + 3], m[ORI:ORI + 4] = poses[0:3], poses[3:7]
    for i in range(9):
        m[CAM + 7 * i:CAM + 7 * i + 7] = poses[7 * (i + 1):7 * (i + 2)]
    for i in range(9, cam_pose_count):
        m[CAM + 7 * i + 3] = 1.0
    return m


def _device_prepare(ctx, trail_len, means, vp, idx, feat, vel, y=None):
    """Runs hv_ekf_visual_prepare_dev for a batch; all filters share the track length."""
    import torch
    B = len(means)
    g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
    for b in range(B):
        g.set_state(b, means[b])
    n, npose = g.n, idx.shape[1]
    rows = 2 * npose * (2 if vp.useStereo else 1)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
    d_idx, d_feat, d_vel = dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64)
    d_y = dev(y, np.float64) if y is not None else None
    H = torch.full((B, n, rows), np.nan, dtype=torch.float64, device="cuda")
    v, f = torch.zeros((B, rows), dtype=torch.float64, device="cuda"), torch.zeros((B, rows), dtype=torch.float64, device="cuda")
    pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
    st = torch.full((B, 2), -1, dtype=torch.int32, device="cuda")
    act = torch.full((B,), 7, dtype=torch.uint8, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g.visual_prepare_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), d_y.data_ptr() if d_y is not None else 0,
                         H.data_ptr(), v.data_ptr(), f.data_ptr(), pf.data_ptr(), st.data_ptr(), act.data_ptr())
    torch.cuda.synchronize()
    g.close()
    return (H.cpu().numpy().transpose(0, 2, 1), v.cpu().numpy(), f.cpu().numpy(), pf.cpu().numpy(), st.cpu().numpy(), act.cpu().numpy())


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def test_reference_visual_fixture_matlab_point(oracle):
    """test/triangulation.cpp "visual": the Matlab point (|.|_1 < 1e-5) and the oracle's H, f -- through the device."""
    fx = np.load(FX)
    m = _state_from_poses(fx["visual_poses"], 20)
    T = oracle.vec2matrix(fx["imu_default"])
    uv, vel, idx = fx["visual_uv"], np.full((10, 2), 0.1), np.arange(10, dtype=np.int32)
    vp = capi.vu_default_params(imu_to_camera=T)
    with capi.Context(width=64, height=64) as ctx:
        H, v, f, pf, st, act = _device_prepare(ctx, 20, [m], vp, idx[None], uv[None], vel[None], y=uv.reshape(1, -1))
    assert st[0].tolist() == [0, 0] and act[0] == 1
    assert np.abs(pf[0] - fx["visual_pf_matlab"]).sum() < 1e-5
    ost, ops, opf, oH, of = oracle.visual_track_prepare(oracle.tri_default_params(), m, idx, T, None, uv, vel)
    assert (ost, ops) == (0, 0)
    assert _rel(pf[0], opf) < TOL and _rel(H[0], oH) < TOL and _rel(f[0], of) < TOL
    assert np.abs(v[0] - (uv.reshape(-1) - of)).max() < 1e-12


def test_reference_stereo_fixture(oracle):
    """test/triangulation.cpp "stereo_visual" inputs: 10 poses x 2 cameras, 40 x 90 Jacobian."""
    fx = np.load(FX)
    m = _state_from_poses(fx["stereo_poses"], 10)
    T1, T2 = oracle.vec2matrix(fx["stereo_imu"]), oracle.vec2matrix(fx["stereo_imu2"])
    T2[:3, 3] += fx["stereo_translation"]
    uv = np.concatenate([fx["stereo_uv"], fx["stereo_uv2"] * 1.1])
    vel, idx = np.full((20, 2), 0.1), np.arange(10, dtype=np.int32)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    with capi.Context(width=64, height=64) as ctx:
        H, v, f, pf, st, act = _device_prepare(ctx, 10, [m], vp, idx[None], uv[None], vel[None])
    ost, ops, opf, oH, of = oracle.visual_track_prepare(oracle.tri_default_params(), m, idx, T1, T2, uv, vel)
    assert st[0].tolist() == [ost, ops] == [0, 0]
    assert H.shape == (1, 40, 90)
    assert _rel(pf[0], opf) < TOL and _rel(H[0], oH) < TOL and _rel(f[0], of) < TOL


def _quat(axis, angle):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle / 2)], np.sin(angle / 2) * axis])


def _random_tracks(oracle, rng, B, trail_len, npose, stereo, bad_fraction=0.25, given_means=None):
    """B filters on smooth random trajectories, one track each: a world point projected into the chosen poses (plus
    pixel noise), some tracks replaced by nonsense so that every failure status occurs."""
    n = 20 + 7 * trail_len
    T1 = oracle.vec2matrix([0, -1, 0, -1, 0, 0, 0, 0, -1])
    T2 = T1.copy(); T2[:3, 3] += [0.11, 0.002, -0.001]
    means, idxs, feats, vels = [], [], [], []
    for b in range(B):
        m = np.zeros(n)
        base_q = _quat(rng.normal(size=3), rng.uniform(0, 0.4))
        vel_dir = rng.normal(size=3) * 0.15
        for k in range(trail_len + 1 if given_means is None else 0):
            ip = POS if k == 0 else CAM + 7 * (k - 1)
            io = ORI if k == 0 else CAM + 7 * (k - 1) + 3
            m[ip:ip + 3] = -vel_dir * k + 0.01 * rng.normal(size=3)
            q = base_q + 0.02 * k * rng.normal(size=4) * 0.2
            m[io:io + 4] = q / np.linalg.norm(q)
        if given_means is not None:
            m = given_means[b].copy()
        idx = np.sort(rng.choice(trail_len + 1, npose, replace=False)).astype(np.int32)
        trail = oracle.extract_camera_pose_trail(m, idx, T1, T2 if stereo else None)
        # a point 2..12 m in front of the first camera of the track
        R0, p0 = np.array(trail[0].R).reshape(3, 3), np.array(trail[0].p)
        pw = p0 + R0.T @ np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 12)])
        ft = []
        for t in trail:
            pc = np.array(t.R).reshape(3, 3) @ (pw - np.array(t.p))
            ft.append(pc[:2] / pc[2] + 2e-3 * rng.normal(size=2))
        ft = np.array(ft)
        if rng.uniform() < bad_fraction:
            ft = rng.normal(size=ft.shape) * rng.choice([0.05, 1.0])
        means.append(m); idxs.append(idx); feats.append(ft); vels.append(rng.normal(size=ft.shape) * 0.2)
    return T1, (T2 if stereo else None), np.array(means), np.array(idxs), np.array(feats), np.array(vels)


@pytest.mark.parametrize("trail_len,npose,stereo", [(20, 10, True), (20, 21, True), (20, 4, False), (20, 21, False), (12, 2, True),
                                                    (20, 7, True), (5, 6, False)])
@pytest.mark.parametrize("variant", ["vu768", "vu384"])
def test_random_tracks_match_the_oracle(oracle, trail_len, npose, stereo, variant):
    rng = np.random.default_rng(100 * trail_len + npose + (7 if stereo else 0))
    B = 48
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, stereo)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2) if stereo else capi.vu_default_params(imu_to_camera=T1)
    y = feat.reshape(B, -1) + 1e-3
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        H, v, f, pf, st, act = _device_prepare(ctx, trail_len, means, vp, idx, feat, vel, y=y)
    par = oracle.tri_default_params()
    seen = set()
    for b in range(B):
        ost, ops, opf, oH, of = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2, feat[b], vel[b])
        assert st[b].tolist() == [ost, ops], (b, st[b], ost, ops)
        assert act[b] == (1 if (ost, ops) == (0, 0) else 0)
        seen.add(oracle.TRI_STATUS[ost])
        if ost == 0 and ops == 0:
            assert _rel(pf[b], opf) < TOL and _rel(H[b], oH) < TOL and _rel(f[b], of) < TOL, (b, _rel(H[b], oH))
            assert np.abs(v[b] - (y[b] - of)).max() < 1e-9
            assert not H[b][:, 3:6].any() and not H[b][:, 10:19].any()          # velocity and biases never enter
    assert "OK" in seen and len(seen) >= 2, seen


@pytest.mark.parametrize("trail_len,npose,stereo", [(20, 10, True), (20, 21, True), (20, 5, False), (12, 2, True), (20, 21, False)])
def test_linear_triangulation_branch_matches_the_oracle(oracle, trail_len, npose, stereo):
    """odometry.useLinearTriangulation (parameter_definitions.c:31; Triangulator::triangulate :146-152 -> triangulateLinear :820-895):
    closed-form point and derivatives, statuses OK / BEHIND (/ BAD_DEPTH from the caller's window), the same prepareVisualUpdate."""
    rng = np.random.default_rng(900 + 100 * trail_len + npose + (7 if stereo else 0))
    B = 48
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, stereo)
    kw = dict(useLinearTriangulation=1, triangulationMaxDist=60.0)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2, **kw) if stereo else capi.vu_default_params(imu_to_camera=T1, **kw)
    y = feat.reshape(B, -1) - 2e-3
    with capi.Context(width=64, height=64) as ctx:
        H, v, f, pf, st, act = _device_prepare(ctx, trail_len, means, vp, idx, feat, vel, y=y)
        vp0 = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2) if stereo else capi.vu_default_params(imu_to_camera=T1)
        H0, _, _, pf0, st0, _ = _device_prepare(ctx, trail_len, means, vp0, idx, feat, vel, y=y)
    par = oracle.tri_default_params(**kw)
    seen, n_ok = set(), 0
    for b in range(B):
        ost, ops, opf, oH, of = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2, feat[b], vel[b])
        assert st[b].tolist() == [ost, ops], (b, st[b], ost, ops)
        assert act[b] == (1 if (ost, ops) == (0, 0) else 0)
        seen.add(oracle.TRI_STATUS[ost])
        if ost == 0 and ops == 0:
            n_ok += 1
            assert _rel(pf[b], opf) < TOL and _rel(H[b], oH) < TOL and _rel(f[b], of) < TOL, (b, _rel(H[b], oH))
            assert np.abs(v[b] - (y[b] - of)).max() < 1e-9
            if st0[b].tolist() == [0, 0]:                     # a different estimator: the parameter did switch branches
                assert not np.array_equal(H[b], H0[b])
    assert "OK" in seen and n_ok >= B // 3, (seen, n_ok)
    assert seen <= {"OK", "BEHIND", "BAD_DEPTH"}, seen      # the closed form has no convergence / conditioning statuses


def test_parameters_reach_the_kernel(oracle):
    """Depth window, iteration limit and the time-shift switch change the result exactly as in the oracle."""
    rng = np.random.default_rng(5)
    B = 16
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, 20, 8, True, bad_fraction=0.0)
    for over in ({"triangulationMaxDist": 4.0}, {"triangulationMinDist": 6.0}, {"triangulationGaussNewtonIterations": 1},
                 {"estimateImuCameraTimeShift": 0}, {"triangulationRcondThreshold": 1e-2}):
        vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2, **over)
        par = oracle.tri_default_params(**over)
        with capi.Context(width=64, height=64) as ctx:
            H, v, f, pf, st, act = _device_prepare(ctx, 20, means, vp, idx, feat, vel)
        statuses = set()
        for b in range(B):
            ost, ops, opf, oH, of = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2, feat[b], vel[b])
            assert st[b].tolist() == [ost, ops], (over, b)
            statuses.add(ost)
            if (ost, ops) == (0, 0):
                assert _rel(H[b], oH) < TOL and _rel(f[b], of) < TOL
                if not par.estimateImuCameraTimeShift:
                    assert not H[b][:, SFT].any()
        if "triangulationGaussNewtonIterations" in over:
            assert statuses == {4}                                                     # NO_CONVERGENCE everywhere
        if "triangulationMaxDist" in over or "triangulationMinDist" in over:
            assert 5 in statuses                                                        # BAD_DEPTH


@pytest.mark.parametrize("variant,npose", [("default", 9), ("vu384", 9), ("dense", 9), ("dense_vu384", 9), ("gate_in_prepare", 9), ("gate_own_launch", 9),
                                           ("gate_own_launch_vu384", 9),
                                           # long tracks (49 .. 84 rows): prepare + big column-sparse gate in one launch (vu_gate_long_kernel, r04:
                                           # every tile count 4 / 5 / 6-tight and the 48-row edge) + two block updates; `long_two_launches`: r03's
                                           # vu_compact_kernel + ekf_sparse_gate_big_kernel; `dense`: r02's H-from-L2 / global-workspace kernels
                                           ("default", 13), ("default", 16), ("default", 17), ("default", 20), ("default", 21), ("dense", 21), ("default", 12),
                                           ("long_two_launches", 13), ("long_two_launches", 20), ("long_two_launches", 21),
                                           ("jacobian_in_front_of_the_gate", 13), ("jacobian_in_front_of_the_gate", 21)])
def test_visual_track_dev_prepare_gate_update_equals_the_reference_sequence(oracle, variant, npose):
    """hv_ekf_visual_track_dev = backend.cpp:1063-1185 for one track per filter: prepare from the device mean, gate with
    trackChiTestOutlierR, update with visualR only where triangulation, prepare and gate pass. Filters that fail stay
    bit-identical; the others match the oracle's EKF to 1e-9 relative."""
    import torch
    rng = np.random.default_rng(11 + npose)
    B, trail_len = 24, 20
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.3)
    y = feat.reshape(B, -1) + 2e-3 * rng.normal(size=(B, feat.shape[1] * 2))
    y[3:12] += 3.0                                                                      # triangulable tracks the gate must reject
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05                                                         # parameter_definitions.c:23,91
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            A = rng.normal(size=(o.n, o.n)) * 0.02
            P = o.P.copy() * 1e-6 + A @ A.T * 1e-3 + np.eye(o.n) * 1e-4
            o.set_state(means[b]); o.set_cov(P)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
        d_idx, d_feat, d_vel, d_y = dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(y, np.float64)
        st = torch.full((B, 2), -1, dtype=torch.int32, device="cuda")
        gs = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        chi = torch.zeros((B,), dtype=torch.float64, device="cuda")
        pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_track_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), d_y.data_ptr(), r_gate, r_update,
                           st.data_ptr(), gs.data_ptr(), chi.data_ptr(), pf.data_ptr())
        torch.cuda.synchronize()
        st, gs, chi = st.cpu().numpy(), gs.cpu().numpy(), chi.cpu().numpy()
        outcomes = set()
        for b, o in enumerate(filters):
            ost, ops, opf, oH, of = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2, feat[b], vel[b])
            assert st[b].tolist() == [ost, ops]
            m0, P0 = o.m.copy(), o.P.copy()
            mg, Pg = g.get_state(b)
            if (ost, ops) != (0, 0):
                assert gs[b] == 1                                                         # NOT_COMPUTED
                assert np.array_equal(mg, m0) and np.array_equal(Pg, P0)
                outcomes.add("skipped")
                continue
            status, chi2 = o.visual_track_outlier_check(oH, of, y[b], r_gate)
            assert gs[b] == status and abs(chi[b] - chi2) <= 1e-7 * max(1.0, abs(chi2))
            if status == 0:
                o.update_visual_track(oH, of, y[b], r_update)
                assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-8, (b, _rel(mg, o.m), _rel(Pg, o.P))
                assert not np.array_equal(mg, m0)
                outcomes.add("updated")
            else:
                assert np.array_equal(mg, m0) and np.array_equal(Pg, P0)
                outcomes.add("gated")
        assert outcomes == {"skipped", "updated", "gated"}, outcomes
        g.close()


@pytest.mark.parametrize("variant,npose", [("default", 13), ("default", 17), ("default", 21), ("long_two_launches", 21), ("default", 9)])
def test_long_gate_chi2_with_time_shift_cross_covariance(oracle, variant, npose):
    """r04 advisor: structured_S (the factor form of the long class's gate, ekf_device.hpp) left W(:, sft) * dpf/dt out of WF = W F4'.
    The term only shows when P couples the IMU-camera time shift (state 19) with the poses -- here strongly -- and the track has a
    feature velocity, i.e. on every real sequence once the filter has run. chi2 and S must agree with the oracle as tightly as the
    MFMA gates do: 1e-10 relative (the missing term was 1e-7 .. 2e-6)."""
    import torch
    rng = np.random.default_rng(500 + npose)
    B, trail_len = 16, 20
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.0)
    vel *= 5.0                                                                           # px/s-scale feature velocities: a large d pf / d t
    y = feat.reshape(B, -1) + 2e-3 * rng.normal(size=(B, feat.shape[1] * 2))
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    assert vp.estimateImuCameraTimeShift
    par = oracle.tri_default_params()
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            A = rng.normal(size=(o.n, o.n)) * 0.02
            A[SFT, :] *= 30.0                                                            # the time shift is the most uncertain, most coupled state
            P = A @ A.T * 1e-3 + np.eye(o.n) * 1e-5
            o.set_state(means[b]); o.set_cov(P)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
        d_idx, d_feat, d_vel, d_y = dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(y, np.float64)
        st = torch.full((B, 2), -1, dtype=torch.int32, device="cuda")
        gs = torch.full((B,), -1, dtype=torch.int32, device="cuda")
        chi = torch.zeros((B,), dtype=torch.float64, device="cuda")
        pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_track_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), d_y.data_ptr(), 1.5, 0.05,
                           st.data_ptr(), gs.data_ptr(), chi.data_ptr(), pf.data_ptr())
        torch.cuda.synchronize()
        st, gs, chi = st.cpu().numpy(), gs.cpu().numpy(), chi.cpu().numpy()
        compared, coupled = 0, 0
        for b, o in enumerate(filters):
            ost, ops, opf, oH, of = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2, feat[b], vel[b])
            assert st[b].tolist() == [ost, ops]
            if (ost, ops) != (0, 0):
                continue
            status, chi2 = o.visual_track_outlier_check(oH, of, y[b], 1.5)
            assert gs[b] == status
            assert abs(chi[b] - chi2) <= 1e-10 * max(1.0, abs(chi2)), (b, chi[b], chi2)
            # the input exercises the term: dropping the time-shift column's coupling moves chi2 by far more than the tolerance
            P2 = o.P.copy(); P2[SFT, :] = 0.0; P2[:, SFT] = 0.0
            o2 = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len)); o2.set_state(means[b]); o2.set_cov(P2)
            coupled += abs(o2.visual_track_outlier_check(oH, of, y[b], 1.5)[1] - chi2) > 1e-8 * abs(chi2)   # 100x the tolerance above
            compared += 1
        assert compared >= B // 2 and coupled >= compared // 2, (compared, coupled)
        g.close()


@pytest.mark.parametrize("variant,npose,B", [("default", 10, 64), ("vu384", 11, 300), ("gate_own_launch", 9, 64), ("default", 21, 64),
                                             ("long_two_launches", 17, 64)])
def test_gate_is_reproducible_to_the_bit(oracle, variant, npose, B):
    """r05: the partial sums of S = Hc P(a, a) Hc' meet in LDS in a fixed order (ekf_device.hpp sparse_gate: a turn counter per column
    tile; the long build's factor form never used atomics), so chi2 -- and with it the gate status -- is the same bit pattern every time
    the same track is gated against the same state. r03 / r04 added the waves' contributions as they arrived (ds_add_f64): two runs
    could differ in the last place. Six repetitions with other work in flight in between."""
    import torch
    rng = np.random.default_rng(900 + npose)
    trail_len = 20
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.0)
    y = feat.reshape(B, -1) + 2e-3 * rng.normal(size=(B, feat.shape[1] * 2))
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    n = 20 + 7 * trail_len
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        Ps = []
        for b in range(B):
            A = rng.normal(size=(n, n)) * 0.02
            Ps.append(A @ A.T * 1e-3 + np.eye(n) * 1e-4)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
        d_idx, d_feat, d_vel, d_y = dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(y, np.float64)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        runs = []
        busy = torch.zeros(1 << 22, device="cuda")
        for rep in range(6):
            for b in range(B):
                g.set_state(b, means[b], Ps[b])
            st = torch.full((B, 2), -1, dtype=torch.int32, device="cuda")
            gs = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            chi = torch.zeros((B,), dtype=torch.float64, device="cuda")
            pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
            if rep % 2:
                busy.add_(1.0)                                   # other work in front: the waves of the gate start on a busier chip
            # r_gate large: nothing is applied by this call's update half, the state stays what the next repetition needs anyway
            g.visual_track_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), d_y.data_ptr(), 1.5, 0.05,
                               st.data_ptr(), gs.data_ptr(), chi.data_ptr(), pf.data_ptr())
            torch.cuda.synchronize()
            runs.append((chi.cpu().numpy().view(np.uint64).copy(), gs.cpu().numpy().copy()))
        assert (runs[0][1] == 0).sum() + (runs[0][1] == 3).sum() >= B // 2           # gated tracks
        for chi_bits, gs_ in runs[1:]:
            assert np.array_equal(chi_bits, runs[0][0]) and np.array_equal(gs_, runs[0][1])
        g.close()


@pytest.mark.parametrize("variant", ["default", "vu384", "dense", "gate_own_launch_vu384"])
def test_frame_loop_with_the_successful_update_quota(oracle, variant):
    """A frame's visual-update loop for a batch (backend.cpp:1012-1240): K tracks per filter, visited in order, each seeing the
    mean the previous one left; a filter stops being visited once maxSuccessfulVisualUpdates updates were applied."""
    import torch
    rng = np.random.default_rng(23)
    B, trail_len, npose, K, quota = 12, 20, 6, 9, 3
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.0)
    tracks = [_random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.3, given_means=means)[3:] for _ in range(K)]
    ys = [t[1].reshape(B, -1) + 2e-3 * rng.normal(size=(B, t[1].shape[1] * 2)) for t in tracks]
    for k in (1, 4):
        ys[k][::3] += 3.0                                                                 # some gate rejections
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            P = o.P.copy() * 1e-6 + np.eye(o.n) * 1e-4
            o.set_state(means[b]); o.set_cov(P)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
        counter = torch.zeros((B,), dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        got = []
        for k in range(K):
            idx, feat, vel = tracks[k]
            d = [dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(ys[k], np.float64)]
            st = torch.full((B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((B,), -9, dtype=torch.int32, device="cuda")
            g.visual_track_limited_dev(vp, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), r_gate, r_update,
                                       st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota)
            torch.cuda.synchronize()
            got.append((st.cpu().numpy(), gs.cpu().numpy()))
        counts = counter.cpu().numpy()
        reached = 0
        for b, o in enumerate(filters):
            done = 0
            for k in range(K):
                idx, feat, vel = tracks[k]
                st, gs = got[k]
                if done >= quota:
                    assert st[b].tolist() == [-1, -1] and gs[b] == 1                       # not visited any more
                    continue
                ost, ops, opf, oH, of = oracle.visual_track_prepare(par, o.m.copy(), idx[b], T1, T2, feat[b], vel[b])
                assert st[b].tolist() == [ost, ops], (b, k)
                if (ost, ops) != (0, 0):
                    assert gs[b] == 1
                    continue
                status, _ = o.visual_track_outlier_check(oH, of, ys[k][b], r_gate)
                assert gs[b] == status, (b, k)
                if status == 0:
                    o.update_visual_track(oH, of, ys[k][b], r_update)
                    done += 1
            assert counts[b] == done
            reached += done >= quota
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
        assert 0 < reached <= B
        g.close()


@pytest.mark.parametrize("B,speculative,variant,npose", [
    (12, True, "default", 6), (12, True, "spec2_vu384", 6), (12, True, "spec3", 6), (12, True, "split", 6),
    (40, False, "default", 6), (40, False, "vu384", 6), (40, False, "dense", 6), (40, False, "dense_vu384", 6),
    (40, False, "gate_own_launch", 6), (40, False, "gate_own_launch_vu384", 6), (40, False, "gate_in_prepare", 6),
    # r04: long tracks (49 .. 84 rows) in the speculative loop -- vu_gate_long_kernel over (filters, tracks), two block-update launches
    (12, True, "default", 16), (5, True, "default", 21), (12, True, "default", 13), (12, False, "long_two_launches", 16), (40, False, "default", 16)])
def test_whole_frame_loop_in_one_call(oracle, B, speculative, variant, npose):
    """hv_ekf_visual_frame_dev = the frame's visit loop in ONE call. For few sequences (B * K <= 256) it runs speculatively: every
    pending track prepared and gated in parallel, the first inlier applied, the rest re-examined (<= quota + 1 passes); for more it
    is the sequential per-visit loop. Both must give the reference's sequential result: statuses per visit, the quota, the filter."""
    import torch
    rng = np.random.default_rng(31 + B + npose)
    trail_len, K, quota = 20, 9, 3
    assert (B * K <= 256 and (4 * npose <= 48 or VARIANTS[variant].get("ekf_long_fused", 1) != 0)) == speculative
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.0)
    tracks = [_random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.3, given_means=means)[3:] for _ in range(K)]
    ys = [t[1].reshape(B, -1) + 2e-3 * rng.normal(size=(B, t[1].shape[1] * 2)) for t in tracks]
    for k in (0, 1, 4, 6):
        ys[k][::2] += 3.0                                                                 # gate rejections, also on the first visits
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            P = o.P.copy() * 1e-6 + np.eye(o.n) * 1e-4
            o.set_state(means[b]); o.set_cov(P)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        d = [dev(np.stack([t[0] for t in tracks]), np.int32), dev(np.stack([t[1] for t in tracks]), np.float64),
             dev(np.stack([t[2] for t in tracks]), np.float64), dev(np.stack(ys), np.float64)]
        st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
        counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")                  # zeroed by the call
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_frame_dev(vp, K, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), r_gate, r_update,
                           st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota)
        torch.cuda.synchronize()
        st, gs, counts = st.cpu().numpy(), gs.cpu().numpy(), counter.cpu().numpy()
        reached, rejected = 0, 0
        for b, o in enumerate(filters):
            done = 0
            for k in range(K):
                idx, feat, vel = tracks[k]
                if done >= quota:
                    assert st[k, b].tolist() == [-1, -1] and gs[k, b] == 1, (b, k)         # not visited any more
                    continue
                ost, ops, opf, oH, of = oracle.visual_track_prepare(par, o.m.copy(), idx[b], T1, T2, feat[b], vel[b])
                assert st[k, b].tolist() == [ost, ops], (b, k)
                if (ost, ops) != (0, 0):
                    assert gs[k, b] == 1
                    continue
                status, _ = o.visual_track_outlier_check(oH, of, ys[k][b], r_gate)
                assert gs[k, b] == status, (b, k)
                if status == 0:
                    o.update_visual_track(oH, of, ys[k][b], r_update); done += 1
                else:
                    rejected += 1
            assert counts[b] == done
            reached += done >= quota
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
        assert 0 < reached <= B and rejected > 0
        g.close()


@pytest.mark.parametrize("variant,npose", [("default", 6), ("spec3", 6), ("default", 16), ("default", 21)])
def test_speculative_frame_loop_under_contention(oracle, variant, npose):
    """the speculative visit loop while another stream keeps most of the chip busy. The default pass (a fused
    prepare + gate launch, then an apply launch) has no inter-workgroup hand-shake, so contention can only delay it; the r02 one-launch
    form (knob ekf_spec_mode 3) waits for the gate decisions of workgroups in front of it and must either produce the same result or
    say so (hv_ekf_frame_error). References: the device's sequential visit loop (knob ekf_no_speculation) on the same inputs, and -- r06,
    VERDICT r05 weak 1(iii): device against device proves consistency, not correctness -- the ORACLE's sequential loop
    (backend.cpp:1012-1252 through oracle/triangulation_oracle.c + ekf_oracle.c) for statuses, update counts and the final state."""
    import torch
    rng = np.random.default_rng(2024)
    B, trail_len, K, quota = 12, 20, 9, 3                     # (npose 16 / 21: the long-track form of the loop, r04)
    T1, T2, means, _, _ = synth.visual_tracks(rng, B, trail_len, npose, True)
    tracks, ys = [], []
    for k in range(K):
        _, _, _, i_, f_ = synth.visual_tracks(np.random.default_rng(500 + k), B, trail_len, npose, True, given_means=means)
        tracks.append((i_, f_, np.zeros_like(f_)))
        y = f_.reshape(B, -1) + 2e-3 * rng.normal(size=(B, f_.shape[1] * 2))
        if k in (0, 1, 4, 6):
            y[::2] += 3.0
        ys.append(y)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    r_gate, r_update = 1.5, 0.05
    P0 = np.eye(20 + 7 * trail_len) * 1e-4

    def run(knobs, busy):
        with capi.Context(width=64, height=64) as ctx:
            for k_, v_ in knobs.items():
                ctx.set_knob(k_, v_)
            g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
            for b in range(B):
                g.set_state(b, means[b], P0)
            dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
            d = [dev(np.stack([t[0] for t in tracks]), np.int32), dev(np.stack([t[1] for t in tracks]), np.float64),
                 dev(np.stack([t[2] for t in tracks]), np.float64), dev(np.stack(ys), np.float64)]
            st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
            counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")
            main = torch.cuda.current_stream()
            ctx.set_stream(main.cuda_stream)
            if busy:                                          # ~100 ms of large matrix products on a second stream, started first
                other = torch.cuda.Stream()
                a_ = torch.randn(8192, 8192, device="cuda"); b_ = torch.randn(8192, 8192, device="cuda")
                torch.cuda.synchronize()
                with torch.cuda.stream(other):
                    for _ in range(12):
                        a_ = (a_ @ b_) * 1e-4
            for _ in range(3 if busy else 1):                 # several frames while the other stream is busy
                for b in range(B):
                    g.set_state(b, means[b], P0)
                g.visual_frame_dev(vp, K, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), r_gate, r_update,
                                   st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota)
            err = g.frame_error()
            torch.cuda.synchronize()
            out = (st.cpu().numpy().copy(), gs.cpu().numpy().copy(), counter.cpu().numpy().copy(), [g.get_state(b) for b in range(B)], err)
            g.close()
            return out

    ref = run({"ekf_no_speculation": 1}, False)
    got = run(dict(VARIANTS[variant]), True)
    if got[4] != 0:                                           # loud failure of the hand-shake form: allowed, a silent difference is not
        assert variant == "spec3"
        return
    assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all() and (got[2] == ref[2]).all()
    for (mg, Pg), (mr, Pr) in zip(got[3], ref[3]):
        assert _rel(mg, mr) < 1e-9 and _rel(Pg, Pr) < 1e-8
    assert ref[2].sum() > 0
    # ... and the oracle's loop on the same frame: every visit's statuses, the applied-update counts, the state the frame leaves
    par = oracle.tri_default_params()
    applied = 0
    for b in range(B):
        o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
        o.set_state(means[b]); o.set_cov(P0)
        done = 0
        for k in range(K):
            if done >= quota:
                assert got[0][k, b].tolist() == [-1, -1] and got[1][k, b] == 1, (b, k)
                continue
            i_, f_, v_ = tracks[k][0][b], tracks[k][1][b], tracks[k][2][b]
            ost, ops, opf, oH, of = oracle.visual_track_prepare(par, o.m.copy(), i_, T1, T2, f_, v_)
            assert got[0][k, b].tolist() == [ost, ops], (b, k, got[0][k, b].tolist(), [ost, ops])
            if (ost, ops) != (0, 0):
                assert got[1][k, b] == 1
                continue
            status, _ = o.visual_track_outlier_check(oH, of, ys[k][b], r_gate)
            assert got[1][k, b] == status, (b, k)
            if status == 0:
                o.update_visual_track(oH, of, ys[k][b], r_update); done += 1
        assert got[2][b] == done, (b, got[2][b], done)
        applied += done
        mg, Pg = got[3][b]
        assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
    assert applied > 0


@pytest.mark.parametrize("B,speculative,stereo,variant,np_max", [
    (10, True, True, "default", 10), (48, False, True, "default", 10), (9, True, False, "default", 10), (10, True, True, "spec3", 10),
    (48, False, True, "vu384", 10), (48, False, True, "dense", 10), (9, True, False, "spec2_vu384", 10), (48, False, False, "vu384", 10),
    (48, False, True, "gate_own_launch_vu384", 10), (48, False, False, "gate_own_launch", 10),
    # tracks of up to 21 poses: stereo batches split into a short class (fused two-per-CU kernels, <= 11 poses) and a long
    # class (dense kernels) per visit; mono tracks of 21 poses still fit the fused kernels (42 rows)
    (48, False, True, "default", 21), (48, False, True, "vu384", 21), (10, True, True, "default", 21), (48, False, False, "default", 21),
    # (r04: few sequences with long tracks run the speculative loop too -- one build of the fused launch for every length)
    (1, True, True, "default", 21), (30, True, True, "default", 21), (10, False, True, "long_two_launches", 21), (10, True, True, "default", 16),
    (48, False, True, "dense", 21), (48, False, True, "updates_one_by_one", 21), (48, False, True, "filter_order", 21), (48, False, True, "one_stream", 21),
    (48, False, True, "long_two_launches", 21), (48, False, True, "long_two_launches_sorted", 21), (48, False, True, "long_two_launches_one_stream", 21),
    (48, False, True, "sorted_small_batch", 21), (48, False, True, "sorted_one_stream", 21), (48, False, True, "fork_r03_arrangement", 21),
    (300, False, True, "fork_r03_arrangement", 21),
    (300, False, True, "default", 21), (300, False, True, "one_stream", 21), (300, False, True, "long_two_launches", 21),
    # r06: the split form (default above one filter per CU) at small batches, on one stream, in filter order; r05's fused kernels behind the knob
    (48, False, True, "split_tri", 21), (48, False, True, "split_tri_one_stream", 21), (48, False, True, "split_tri_unsorted", 21),
    (48, False, True, "split_tri", 10), (300, False, True, "fused_front", 21), (300, False, True, "split_tri_one_stream", 21),
    # ... the class boundary: 12 poses (48 rows) are the record-fed short gate's longest track, 13 the long class's shortest
    (48, False, True, "split_tri", 12), (48, False, True, "split_tri", 13), (300, False, True, "split_tri_short11", 21), (300, False, True, "default", 12),
    # r06: the headline's grid -- 1024 distinct ragged filters in one frame loop, every one against the oracle
    (1024, False, True, "default", 21)])
def test_whole_frame_loop_with_ragged_track_lengths(oracle, B, speculative, stereo, variant, np_max):
    """hv_ekf_visual_frame_ragged_dev: the sequences of a batch do not share track lengths -- every (visit, filter) record has its own
    pose count (2 .. n_poses_max, 0 = this filter has no candidate at this visit), padded to the longest. Result = the reference's
    sequential loop run per filter over its own tracks."""
    import torch
    rng = np.random.default_rng(77 + B)
    trail_len, K, quota = 20, 8, 3
    assert (B * K <= 256 and (2 * np_max * (2 if stereo else 1) <= 48 or VARIANTS[variant].get("ekf_long_fused", 1) != 0)) == speculative
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, trail_len, 6, stereo, bad_fraction=0.0)
    ncam = 2 if stereo else 1
    lens = rng.integers(2, np_max + 1, (K, B)).astype(np.int32)
    lens[rng.uniform(size=(K, B)) < 0.15] = 0                                             # no track for this filter at this visit
    lens[0, 0], lens[1, 1 % B] = np_max, 2
    idx = np.zeros((K, B, np_max), np.int32); feat = np.zeros((K, B, ncam * np_max, 2)); vel = np.zeros_like(feat)
    ys = np.zeros((K, B, 2 * ncam * np_max))
    per = {}
    for k in range(K):
        for n in sorted(set(lens[k].tolist()) - {0}):
            sel = np.nonzero(lens[k] == n)[0]
            _, _, _, i_, f_, v_ = _random_tracks(oracle, rng, len(sel), trail_len, n, stereo, bad_fraction=0.25, given_means=means[sel])
            for j, b in enumerate(sel):
                yy = f_[j].reshape(-1) + 2e-3 * rng.normal(size=f_[j].size) + (3.0 if (k + b) % 3 == 0 else 0.0)
                idx[k, b, :n] = i_[j]; feat[k, b, :ncam * n] = f_[j]; vel[k, b, :ncam * n] = v_[j]; ys[k, b, :2 * ncam * n] = yy
                per[(k, b)] = (i_[j], f_[j], v_[j], yy)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2) if stereo else capi.vu_default_params(imu_to_camera=T1)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            P = o.P.copy() * 1e-6 + np.eye(o.n) * 1e-4
            o.set_state(means[b]); o.set_cov(P)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        d = [dev(lens, np.int32), dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(ys, np.float64)]
        st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
        counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_frame_ragged_dev(vp, K, np_max, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                  r_gate, r_update, st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota)
        torch.cuda.synchronize()
        st, gs, counts = st.cpu().numpy(), gs.cpu().numpy(), counter.cpu().numpy()
        applied, rejected, lengths_applied, ties = 0, 0, set(), 0
        for b, o in enumerate(filters):
            done = 0
            for k in range(K):
                if done >= quota or lens[k, b] == 0:
                    assert st[k, b].tolist() == [-1, -1] and gs[k, b] == 1, (b, k)         # not visited / no track
                    continue
                i_, f_, v_, yy = per[(k, b)]
                ost, ops, opf, oH, of = oracle.visual_track_prepare(par, o.m.copy(), i_, T1, T2 if stereo else None, f_, v_)
                if st[k, b].tolist() != [ost, ops]:
                    # The failure status of a track is only a property of the input while the Gauss-Newton iteration is well conditioned.
                    # Deterministic rule (r04, replaces r03's "some 1e-13 perturbation of the oracle reaches the device's status"): the
                    # oracle reports how close ITS iteration came to a singular step (oracle.tri_last_diag: smallest rcond of E'E,
                    # smallest |h_z| / |h| of a projection, iterations, convergence). A track that converged within 5 iterations with
                    # rcond > 1e-8 and no projection within 1e-3 of a camera plane is WELL conditioned: the device must report exactly the
                    # oracle's status. Any other track (a quarter of this test's tracks are nonsense on purpose) is DEGENERATE: its
                    # iteration amplifies the last bit of every sum, BEHIND / BAD_COND / NO_CONVERGENCE are one class "triangulation
                    # failed", and the assertion is on the class -- nothing downstream depends on which member (gate NOT_COMPUTED, filter
                    # untouched: asserted below).
                    assert not _well_conditioned(oracle.tri_last_diag()), (b, k, lens[k, b], st[k, b].tolist(), [ost, ops], oracle.tri_last_diag())
                    assert ost != 0 and st[k, b, 0] != 0, (b, k, lens[k, b], st[k, b].tolist(), [ost, ops])
                    ties += 1
                if (ost, ops) != (0, 0):
                    assert gs[k, b] == 1
                    continue
                status, _ = o.visual_track_outlier_check(oH, of, yy, r_gate)
                assert gs[k, b] == status, (b, k, lens[k, b])
                if status == 0:
                    o.update_visual_track(oH, of, yy, r_update); done += 1; lengths_applied.add(int(lens[k, b]))
                else:
                    rejected += 1
            assert counts[b] == done
            applied += done
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
        if B >= 10:                                            # (coverage of the case itself; a single sequence cannot promise it)
            assert applied > B // 2 and rejected > 0 and len(lengths_applied) >= 3, (applied, rejected, lengths_applied)
        assert ties <= 1 + B * K // 1000, ties
        if np_max > 12 and B >= 10:
            assert max(lengths_applied) > 12 and min(lengths_applied) < 12, lengths_applied
        g.close()


def _batch_loop_case(oracle, seed, B, np_max, K, quota, max_rows, stereo):
    """Inputs of one hv_ekf_visual_frame_batch_dev call over B distinct filters, and `verify(g, st, gs, counts)`: the oracle's
    batchVisualUpdate loop per filter (backend.cpp:1001-1010,1169-1183,1255-1262) against what the device left."""
    rng = np.random.default_rng(seed)
    trail_len = 20
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, trail_len, 6, stereo, bad_fraction=0.0)
    ncam = 2 if stereo else 1
    lens = rng.integers(2, np_max + 1, (K, B)).astype(np.int32)
    lens[rng.uniform(size=(K, B)) < 0.1] = 0
    lens[0, 0] = np_max
    idx = np.zeros((K, B, np_max), np.int32); feat = np.zeros((K, B, ncam * np_max, 2)); vel = np.zeros_like(feat)
    ys = np.zeros((K, B, 2 * ncam * np_max))
    per = {}
    for k in range(K):
        for n in sorted(set(lens[k].tolist()) - {0}):
            sel = np.nonzero(lens[k] == n)[0]
            _, _, _, i_, f_, v_ = _random_tracks(oracle, rng, len(sel), trail_len, n, stereo, bad_fraction=0.15, given_means=means[sel])
            for j, b in enumerate(sel):
                yy = f_[j].reshape(-1) + 2e-3 * rng.normal(size=f_[j].size) + (3.0 if (k + b) % 4 == 0 else 0.0)
                idx[k, b, :n] = i_[j]; feat[k, b, :ncam * n] = f_[j]; vel[k, b, :ncam * n] = v_[j]; ys[k, b, :2 * ncam * n] = yy
                per[(k, b)] = (i_[j], f_[j], v_[j], yy)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2) if stereo else capi.vu_default_params(imu_to_camera=T1)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    o0 = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
    n_state = o0.n
    P0 = np.zeros((B, n_state, n_state))
    for b in range(B):
        A = rng.normal(size=(n_state, n_state)) * 0.005
        P0[b] = o0.P * 1e-6 + np.eye(n_state) * 1e-4 + (A @ A.T * 1e-3 if B > 8 else 0.0)     # (distinct per filter in the large cases)

    def verify(g, st, gs, counts):
        cap = max_rows if max_rows > 0 else g.n
        flushes, applied, rejected, ties = 0, 0, 0, 0
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            o.set_state(means[b]); o.set_cov(P0[b])
            Hb, fb, yb, rows, done = [], [], [], 0, 0

            def flush():
                o.update_visual_track(np.vstack(Hb), np.concatenate(fb), np.concatenate(yb), r_update)
                Hb.clear(); fb.clear(); yb.clear()

            for k in range(K):
                if done >= quota or lens[k, b] == 0:
                    assert st[k, b].tolist() == [-1, -1] and gs[k, b] == 1, (b, k)
                    continue
                i_, f_, v_, yy = per[(k, b)]
                ost, ops, opf, oH, of = oracle.visual_track_prepare(par, o.m.copy(), i_, T1, T2 if stereo else None, f_, v_)
                if st[k, b].tolist() != [ost, ops]:
                    assert not _well_conditioned(oracle.tri_last_diag()) and ost != 0 and st[k, b, 0] != 0, (b, k, st[k, b].tolist(), [ost, ops])
                    ties += 1
                if (ost, ops) != (0, 0):
                    assert gs[k, b] == 1
                    continue
                status, _ = o.visual_track_outlier_check(oH, of, yy, r_gate)
                assert gs[k, b] == status, (b, k, lens[k, b])
                if status != 0:
                    rejected += 1
                    continue
                nr = oH.shape[0]
                if rows + nr > cap:
                    flush(); rows = 0; flushes += 1
                Hb.append(oH); fb.append(of); yb.append(yy); rows += nr; done += 1
            if rows:
                flush()
            assert counts[b] == done, (b, counts[b], done)
            applied += done
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
        assert applied >= B and rejected > 0 and ties <= 1 + B * K // 1000, (applied, rejected, ties)
        if 0 < max_rows <= 64:
            assert flushes > 0, "no batch overflowed: the carried-block path was not exercised"

    return dict(B=B, K=K, np_max=np_max, quota=quota, max_rows=max_rows, trail_len=trail_len, vp=vp, lens=lens, idx=idx, feat=feat, vel=vel,
                ys=ys, means=means, P0=P0, r_gate=r_gate, r_update=r_update, verify=verify)


def _run_batch_loop(case, ctx, g):
    """Uploads a case's state and inputs and enqueues the call on the CURRENT torch stream; returns the device outputs."""
    import torch
    for b in range(case["B"]):
        g.set_state(b, case["means"][b], case["P0"][b])
    dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
    d = [dev(case["lens"], np.int32), dev(case["idx"], np.int32), dev(case["feat"], np.float64), dev(case["vel"], np.float64), dev(case["ys"], np.float64)]
    K, B = case["K"], case["B"]
    st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
    counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")
    return d, st, gs, counter


@pytest.mark.parametrize("B,np_max,K,quota,max_rows,stereo,variant", [
    (6, 8, 10, 6, 64, True, "default"),          # short tracks (<= 32 rows), batches of 64 rows: several flushes per frame, carried blocks
    (5, 16, 9, 5, 0, True, "default"),           # long tracks (<= 64 rows, the long build), the reference's default batch of stateDim rows
    (7, 10, 8, 8, 40, False, "default"),         # mono, quota = every inlier, small batches
    (3, 21, 8, 4, 160, True, "default"),         # 84-row tracks
    # r05: more filters than the chip has CUs, distinct covariances, forced kernel variants
    (300, 10, 6, 4, 64, True, "default"),        # short build, 1800 records, carried blocks
    (300, 10, 6, 4, 64, True, "vu384"),          # ... on the two-per-CU build
    (300, 16, 6, 3, 0, True, "default"),         # long build over 1800 records
    (130, 21, 6, 3, 160, True, "jacobian_in_front_of_the_gate"),   # 84-row tracks, r04's placement of the compact Jacobian
    (64, 12, 8, 5, 96, True, "vu768")])          # the 48-row edge of the short class on the latency build
def test_frame_loop_with_batch_visual_update(oracle, B, np_max, K, quota, max_rows, stereo, variant):
    """hv_ekf_visual_frame_batch_dev = Session::trackerVisualUpdate with batchVisualUpdate (backend.cpp:1001-1010,1169-1183,1255-1262):
    the inliers' blocks are stacked and applied as one update per batch; every track between two flushes is prepared and gated against the
    state the last flush left; a block that does not fit flushes the batch and opens the next one as it was prepared."""
    import torch
    case = _batch_loop_case(oracle, 500 + B + np_max, B, np_max, K, quota, max_rows, stereo)
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=case["trail_len"]), B)
        d, st, gs, counter = _run_batch_loop(case, ctx, g)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_frame_batch_dev(case["vp"], K, np_max, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                 case["r_gate"], case["r_update"], st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota, max_rows)
        torch.cuda.synchronize()
        case["verify"](g, st.cpu().numpy(), gs.cpu().numpy(), counter.cpu().numpy())
        g.close()


def test_batch_visual_update_refuses_what_it_cannot_serve(oracle):
    """hv_ekf_visual_frame_batch_dev: a batch smaller than the longest track's block, and a threshold growth factor != 1 (the batch
    loop has no per-track rejection to grow on), are errors -- not silent truncation (r05 advisor: the only negative test of the entry
    went away with r05's refactor; EKF::visualFrameBatch forwards maxUpdateRows straight into this check)."""
    import torch
    B, np_max, K, quota = 6, 6, 3, 2
    case = _batch_loop_case(oracle, 4242, B, np_max, K, quota, 96, True)
    with capi.Context(width=64, height=64) as ctx:
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=case["trail_len"]), B)
        d, st, gs, counter = _run_batch_loop(case, ctx, g)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        call = lambda vp, max_rows: g.visual_frame_batch_dev(vp, K, np_max, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                                             d[4].data_ptr(), case["r_gate"], case["r_update"], st.data_ptr(), gs.data_ptr(),
                                                             counter.data_ptr(), quota, max_rows)
        with pytest.raises(capi.HvError):
            call(case["vp"], 2 * 2 * np_max - 2)                  # the longest track's block (24 rows) does not fit the batch
        vp_growth = capi.vu_default_params(imu_to_camera=list(case["vp"].imuToCamera), second_imu_to_camera=list(case["vp"].secondImuToCamera))
        vp_growth.trackOutlierThresholdGrowthFactor = 1.5
        with pytest.raises(capi.HvError):
            call(vp_growth, 96)
        call(case["vp"], 96)                                      # ... and the context still serves a valid call afterwards
        torch.cuda.synchronize()
        case["verify"](g, st.cpu().numpy(), gs.cpu().numpy(), counter.cpu().numpy())
        g.close()


@pytest.mark.parametrize("stereo,npose", [(True, 6), (False, 9), (True, 14)])
def test_hybrid_map_track_visit(oracle, stereo, npose):
    """hv_ekf_visual_track_hybrid_dev (backend.cpp:1016,1075-1082,1146,1160-1168) with odometry.hybridMapSize 3 (state 169 wide): pose-trail
    tracks applied or -- when offered a slot -- inserted as map points, mapPointUpdate tracks prepared from the point in the state
    (status HYBRID, dip R in the point's columns), gate rejections of both kinds; every filter against the oracle's sequence."""
    import torch
    rng = np.random.default_rng(900 + npose)
    B, trail_len, M = 12, 20, 3
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, stereo, bad_fraction=0.0)
    n = 20 + 7 * trail_len + 3 * M
    base = n - 3 * M
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2) if stereo else capi.vu_default_params(imu_to_camera=T1)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    kind = np.arange(B) % 4                  # 0 pose-trail track applied, 1 offered a slot, 2 map-point track, 3 map-point track with a gross error
    map_update = np.where(kind >= 2, np.arange(B) % M, -1).astype(np.int32)
    map_offer = np.where(kind == 1, (np.arange(B) + 1) % M, -1).astype(np.int32)
    map_offer[5] = 2                                   # ... and one offered slot whose track the gate rejects (see ys below)
    ys = feat.reshape(B, -1) + 2e-3 * rng.normal(size=(B, feat.shape[1] * 2))
    ys[kind == 3] += 3.0
    ys[5] += 3.0
    with capi.Context(width=64, height=64) as ctx:
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len, hybridMapSize=M), B)
        assert g.n == n
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len, hybridMapSize=M))
            m = np.concatenate([means[b], rng.normal(size=3 * M)])
            if map_update[b] >= 0:                     # the map point of the track = the point its features were projected from (+ 1 mm)
                _, _, pf_true, _, _ = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2 if stereo else None, feat[b], vel[b])
                m[base + 3 * map_update[b]: base + 3 * map_update[b] + 3] = pf_true + 1e-3 * rng.normal(size=3)
            A = rng.normal(size=(n, n)) * 1e-3
            P = np.eye(n) * 1e-4 + A @ A.T
            o.set_state(m); o.set_cov(P)
            g.set_state(b, m, P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        d = [dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(ys, np.float64), dev(map_update, np.int32), dev(map_offer, np.int32)]
        st = torch.full((B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((B,), -9, dtype=torch.int32, device="cuda")
        chi = torch.zeros(B, dtype=torch.float64, device="cuda"); pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        g.visual_track_hybrid_dev(vp, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
                                  r_gate, r_update, st.data_ptr(), gs.data_ptr(), chi.data_ptr(), pf.data_ptr())
        torch.cuda.synchronize()
        st, gs, chi, pf = st.cpu().numpy(), gs.cpu().numpy(), chi.cpu().numpy(), pf.cpu().numpy()
        seen = {"applied": 0, "inserted": 0, "map_applied": 0, "map_rejected": 0, "offer_rejected": 0}
        for b, o in enumerate(filters):
            m0 = o.m.copy()
            if map_update[b] >= 0:
                off = base + 3 * map_update[b]
                trail = oracle.extract_camera_pose_trail(m0, idx[b], T1, T2 if stereo else None)
                ops, oH, of = oracle.prepare_visual_update(m0[off:off + 3], None, None, np.zeros(3), vel[b], trail, idx[b], n, truncated=False,
                                                           map_point_offset=off)
                assert st[b].tolist() == [1, ops], (b, st[b].tolist(), ops)          # HV_TRI_HYBRID
                assert np.allclose(pf[b], m0[off:off + 3], rtol=0, atol=0)
                ok = ops == 0
            else:
                ost, ops, opf, oH, of = oracle.visual_track_prepare(par, m0, idx[b], T1, T2 if stereo else None, feat[b], vel[b])
                assert st[b].tolist() == [ost, ops], (b, st[b].tolist(), [ost, ops])
                ok = (ost, ops) == (0, 0)
                if ok:
                    assert _rel(pf[b], opf) < 1e-9
            if not ok:
                assert gs[b] == 1
            else:
                status, c2 = o.visual_track_outlier_check(oH, of, ys[b], r_gate)
                assert gs[b] == status and abs(chi[b] - c2) <= 1e-7 * max(1.0, abs(c2)), (b, gs[b], status, chi[b], c2)
                if status == 0:
                    if map_update[b] < 0 and map_offer[b] >= 0:
                        o.insert_map_point(int(map_offer[b]), opf); seen["inserted"] += 1
                    else:
                        o.update_visual_track(oH, of, ys[b], r_update); seen["map_applied" if map_update[b] >= 0 else "applied"] += 1
                elif map_update[b] >= 0:
                    seen["map_rejected"] += 1
                elif map_offer[b] >= 0:
                    seen["offer_rejected"] += 1
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, kind[b], _rel(mg, o.m), _rel(Pg, o.P))
        assert all(v > 0 for v in seen.values()), seen
        g.close()

@pytest.mark.parametrize("B,np_max,growth,rmse_thr,variant", [
    (48, 21, 1.5, 2.5, "sorted_small_batch"),      # two sorted length classes: vu_gate_kernel + vu_gate_long_kernel keep the per-filter multiplier
    (300, 21, 1.3, 2.5, "default"),                # ... more filters than CUs (two-per-CU build, second stream)
    (48, 21, 1.5, -1.0, "default"),                # growth without the RMSE test
    (10, 10, 1.0, 2.5, "default"),                 # few sequences: the speculative loop with the RMSE test (growth 1)
    (10, 10, 1.5, 2.5, "default"),                 # ... a growth factor forces the sequential loop
    (24, 10, 1.4, 2.2, "vu384")])
def test_frame_loop_with_adaptive_outlier_thresholds(oracle, B, np_max, growth, rmse_thr, variant):
    """backend.cpp:994-996,1159,1192-1193 inside hv_ekf_visual_frame_ragged_dev (ABI 3): visualTrackOutlierCheck's early RMSE test
    (ekf.cpp:797-801; gate status 2) and the per-session growth of BOTH thresholds after every rejected track, kept per filter on the
    device. Tracks carry no, a medium (2.0: passes the RMSE test, fails chi2 at trackChiTestOutlierR 1.5 when it is long enough) or a
    gross (3.0: fails the RMSE test) measurement offset, so that one frame sees RMSE rejections, chi2 rejections, and tracks that pass
    only because earlier rejections raised the thresholds. Reference: the oracle's loop with the same
    rule, per filter."""
    import torch
    rng = np.random.default_rng(4000 + B + np_max)
    trail_len, K, quota = 20, 8, 3
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, trail_len, 6, True, bad_fraction=0.0)
    lens = rng.integers(2, np_max + 1, (K, B)).astype(np.int32)
    lens[rng.uniform(size=(K, B)) < 0.1] = 0
    idx = np.zeros((K, B, np_max), np.int32); feat = np.zeros((K, B, 2 * np_max, 2)); vel = np.zeros_like(feat)
    ys = np.zeros((K, B, 4 * np_max))
    per = {}
    for k in range(K):
        for n in sorted(set(lens[k].tolist()) - {0}):
            sel = np.nonzero(lens[k] == n)[0]
            _, _, _, i_, f_, v_ = _random_tracks(oracle, rng, len(sel), trail_len, n, True, bad_fraction=0.0, given_means=means[sel])
            for j, b in enumerate(sel):
                off = rng.choice([0.0, 2.0, 3.0], p=[0.4, 0.35, 0.25])
                yy = f_[j].reshape(-1) + 2e-3 * rng.normal(size=f_[j].size) + off
                idx[k, b, :n] = i_[j]; feat[k, b, :2 * n] = f_[j]; vel[k, b, :2 * n] = v_[j]; ys[k, b, :4 * n] = yy
                per[(k, b)] = (i_[j], f_[j], v_[j], yy)
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2, trackRmseThreshold=rmse_thr,
                                trackOutlierThresholdGrowthFactor=growth)
    par = oracle.tri_default_params()
    r_gate, r_update = 1.5, 0.05
    with capi.Context(width=64, height=64) as ctx:
        _apply(ctx, variant)
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        filters = []
        for b in range(B):
            o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=trail_len))
            P = o.P.copy() * 1e-6 + np.eye(o.n) * 1e-4
            o.set_state(means[b]); o.set_cov(P)
            g.set_state(b, means[b], P)
            filters.append(o)
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        d = [dev(lens, np.int32), dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(ys, np.float64)]
        st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
        counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        for rep in range(2):                                   # twice from the same start: the multipliers are reset per frame
            for b in range(B):
                g.set_state(b, means[b], P)
            g.visual_frame_ragged_dev(vp, K, np_max, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                      r_gate, r_update, st.data_ptr(), gs.data_ptr(), counter.data_ptr(), quota)
            torch.cuda.synchronize()
        st, gs, counts = st.cpu().numpy(), gs.cpu().numpy(), counter.cpu().numpy()
        seen = {0: 0, 2: 0, 3: 0}
        grown_inliers = 0
        for b, o in enumerate(filters):
            done, scale = 0, 1.0
            for k in range(K):
                if done >= quota or lens[k, b] == 0:
                    assert st[k, b].tolist() == [-1, -1] and gs[k, b] == 1, (b, k)
                    continue
                i_, f_, v_, yy = per[(k, b)]
                ost, ops, _, oH, of = oracle.visual_track_prepare(par, o.m.copy(), i_, T1, T2, f_, v_)
                if st[k, b].tolist() != [ost, ops]:
                    assert not _well_conditioned(oracle.tri_last_diag()) and ost != 0 and st[k, b, 0] != 0, (b, k)
                if (ost, ops) != (0, 0):
                    assert gs[k, b] == 1
                    continue
                status, _ = o.visual_track_outlier_check(oH, of, yy, r_gate * scale, rmse_thr * scale if rmse_thr >= 0 else -1.0)
                assert gs[k, b] == status, (b, k, lens[k, b], gs[k, b], status, scale)
                seen[int(status)] += 1
                if status == 0:
                    if scale > 1.0:
                        base, _ = o.visual_track_outlier_check(oH, of, yy, r_gate, rmse_thr)
                        grown_inliers += base != 0             # passes only because of the growth
                    o.update_visual_track(oH, of, yy, r_update); done += 1
                else:
                    scale *= growth
            assert counts[b] == done
            mg, Pg = g.get_state(b)
            assert _rel(mg, o.m) < 1e-8 and _rel(Pg, o.P) < 1e-7, (b, _rel(mg, o.m), _rel(Pg, o.P))
        assert seen[0] > 0 and seen[3] > 0 and (rmse_thr < 0 or seen[2] > 0), seen
        assert growth == 1.0 or grown_inliers > 0, (seen, grown_inliers)
        g.close()


def test_adaptive_thresholds_are_refused_where_no_kernel_serves_them(oracle):
    """The dense gate kernels (knob ekf_fused_gate 0) and r03's two-launch long class know neither the RMSE test nor the per-filter
    growth: the frame entry point must say so (HV_ERR_UNSUPPORTED), never ignore the parameters."""
    import torch
    rng = np.random.default_rng(5)
    B, K, npose = 8, 3, 6
    T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, 20, npose, True, bad_fraction=0.0)
    for knobs, rmse, growth in (({"ekf_fused_gate": 0}, 0.3, 1.0), ({"ekf_fused_gate": 0, "ekf_no_speculation": 1}, -1.0, 1.5), ({"ekf_fused_gate": 2, "ekf_no_speculation": 1}, 0.3, 1.0)):
        with capi.Context(width=64, height=64) as ctx:
            for k_, v_ in knobs.items():
                ctx.set_knob(k_, v_)
            g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=20), B)
            for b in range(B):
                g.set_state(b, means[b], np.eye(g.n) * 1e-4)
            vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2, trackRmseThreshold=rmse, trackOutlierThresholdGrowthFactor=growth)
            dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
            d = [dev(np.stack([idx] * K), np.int32), dev(np.stack([feat] * K), np.float64), dev(np.stack([vel] * K), np.float64),
                 dev(np.stack([feat.reshape(B, -1)] * K), np.float64)]
            st = torch.zeros((K, B, 2), dtype=torch.int32, device="cuda"); gs = torch.zeros((K, B), dtype=torch.int32, device="cuda")
            counter = torch.zeros((B,), dtype=torch.int32, device="cuda")
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            with pytest.raises(capi.HvError, match="unsupported"):
                g.visual_frame_dev(vp, K, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 1.5, 0.05,
                                   st.data_ptr(), gs.data_ptr(), counter.data_ptr(), 2)
            torch.cuda.synchronize()
            g.close()


def test_long_trail_track_is_rejected_not_corrupted():
    """cameraTrailLength > 20 is a valid filter size, but the prepare kernel's LDS arrays hold 21 poses per camera:
    a 22-pose mono track must come back as HV_ERR_UNSUPPORTED (r01 advisor: it used to overrun s_dpf / s_idx silently);
    a 21-pose track on the same filter still works."""
    rng = np.random.default_rng(5)
    with capi.Context(width=64, height=64, levels=1, pool_size=1) as ctx:
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=24), 1)
        assert g.n == 20 + 7 * 24
        T1, _, means, idx, feat = synth.visual_tracks(rng, 1, 24, 22, False)
        g.set_state(0, means[0], np.eye(g.n) * 1e-4)
        vp = capi.vu_default_params(imu_to_camera=T1)
        with pytest.raises(capi.HvError, match="unsupported"):
            g.visual_track(vp, idx, feat, np.zeros_like(feat), feat.reshape(1, -1), 1.5, 0.05)
        T1, _, _, idx21, feat21 = synth.visual_tracks(rng, 1, 24, 21, False, given_means=means)
        st, gs, _, _ = g.visual_track(vp, idx21, feat21, np.zeros_like(feat21), feat21.reshape(1, -1), 1.5, 0.05)
        assert st[0, 0] != -1
        g.close()
