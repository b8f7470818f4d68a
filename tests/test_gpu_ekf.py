"""GPU parity tests of the HIP EKF (hv_ekf_* in include/hybvio_hip.h) against the CPU oracle
(oracle/ekf_oracle.c, itself pinned by the reference's EKF tests). Bar (BASELINE.json north_star /
SURVEY.md 8d): ||dm|| / ||m|| <= 1e-5 and ||dP||_F / ||P||_F <= 1e-5 per call with identical inputs
injected, and after a closed-loop replay. The device uses restructured algebra (Cholesky on the
tall matrix, P -= Y'Y, rank-7 Joseph form), so equality is to rounding, not bitwise.
"""
import os

import numpy as np
import pytest

from hybvio_amd import capi

pytestmark = pytest.mark.gpu
TOL = 1e-5
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ekf_reference_fixtures.npz")
POS, VEL, ORI, BGA, BAA, BAT, SFT, CAM = 0, 3, 6, 10, 13, 16, 19, 20
PARAM_KEYS = [k for k, _ in capi.EkfParams._fields_]


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def same_params(oracle, **over):
    po = oracle.ekf_default_params(**over)
    pg = capi.ekf_default_params(**over)
    for k in PARAM_KEYS:
        assert getattr(po, k) == getattr(pg, k), k     # both default sets mirror parameter_definitions.c
    return po, pg


def random_state(rng, n, trail, scale=1e2):
    m = rng.normal(size=n)
    m[ORI:ORI + 4] /= np.linalg.norm(m[ORI:ORI + 4])
    m[BAT:BAT + 3] = 1 + 0.01 * rng.normal(size=3)
    for c in range(trail):
        m[CAM + 7 * c + 3: CAM + 7 * c + 7] /= np.linalg.norm(m[CAM + 7 * c + 3: CAM + 7 * c + 7])
    A = rng.normal(size=(n, n))
    return m, scale * (A @ A.T / n + 0.1 * np.eye(n))


@pytest.fixture()
def ctx():
    c = capi.Context(width=64, height=64, levels=1, pool_size=1)
    yield c
    c.close()


def make_pair(oracle, ctx, rng, batch=1, trail=20):
    po, pg = same_params(oracle, cameraTrailLength=trail)
    g = capi.EkfBatch(ctx, pg, batch)
    os_ = []
    for b in range(batch):
        o = oracle.Ekf(po)
        m, P = random_state(rng, o.n, trail)
        o.set_state(m); o.set_cov(P)
        g.set_state(b, m, P)
        os_.append(o)
    return os_, g


def check(os_, g, tol=TOL):
    for b, o in enumerate(os_):
        m, P = g.get_state(b)
        assert rel(m, o.m) <= tol, (b, rel(m, o.m))
        assert rel(P, o.P) <= tol, (b, rel(P, o.P))
    return max(max(rel(g.get_state(b)[0], o.m), rel(g.get_state(b)[1], o.P)) for b, o in enumerate(os_))


def test_constructor_state_matches_reference_initialisation(oracle, ctx):
    for trail in (20, 5, 1):                     # 1: the SLAM-coordinate helper filter (backend.cpp:55-56)
        po, pg = same_params(oracle, cameraTrailLength=trail)
        o, g = oracle.Ekf(po), capi.EkfBatch(ctx, pg, 2)
        assert g.n == o.n == 20 + 7 * trail
        for b in range(2):
            m, P = g.get_state(b)
            np.testing.assert_array_equal(m, o.m)
            np.testing.assert_array_equal(P, o.P)


def test_predict_parity_batch(oracle, ctx):
    rng = np.random.default_rng(1)
    os_, g = make_pair(oracle, ctx, rng, batch=3)
    for o in os_:
        o.set_first_sample_time(1.0)
    t = np.full(3, 1.0)
    for step in range(12):
        dt = np.array([0.005, 0.004, 0.0 if step == 3 else 0.006])     # one filter skips a sample
        gy, ac = rng.normal(0, 0.3, (3, 3)), rng.normal(0, 1.0, (3, 3)) + [0, 0, 9.8]
        g.predict(dt, gy, ac)
        t += dt
        for b, o in enumerate(os_):
            o.predict(t[b], gy[b], ac[b])
    worst = check(os_, g)
    assert worst < 1e-11
    np.testing.assert_allclose(g.get_dydx(0), os_[0].dydx, rtol=1e-12, atol=1e-14)


def test_predict_n_samples_in_one_launch(oracle, ctx):
    """hv_ekf_predict_n_dev == n sequential predicts (oracle), incl. a skipped sample (dt <= 0) and
    per-filter inputs; the off-diagonal blocks see the product of the samples' F once."""
    import torch
    rng = np.random.default_rng(31)
    for trail in (20, 5):
        os_, g = make_pair(oracle, ctx, rng, batch=3, trail=trail)
        nS = 10
        dt = np.full((nS, 3), 0.005); dt[4, 1] = 0.0; dt[7, :] = 0.0025
        gy = rng.normal(0, 0.05, (nS, 3, 3)); ac = rng.normal(0, 0.05, (nS, 3, 3)) + [0.1, -0.2, 9.8]
        for o in os_:
            o.set_first_sample_time(1.0)
        tt = np.full(3, 1.0)
        for s_ in range(nS):
            tt += dt[s_]
            for b, o in enumerate(os_):
                o.predict(tt[b], gy[s_, b], ac[s_, b])          # dt == 0 -> the oracle skips the sample too
        d_dt, d_gy, d_ac = (torch.from_numpy(x).cuda() for x in (dt, gy, ac))
        g.predict_n_dev(nS, d_dt.data_ptr(), d_gy.data_ptr(), d_ac.data_ptr())
        assert check(os_, g) < 1e-11
        np.testing.assert_allclose(g.get_dydx(0), os_[0].dydx, rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("B,nS,trail", [(1, 10, 20), (3, 7, 20), (3, 12, 5), (300, 10, 20), (2, 1, 20)])
def test_predict_kernel_forms_agree_to_the_bit(oracle, ctx, B, nS, trail):
    """ekf_predict_chain_kernel (knob 2: every launch; default 1: launches of three or more samples -- the samples' mean recursion on one wavefront, F / L of up to five samples per pass) against
    ekf_predict_kernel (knob ekf_predict_chain 0: nine barrier-separated stages per sample): the same expressions in the same order --
    mean, covariance and dydx identical to the bit, with skipped samples (dt <= 0), a sample that changes dt, more samples than a chunk."""
    import torch
    rng = np.random.default_rng(1000 * B + nS)
    os_, g = make_pair(oracle, ctx, rng, batch=B, trail=trail)
    dt = np.full((nS, B), 0.005)
    if nS > 4:
        dt[4, B // 2] = 0.0; dt[nS - 2, :] = 0.0025; dt[nS - 1, 0] = 0.0
    gy = rng.normal(0, 0.05, (nS, B, 3)); ac = rng.normal(0, 0.05, (nS, B, 3)) + [0.1, -0.2, 9.8]
    d_dt, d_gy, d_ac = (torch.from_numpy(x).cuda() for x in (dt, gy, ac))
    states = [g.get_state(b) for b in range(B)]
    out = {}
    for form in (2, 0):
        for b in range(B):
            g.set_state(b, *states[b])
        ctx.set_knob("ekf_predict_chain", form)
        assert ctx.get_knob("ekf_predict_chain") == form
        g.predict_n_dev(nS, d_dt.data_ptr(), d_gy.data_ptr(), d_ac.data_ptr())
        out[form] = ([g.get_state(b) for b in range(0, B, max(1, B // 7))], g.get_dydx(0).copy(), g.get_dydx(B - 1).copy())
    ctx.set_knob("ekf_predict_chain", 1)
    for (m1, P1), (m0, P0) in zip(out[2][0], out[0][0]):
        assert np.array_equal(m1, m0) and np.array_equal(P1, P0)
    assert np.array_equal(out[2][1], out[0][1]) and np.array_equal(out[2][2], out[0][2])
    assert np.isfinite(out[2][0][0][1]).all() and not np.array_equal(out[2][0][0][1], states[0][1])


def test_reference_der_predict_on_gpu(oracle, ctx):
    """test/ekf.cpp:73-117 run against the HIP predict: analytic dydx vs forward differences < 1e-3."""
    fx = np.load(GOLD)
    _, pg = same_params(oracle, cameraTrailLength=5)
    g = capi.EkfBatch(ctx, pg, 1)
    m0 = np.zeros(g.n)
    m0[POS:POS + 3], m0[ORI:ORI + 4] = fx["poses"][0:3], fx["poses"][3:7]
    m0[BAT:BAT + 3] = 1.0
    _, P0 = g.get_state(0)

    def run(x):
        m = m0.copy(); m[:20] = x
        g.set_state(0, m, P0)
        g.predict(0.01, fx["gyro"], fx["acc"])
        return g.get_state(0)[0][:20], g.get_dydx(0)

    f0, F = run(m0[:20])
    D = np.zeros((20, 20))
    for i in range(20):
        x = m0[:20].copy(); x[i] += 1e-7
        D[:, i] = (run(x)[0] - f0) / 1e-7
    assert np.abs(D - F).max() < 1e-3


# (rows, columns of H): the all-in-LDS kernel with 1 / 2 / 3 row tiles incl. its 47-row limit and ragged
# K halves (l = 41, 76, 100), the LDS + streamed-H kernel (48, 64) and the global-workspace kernel (84, 128)
@pytest.mark.parametrize("nr,l", [(40, 160), (84, 160), (16, 76), (8, 41), (3, 160), (128, 160),
                                  (47, 160), (48, 160), (33, 160), (64, 160), (17, 100), (32, 27)])
def test_visual_gate_and_update_parity(oracle, ctx, nr, l):
    rng = np.random.default_rng(nr * 7 + l)
    os_, g = make_pair(oracle, ctx, rng, batch=2)
    H = rng.normal(size=(2, nr, l))
    f = rng.normal(size=(2, nr))
    y = f + 0.05 * rng.normal(size=(2, nr))
    y[1] = f[1] + 5.0 * rng.normal(size=nr)                 # filter 1: a gross outlier
    r = 0.05
    before = [gg.copy() for gg in g.get_state(0)]
    chi2, st = g.visual_gate(H, y - f, r)
    for b, o in enumerate(os_):
        so, co = o.visual_track_outlier_check(H[b], f[b], y[b], r)
        assert st[b] == so and abs(chi2[b] - co) <= 1e-9 * max(1.0, abs(co)), (b, chi2[b], co)
    assert st[1] == 3 and rel(g.get_state(0)[1], before[1]) == 0.0      # the gate leaves the filter untouched
    g.visual_update(H, y - f, r, active=[1, 0])             # apply only to the inlier filter
    os_[0].update_visual_track(H[0], f[0], y[0], r)
    assert check(os_, g) < 1e-9


def test_fused_gate_update_on_device(oracle, ctx):
    """mode 2 (bench path): inputs resident on the device, update applied only where the gate passes."""
    import torch
    rng = np.random.default_rng(4)
    os_, g = make_pair(oracle, ctx, rng, batch=3)
    nr, l, r = 40, 160, 0.05
    H = rng.normal(size=(3, nr, l))
    v = 0.05 * rng.normal(size=(3, nr))
    v[2] = 4.0 * rng.normal(size=nr)
    g.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dH = torch.from_numpy(np.ascontiguousarray(np.transpose(H, (0, 2, 1)))).cuda()
    dv = torch.from_numpy(v).cuda()
    chi2 = torch.zeros(3, dtype=torch.float64, device="cuda")
    st = torch.zeros(3, dtype=torch.int32, device="cuda")
    g.visual_dev(nr, l, dH.data_ptr(), dv.data_ptr(), r, 2, chi2.data_ptr(), st.data_ptr())
    torch.cuda.synchronize()
    st = st.cpu().numpy()
    assert list(st) == [0, 0, 3]
    for b, o in enumerate(os_):
        so, co = o.visual_track_outlier_check(H[b], np.zeros(nr), v[b], r)
        assert so == st[b]
        if so == 0:
            o.update_visual_track(H[b], np.zeros(nr), v[b], r)
    check(os_, g)


def test_generic_updates_parity(oracle, ctx):
    """updateZupt / updateZrupt / updatePosition-style truncated updates (ekf.cpp:573-677)."""
    rng = np.random.default_rng(8)
    os_, g = make_pair(oracle, ctx, rng, batch=2)
    ns = os_[0].params.noiseScale ** 2
    for o in os_:
        o.set_first_sample_time(10.0)
    H = np.zeros((3, 6)); H[:, 3:6] = np.eye(3)
    g.update(H, np.zeros(3), 1e-3 * ns)
    for o in os_:
        o.update_zupt(1e-3)
    check(os_, g)
    H = np.zeros((3, 13)); H[:, 10:13] = np.eye(3)
    xg = np.array([0.01, -0.02, 0.005])
    g.update(H, xg, os_[0].params.rotationZuptR * ns)
    for o in os_:
        o.update_zrupt(xg)
    check(os_, g)
    H = np.zeros((3, 3)); H[:, 0:3] = np.eye(3)
    g.update(H, np.array([1.0, 2.0, 3.0]), 0.1 * ns)
    g.symmetrize()
    for o in os_:
        o.update_position(np.array([1.0, 2.0, 3.0]), 0.1)
    check(os_, g)
    H = np.zeros((4, 10)); H[:, 6:10] = np.eye(4)
    q = np.array([0.9, 0.1, -0.2, 0.3]); q /= np.linalg.norm(q)
    g.update(H, q, 0.01 * ns, normalize_all=True)
    g.symmetrize()
    for o in os_:
        o.update_orientation(q, 0.01)
    check(os_, g)


@pytest.mark.parametrize("k", [19, 18, 17, 16, 0, -1])
def test_pose_augmentation_and_undo_parity(oracle, ctx, k):
    rng = np.random.default_rng(50 + k)
    os_, g = make_pair(oracle, ctx, rng, batch=2)
    g.augment([k, k])
    for o in os_:
        o.update_visual_pose_augmentation(k)
    worst = check(os_, g)
    assert worst < 1e-9
    _, P = g.get_state(0)
    assert np.array_equal(P, P.T)                           # maintainPositiveSemiDefinite is fused in
    g.undo_augment()
    for o in os_:
        o.update_undo_augmentation()
    check(os_, g)


def test_symmetrize_is_exact_and_undo_respects_active_mask(oracle, ctx):
    """P = (P + P') / 2 bit for bit on an asymmetric P (tile-pair kernel incl. the ragged last tile), and
    the undo shift leaves masked-out filters alone although the covariance buffers are ping-ponged."""
    rng = np.random.default_rng(5)
    for trail in (20, 5, 1):
        os_, g = make_pair(oracle, ctx, rng, batch=3, trail=trail)
        n = g.n
        Ps = []
        for b in range(3):
            m, _ = g.get_state(b)
            P = rng.normal(size=(n, n))
            g.set_state(b, m, P)
            Ps.append(P)
        g.symmetrize()
        for b in range(3):
            assert np.array_equal(g.get_state(b)[1], 0.5 * (Ps[b] + Ps[b].T))
        g.undo_augment(active=[1, 0, 1])
        assert np.array_equal(g.get_state(1)[1], 0.5 * (Ps[1] + Ps[1].T))
        os_[0].set_cov(0.5 * (Ps[0] + Ps[0].T))
        os_[0].update_undo_augmentation()
        assert rel(g.get_state(0)[1], os_[0].P) < 1e-15


def test_symmetrize_folded_into_the_augmentation_is_bit_identical(oracle, ctx):
    """hv_ekf_symmetrize_augment_dev == hv_ekf_symmetrize then hv_ekf_augment_dev, bit for bit, on asymmetric covariances
    (as the visual updates leave them), per-filter discard indices and an active mask: masked-out filters come out symmetrised
    and otherwise untouched."""
    import torch
    rng = np.random.default_rng(11)
    for trail in (20, 5, 1):
        B = 5
        _, ga = make_pair(oracle, ctx, rng, batch=B, trail=trail)
        _, gb = make_pair(oracle, ctx, rng, batch=B, trail=trail)
        n = ga.n
        for b in range(B):
            m, P = ga.get_state(b)
            P = P + 1e-3 * np.abs(P).max() * rng.normal(size=(n, n))      # not symmetric
            ga.set_state(b, m, P); gb.set_state(b, m, P)
        ks = rng.integers(-1, trail, size=B).astype(np.int32)
        for active in (None, np.array([1, 0, 1, 1, 0], np.uint8)):
            kd = torch.from_numpy(ks).cuda()
            ad = torch.from_numpy(active).cuda() if active is not None else None
            ap = ad.data_ptr() if ad is not None else 0
            ga.symmetrize(); ga.augment_dev(kd.data_ptr(), ap)
            gb.symmetrize_augment_dev(kd.data_ptr(), ap)
            ctx.synchronize()
            for b in range(B):
                ma, Pa = ga.get_state(b); mb, Pb = gb.get_state(b)
                assert np.array_equal(ma, mb) and np.array_equal(Pa, Pb), (trail, b)
                assert np.array_equal(Pb, Pb.T)
            for b in range(B):                                              # asymmetric again for the masked round
                m, P = ga.get_state(b)
                P = P + 1e-3 * np.abs(P).max() * rng.normal(size=(n, n))
                ga.set_state(b, m, P); gb.set_state(b, m, P)


def test_mixed_discard_indices_and_active_mask(oracle, ctx):
    rng = np.random.default_rng(77)
    os_, g = make_pair(oracle, ctx, rng, batch=4)
    ks = [19, 16, 18, 17]
    g.augment(ks, active=[1, 1, 0, 1])
    for b, o in enumerate(os_):
        if b != 2:
            o.update_visual_pose_augmentation(ks[b])
    check(os_, g)


def test_reference_transform_to_round_trip_on_gpu(oracle, ctx):
    """test/ekf.cpp:119-145 against hv_ekf_transform, with the adapter-side quaternion math done here."""
    fx = np.load(GOLD)
    po, pg = same_params(oracle, cameraTrailLength=5)
    g = capi.EkfBatch(ctx, pg, 1)
    g.set_state(0, fx["m55"], fx["P55"])
    A = 2

    def transform_to(pos, q, idx):
        m, _ = g.get_state(0)
        q0 = m[CAM + 7 * idx + 3: CAM + 7 * idx + 7]
        a = np.array([q0[0], -q0[1], -q0[2], -q0[3]])
        p1 = a[0]*q[0] - a[1]*q[1] - a[2]*q[2] - a[3]*q[3]
        p2 = a[0]*q[1] + a[1]*q[0] + a[2]*q[3] - a[3]*q[2]
        p3 = a[0]*q[2] - a[1]*q[3] + a[2]*q[0] + a[3]*q[1]
        p4 = a[0]*q[3] + a[1]*q[2] - a[2]*q[1] + a[3]*q[0]
        qC = np.array([[p1, -p2, -p3, -p4], [p2, p1, p4, -p3], [p3, -p4, p1, p2], [p4, p3, -p2, p1]])
        w, x, y, z = p1, p2, p3, p4
        R = np.array([[1 - 2*(y*y + z*z), 2*(x*y - z*w), 2*(x*z + y*w)],
                      [2*(x*y + z*w), 1 - 2*(x*x + z*z), 2*(y*z - x*w)],
                      [2*(x*z - y*w), 2*(y*z + x*w), 1 - 2*(x*x + y*y)]])
        pC = R.T
        tr = pos - pC @ m[CAM + 7 * idx: CAM + 7 * idx + 3]
        g.transform(0, pC, qC, tr)

    m0, P0 = g.get_state(0)
    pos0, rot0 = m0[CAM + 7 * A: CAM + 7 * A + 3].copy(), m0[CAM + 7 * A + 3: CAM + 7 * A + 7].copy()
    o = oracle.Ekf(po); o.set_state(fx["m55"]); o.set_cov(fx["P55"])
    to_pos, to_rot = np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0, 0.0])
    transform_to(to_pos, to_rot, A)
    o.transform_to(to_pos, to_rot, A)
    m1, P1 = g.get_state(0)
    assert np.linalg.norm(m1[CAM + 7 * A: CAM + 7 * A + 3] - to_pos) < 1e-6
    assert np.linalg.norm(m1[CAM + 7 * A + 3: CAM + 7 * A + 7] - to_rot) < 1e-6
    assert rel(m1, o.m) < 1e-12 and rel(P1, o.P) < 1e-12
    transform_to(pos0, rot0, A)
    m2, P2 = g.get_state(0)
    assert np.linalg.norm(m2 - fx["m55"]) < 1e-3 and np.linalg.norm(P2 - fx["P55"]) < 1e-3


@pytest.mark.parametrize("frames", [60, 400, 2000])
def test_closed_loop_replay(oracle, ctx, frames):
    """60 / 400 / 2000 (the length SURVEY.md 8(d) names) frames of the per-frame EKF call sequence: 10 predicts, up to 8 gated
    visual updates, symmetrise, augmentation with the Hanoi discard pattern; same inputs to both.
    Visual updates start once every trail slot has been cloned from a real pose (as in the reference,
    where a track needs >= 4 frames): before that P mixes 1e8 prior variances with 1e-6 ones and the
    subtractive update P -= K HP itself loses ~10 digits in BOTH implementations."""
    rng = np.random.default_rng(2024)
    po, pg = same_params(oracle)
    o, g = oracle.Ekf(po), capi.EkfBatch(ctx, pg, 1)
    acc0 = np.array([0.2, -0.1, 9.8])
    o.initialize_orientation(acc0)
    g.set_state(0, o.m.copy(), o.P.copy())
    o.set_first_sample_time(0.0)
    t, applied, rejected = 0.0, 0, 0
    for frame in range(frames):
        for _ in range(10):
            t += 0.005
            gy, ac = rng.normal(0, 0.05, 3), acc0 + rng.normal(0, 0.05, 3)
            o.predict(t, gy, ac)
            g.predict(0.005, gy, ac)
        if frame >= 24:
            for _ in range(8):
                poses = int(rng.integers(4, 11))
                nr, l = 4 * poses, 20 + 7 * int(rng.integers(poses, 21))
                H = rng.normal(size=(nr, l))
                v = rng.normal(size=nr) * (0.5 if rng.random() < 0.25 else 0.02)   # some gross outliers
                so, _ = o.visual_track_outlier_check(H, np.zeros(nr), v, 0.05)
                _, sg = g.visual_gate(H, v, 0.05)
                assert so == sg[0]
                if so == 0:
                    o.update_visual_track(H, np.zeros(nr), v, 0.05)
                    g.visual_update(H, v, 0.05)
                    applied += 1
                else:
                    rejected += 1
            o.maintain_psd(); g.symmetrize()
        k = [19, 16, 17, 16, 18, 16, 17, 16][frame % 8] if frame >= 20 else -1
        o.update_visual_pose_augmentation(k)
        g.augment([k])
    assert applied > 10 and rejected > 10
    m, P = g.get_state(0)
    print(f"closed loop {frames} frames: {applied} updates, {rejected} rejected, rel err m {rel(m, o.m):.2e} P {rel(P, o.P):.2e}")
    assert rel(m, o.m) <= TOL and rel(P, o.P) <= TOL, (rel(m, o.m), rel(P, o.P))


@pytest.mark.parametrize("seed", range(10))
def test_randomised_update_shapes(oracle, ctx, seed):
    """Seeded sweep over (trail length, measurement rows, columns of H): gate status / chi2 and the updated
    state against the oracle, whatever kernel variant and tile raggedness the shape selects."""
    rng = np.random.default_rng(500 + seed)
    trail = int(rng.choice([1, 3, 5, 12, 20]))
    os_, g = make_pair(oracle, ctx, rng, batch=2, trail=trail)
    n = g.n
    nr = int(rng.integers(1, min(n, 100) + 1))
    l = int(rng.integers(max(20, min(nr, n) // 2), n + 1))
    H = rng.normal(size=(2, nr, l))
    f = rng.normal(size=(2, nr))
    y = f + 0.05 * rng.normal(size=(2, nr))
    chi2, st = g.visual_gate(H, y - f, 0.05)
    for b, o in enumerate(os_):
        so, co = o.visual_track_outlier_check(H[b], f[b], y[b], 0.05)
        assert st[b] == so and abs(chi2[b] - co) <= 1e-9 * max(1.0, abs(co)), (trail, nr, l, b, chi2[b], co)
    g.visual_update(H, y - f, 0.05)
    for b, o in enumerate(os_):
        o.update_visual_track(H[b], f[b], y[b], 0.05)
    assert check(os_, g) < 1e-9, (trail, nr, l)


def test_hybrid_map_state_n_above_160(oracle, ctx):
    """hybridMapSize > 0 (state 20 + 7*20 + 3*15 = 205): every kernel on a state wider than the 160-column fast
    path -- predict (off-diagonal tiles beyond the trail), augmentation / undo (map rows must not shift,
    dynamic LDS above 64 KB), symmetrise, gate and update (LDS / global-workspace kernels)."""
    rng = np.random.default_rng(99)
    po, pg = same_params(oracle, hybridMapSize=15)
    g = capi.EkfBatch(ctx, pg, 2)
    os_ = []
    for b in range(2):
        o = oracle.Ekf(po)
        m, P = random_state(rng, o.n, 20)
        o.set_state(m); o.set_cov(P); g.set_state(b, m, P); os_.append(o)
    assert g.n == 205
    for o in os_:
        o.set_first_sample_time(0.0)
    t = 0.0
    for k in (19, 17, -1):
        for _ in range(3):
            t += 0.005
            gy, ac = rng.normal(0, 0.1, (2, 3)), rng.normal(0, 0.5, (2, 3)) + [0, 0, 9.8]
            g.predict(np.full(2, 0.005), gy, ac)
            for b, o in enumerate(os_):
                o.predict(t, gy[b], ac[b])
        g.augment([k, k])
        for o in os_:
            o.update_visual_pose_augmentation(k)
        assert check(os_, g) < 1e-9
    g.undo_augment()
    for o in os_:
        o.update_undo_augmentation()
    g.symmetrize()
    for o in os_:
        o.maintain_psd()
    assert check(os_, g) < 1e-9
    for nr, l in ((24, 205), (60, 160), (8, 97)):
        H = rng.normal(size=(2, nr, l)); f = rng.normal(size=(2, nr)); y = f + 0.05 * rng.normal(size=(2, nr))
        chi2, st = g.visual_gate(H, y - f, 0.05)
        for b, o in enumerate(os_):
            so, co = o.visual_track_outlier_check(H[b], f[b], y[b], 0.05)
            assert st[b] == so and abs(chi2[b] - co) <= 1e-9 * max(1.0, abs(co))
        g.visual_update(H, y - f, 0.05)
        for b, o in enumerate(os_):
            o.update_visual_track(H[b], f[b], y[b], 0.05)
        assert check(os_, g) < 1e-9


def test_singular_innovation_covariance_leaves_the_filter_untouched(oracle, ctx):
    """r = 0 with a rank-deficient H makes S = H P H' singular: the blocked Cholesky meets a non-positive pivot. The filter must
    come back unchanged and flagged (status CHI2) from the gate AND from a plain update, never with NaNs in m / P (r01 advisor)."""
    rng = np.random.default_rng(8)
    os_, g = make_pair(oracle, ctx, rng, batch=2)
    nr, l = 12, 76
    H = rng.normal(size=(2, nr, l))
    H[0, 6:] = H[0, :6]                                     # filter 0: duplicated rows -> S exactly singular at r = 0
    v = 0.01 * rng.normal(size=(2, nr))
    before = [[x.copy() for x in g.get_state(b)] for b in range(2)]
    chi2, st = g.visual_gate(H, v, 0.0)
    assert st[0] == 3                                       # broken pivot reported as an outlier
    g.visual_update(H, v, 0.0)
    m0, P0 = g.get_state(0)
    assert np.isfinite(m0).all() and np.isfinite(P0).all()
    assert rel(m0, before[0][0]) == 0.0 and rel(P0, before[0][1]) == 0.0
    m1, P1 = g.get_state(1)                                 # filter 1 (full rank, r = 0 still positive definite) was updated
    assert np.isfinite(P1).all() and rel(m1, before[1][0]) > 0.0
