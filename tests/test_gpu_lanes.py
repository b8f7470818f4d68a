"""GPU parity of the configuration the benchmark's headline runs: TWO lanes of one hv_lanes set (each its own
hv context on library-owned high-priority streams, include/hybvio_hip.h "lanes"), 300 DISTINCT ragged filters per lane with tracks of
2 .. 21 stereo poses, whole frame loops (hv_ekf_visual_frame_ragged_dev: sorted two-class visits, the long class's launch on the
context's second stream, shared update grids) enqueued on both lanes' streams at once -- eagerly and as captured HIP graphs replayed
beside each other -- every filter against the oracle's sequential per-track loop (backend.cpp:1012-1252 semantics: per filter the
visits are sequential, the filters are independent)."""
import numpy as np
import pytest

from hybvio_amd import capi

pytestmark = pytest.mark.gpu

B, K, NP_MAX, QUOTA, TRAIL = 300, 6, 21, 3, 20      # (B: the two-lane test; the four-lane test runs 64 filters per lane)
R_GATE, R_UPDATE = 1.5, 0.05


def _rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _problem(oracle, seed, B=B):
    """One lane's inputs (host arrays) and what the oracle's sequential loop makes of them."""
    from test_gpu_visual_prepare import _random_tracks, _well_conditioned
    rng = np.random.default_rng(seed)
    T1, T2, means, _, _, _ = _random_tracks(oracle, rng, B, TRAIL, 6, True, bad_fraction=0.0)
    lens = rng.integers(2, NP_MAX + 1, (K, B)).astype(np.int32)
    lens[rng.uniform(size=(K, B)) < 0.1] = 0                                      # no candidate track for this filter at this visit
    lens[0, 0], lens[1, 1], lens[2, 2] = NP_MAX, 2, 12                            # both ends and the 48-row edge of the short class
    idx = np.zeros((K, B, NP_MAX), np.int32); feat = np.zeros((K, B, 2 * NP_MAX, 2)); vel = np.zeros_like(feat)
    ys = np.zeros((K, B, 4 * NP_MAX))
    per = {}
    for k in range(K):
        for n in sorted(set(lens[k].tolist()) - {0}):
            sel = np.nonzero(lens[k] == n)[0]
            _, _, _, i_, f_, v_ = _random_tracks(oracle, rng, len(sel), TRAIL, n, True, bad_fraction=0.0, given_means=means[sel])
            for j, b in enumerate(sel):
                yy = f_[j].reshape(-1) + 2e-3 * rng.normal(size=f_[j].size) + (3.0 if rng.uniform() < 0.5 else 0.0)
                idx[k, b, :n] = i_[j]; feat[k, b, :2 * n] = f_[j]; vel[k, b, :2 * n] = v_[j]; ys[k, b, :4 * n] = yy
                per[(k, b)] = (i_[j], f_[j], v_[j], yy)
    P0 = np.zeros((B, 160, 160))
    par = oracle.tri_default_params()
    exp_st = np.full((K, B, 2), -1, np.int32); exp_gs = np.ones((K, B), np.int32); exp_cnt = np.zeros(B, np.int32)
    exp_m, exp_P = np.zeros((B, 160)), np.zeros((B, 160, 160))
    well = np.ones((K, B), bool)                    # tracks whose triangulation status is reproducible to the bit (test_gpu_visual_prepare)
    long_applied = short_applied = rejected = 0
    for b in range(B):
        o = oracle.Ekf(oracle.ekf_default_params(cameraTrailLength=TRAIL))
        A = rng.normal(size=(o.n, o.n)) * 0.01
        P0[b] = o.P * 1e-6 + A @ A.T * 1e-3 + np.eye(o.n) * 1e-4                  # a distinct SPD covariance per filter
        o.set_state(means[b]); o.set_cov(P0[b])
        done = 0
        for k in range(K):
            if done >= QUOTA or lens[k, b] == 0:
                continue                                                          # not visited: (-1, -1), NOT_COMPUTED
            i_, f_, v_, yy = per[(k, b)]
            ost, ops, _, oH, of = oracle.visual_track_prepare(par, o.m.copy(), i_, T1, T2, f_, v_)
            exp_st[k, b] = (ost, ops)
            well[k, b] = _well_conditioned(oracle.tri_last_diag())
            if (ost, ops) != (0, 0):
                continue
            status, _ = o.visual_track_outlier_check(oH, of, yy, R_GATE)
            exp_gs[k, b] = status
            if status == 0:
                o.update_visual_track(oH, of, yy, R_UPDATE); done += 1
                long_applied += lens[k, b] > 11; short_applied += lens[k, b] <= 11
            else:
                rejected += 1
        exp_cnt[b] = done
        o.maintain_psd()
        exp_m[b], exp_P[b] = o.m, o.P
    assert long_applied > B // 4 and short_applied > B // 4 and rejected > B, (long_applied, short_applied, rejected)
    return dict(B=B, T1=T1, T2=T2, means=means, P0=P0, lens=lens, idx=idx, feat=feat, vel=vel, ys=ys,
                exp=(exp_st, exp_gs, exp_cnt, exp_m, exp_P), well=well)


@pytest.fixture(scope="module")
def problems(oracle):
    return [_problem(oracle, 9001), _problem(oracle, 9002)]


class _DevView:                                     # zero-copy torch view of the library's device buffers
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False), "version": 2}


class _Lane:
    def __init__(self, ctx, prob):
        import torch
        self.t, self.ctx, self.prob = torch, ctx, prob
        B = self.B = prob["B"]
        self.g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=TRAIL), B)
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        self.m0, self.P0 = dev(prob["means"], np.float64), dev(prob["P0"], np.float64)
        self.d = [dev(prob["lens"], np.int32), dev(prob["idx"], np.int32), dev(prob["feat"], np.float64), dev(prob["vel"], np.float64),
                  dev(prob["ys"], np.float64)]
        self.st = torch.full((K, B, 2), -9, dtype=torch.int32, device="cuda"); self.gs = torch.full((K, B), -9, dtype=torch.int32, device="cuda")
        self.counter = torch.full((B,), 77, dtype=torch.int32, device="cuda")
        self.vp = capi.vu_default_params(imu_to_camera=prob["T1"], second_imu_to_camera=prob["T2"])
        self.stream = torch.cuda.ExternalStream(ctx.get_stream())

    def reset(self):                                # on the torch default stream; the caller synchronises
        B = self.B
        mp, pp = self.g.device_pointers()
        self.t.as_tensor(_DevView(mp, (B, 160)), device="cuda").copy_(self.m0)
        self.t.as_tensor(_DevView(pp, (B, 160, 160)), device="cuda").copy_(self.P0)
        self.st.fill_(-9); self.gs.fill_(-9); self.counter.fill_(77)

    def frame(self):                                # asynchronous on the lane's stream (the library's own, high priority)
        d = self.d
        self.g.visual_frame_ragged_dev(self.vp, K, NP_MAX, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(),
                                       R_GATE, R_UPDATE, self.st.data_ptr(), self.gs.data_ptr(), self.counter.data_ptr(), QUOTA)
        self.g.symmetrize()

    def check(self, tag):
        B = self.B
        exp_st, exp_gs, exp_cnt, exp_m, exp_P = self.prob["exp"]
        st, gs, cnt = self.st.cpu().numpy(), self.gs.cpu().numpy(), self.counter.cpu().numpy()
        assert self.g.frame_error() == 0
        # exact statuses; a DEGENERATE track (oracle.tri_last_diag, see test_gpu_visual_prepare._well_conditioned) may report another
        # member of the class "triangulation failed"
        same = (st == exp_st).all(axis=2) | (~self.prob["well"] & (st[..., 0] > 0) & (exp_st[..., 0] > 0))
        bad = np.argwhere(~same | (gs != exp_gs))
        assert len(bad) == 0, (tag, bad[:8].tolist(), st[tuple(bad[0])].tolist(), exp_st[tuple(bad[0])].tolist())
        assert int((~(st == exp_st).all(axis=2)).sum()) <= 2, tag
        assert np.array_equal(cnt, exp_cnt), tag
        worst = (0.0, 0.0)
        for b in range(B):
            mg, Pg = self.g.get_state(b)
            em, eP = _rel(mg, exp_m[b]), _rel(Pg, exp_P[b])
            assert em < 1e-8 and eP < 1e-7, (tag, b, em, eP)
            worst = (max(worst[0], em), max(worst[1], eP))
        return worst


@pytest.mark.parametrize("mode", ["one_after_the_other", "concurrent_eager", "concurrent_graph_replay"])
def test_two_lanes_of_300_distinct_ragged_filters(problems, mode):
    import torch
    with capi.Lanes(2, width=64, height=64) as lanes:
        assert len(lanes.ctx) == 2 and lanes.ctx[0].get_stream() != lanes.ctx[1].get_stream() != 0
        with pytest.raises(capi.HvError, match="invalid"):      # a lane keeps the library's stream
            lanes.ctx[0].set_stream(torch.cuda.current_stream().cuda_stream)
        L = [_Lane(lanes.ctx[i], problems[i]) for i in range(2)]
        for l in L:
            l.ctx.set_knob("ekf_visit_order", 1)    # (default: the per-frame sort + second-stream schedule, B = 300 > 256 CUs)
            l.reset()
        torch.cuda.synchronize()
        if mode == "one_after_the_other":
            for l in L:
                l.frame()
                torch.cuda.synchronize()
        elif mode == "concurrent_eager":
            for rep in range(2):                    # twice: the second pass runs with every work buffer allocated (no hidden syncs)
                for l in L:
                    l.reset()
                torch.cuda.synchronize()
                L[0].frame(); L[1].frame()          # both frame loops (~60 launches each) in flight together
                torch.cuda.synchronize()
        else:
            for l in L:                             # first use allocates the library's work buffers: not inside a capture
                l.frame()
            torch.cuda.synchronize()
            graphs = []
            for l in L:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=l.stream):
                    l.frame()
                graphs.append(g)
            for rep in range(3):                    # replay both graphs beside each other, three frames from the same start state
                for l in L:
                    l.reset()
                torch.cuda.synchronize()
                for l, g in zip(L, graphs):
                    with torch.cuda.stream(l.stream):
                        g.replay()
                torch.cuda.synchronize()
            del graphs
        for i, l in enumerate(L):
            l.check(f"{mode} lane {i}")
        for l in L:
            l.g.close()


@pytest.fixture(scope="module")
def problems4(oracle):
    return [_problem(oracle, 9100 + i, B=64) for i in range(4)]


@pytest.mark.parametrize("mode", ["concurrent_eager", "concurrent_graph_replay"])
def test_four_lanes_of_64_distinct_ragged_filters(problems4, mode):
    """The benchmark's default lane count: four lanes in flight at once (eager: each forks onto its own second stream; captured: one
    stream per lane), the sorted two-class schedule forced at 64 filters per lane (knob ekf_visit_order 2), every filter vs the oracle."""
    import torch
    with capi.Lanes(4, width=64, height=64) as lanes:
        assert len({c.get_stream() for c in lanes.ctx}) == 4
        L = [_Lane(lanes.ctx[i], problems4[i]) for i in range(4)]
        for l in L:
            l.ctx.set_knob("ekf_visit_order", 2)
            assert l.ctx.get_knob("ekf_side_stream") == 5
            l.reset()
        torch.cuda.synchronize()
        if mode == "concurrent_eager":
            for rep in range(2):
                for l in L:
                    l.reset()
                torch.cuda.synchronize()
                for l in L:
                    l.frame()
                torch.cuda.synchronize()
        else:
            for l in L:
                l.frame()
            torch.cuda.synchronize()
            graphs = []
            for l in L:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=l.stream):
                    l.frame()
                graphs.append(g)
            for rep in range(3):
                for l in L:
                    l.reset()
                torch.cuda.synchronize()
                for l, g in zip(L, graphs):
                    with torch.cuda.stream(l.stream):
                        g.replay()
                torch.cuda.synchronize()
            del graphs
        for i, l in enumerate(L):
            l.check(f"{mode} lane {i} of 4")
        for l in L:
            l.g.close()


def test_lanes_driven_from_their_own_host_threads(problems4):
    """one driver thread per lane. Two host threads enqueue whole frames on their lanes at the same time
    (ctypes releases the GIL inside every C-ABI call); nothing but the device is shared between lanes, so every filter must still equal
    the oracle's sequential loop."""
    import threading
    import torch
    with capi.Lanes(2, width=64, height=64) as lanes:
        L = [_Lane(lanes.ctx[i], problems4[i]) for i in range(2)]
        for l in L:
            l.ctx.set_knob("ekf_visit_order", 2)
        errors = []

        def drive(l):
            try:
                torch.cuda.set_device(0)
                for rep in range(3):
                    with torch.cuda.stream(l.stream):          # the lane's own stream: resets and frames stay ordered without a device sync
                        l.reset()
                        l.frame()
                l.ctx.synchronize()
            except Exception as ex:                              # pragma: no cover
                errors.append(repr(ex))
        torch.cuda.synchronize()
        threads = [threading.Thread(target=drive, args=(l,)) for l in L]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        assert not errors, errors
        for i, l in enumerate(L):
            l.check(f"threaded lane {i}")
        for l in L:
            l.g.close()


def test_two_lanes_run_the_batch_loop_and_the_hybrid_visit_beside_each_other(oracle):
    """r05: the r04 entries under lanes. Lane 0 runs hv_ekf_visual_frame_batch_dev over 130 distinct filters
    (84-row tracks: the long build + dense batched updates) while lane 1 runs it over 300 filters of short tracks with carried blocks,
    both enqueued before either is waited for; every filter of both lanes against the oracle's batchVisualUpdate loop. Then both lanes
    run a hybrid-map visit (hv_ekf_visual_track_hybrid_dev) concurrently and must leave what each leaves alone."""
    import torch
    from test_gpu_visual_prepare import _batch_loop_case, _run_batch_loop
    cases = [_batch_loop_case(oracle, 7101, 130, 21, 6, 3, 160, True), _batch_loop_case(oracle, 7102, 300, 10, 6, 4, 64, True)]
    with capi.Lanes(2, width=64, height=64) as lanes:
        gs_, outs = [], []
        for ctx, case in zip(lanes.ctx, cases):
            g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=case["trail_len"]), case["B"])
            gs_.append(g); outs.append(_run_batch_loop(case, ctx, g))
        for rep in range(2):                                    # the second round runs with every work buffer allocated
            for g, case, (d, st, gsx, counter) in zip(gs_, cases, outs):
                for b in range(case["B"]):
                    g.set_state(b, case["means"][b], case["P0"][b])
                st.fill_(-9); gsx.fill_(-9); counter.fill_(77)
            torch.cuda.synchronize()
            for g, case, (d, st, gsx, counter) in zip(gs_, cases, outs):     # asynchronous on the lane's own stream
                g.visual_frame_batch_dev(case["vp"], case["K"], case["np_max"], d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(),
                                         d[4].data_ptr(), case["r_gate"], case["r_update"], st.data_ptr(), gsx.data_ptr(), counter.data_ptr(),
                                         case["quota"], case["max_rows"])
            torch.cuda.synchronize()
        for g, case, (d, st, gsx, counter) in zip(gs_, cases, outs):
            case["verify"](g, st.cpu().numpy(), gsx.cpu().numpy(), counter.cpu().numpy())
            g.close()


def test_hybrid_visit_on_two_lanes_equals_the_visit_alone(oracle):
    import torch
    from test_gpu_visual_prepare import _random_tracks
    trail_len, M, npose, B = 20, 3, 8, 48
    n = 20 + 7 * trail_len + 3 * M
    inputs = []
    for seed in (8101, 8102):
        rng = np.random.default_rng(seed)
        T1, T2, means, idx, feat, vel = _random_tracks(oracle, rng, B, trail_len, npose, True, bad_fraction=0.1)
        m = np.concatenate([means, rng.normal(size=(B, 3 * M))], axis=1)
        P = np.zeros((B, n, n))
        for b in range(B):
            A = rng.normal(size=(n, n)) * 1e-3
            P[b] = np.eye(n) * 1e-4 + A @ A.T
        ys = feat.reshape(B, -1) + 2e-3 * rng.normal(size=(B, feat.shape[1] * 2))
        ys[::5] += 3.0
        offer = np.where(np.arange(B) % 3 == 1, np.arange(B) % M, -1).astype(np.int32)
        inputs.append(dict(T1=T1, T2=T2, m=m, P=P, idx=idx, feat=feat, vel=vel, ys=ys, offer=offer))

    def visit(ctx, inp, sync):
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len, hybridMapSize=M), B)
        for b in range(B):
            g.set_state(b, inp["m"][b], inp["P"][b])
        dev = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C")).cuda()
        d = [dev(inp["idx"], np.int32), dev(inp["feat"], np.float64), dev(inp["vel"], np.float64), dev(inp["ys"], np.float64),
             dev(np.full(B, -1, np.int32), np.int32), dev(inp["offer"], np.int32)]
        st = torch.full((B, 2), -9, dtype=torch.int32, device="cuda"); gs = torch.full((B,), -9, dtype=torch.int32, device="cuda")
        chi = torch.zeros(B, dtype=torch.float64, device="cuda"); pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        vp = capi.vu_default_params(imu_to_camera=inp["T1"], second_imu_to_camera=inp["T2"])
        torch.cuda.synchronize()

        def launch():
            g.visual_track_hybrid_dev(vp, npose, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(),
                                      1.5, 0.05, st.data_ptr(), gs.data_ptr(), chi.data_ptr(), pf.data_ptr())

        def result():
            out = (st.cpu().numpy().copy(), gs.cpu().numpy().copy(), [g.get_state(b) for b in range(B)])
            g.close()
            return out
        if sync:
            launch(); torch.cuda.synchronize()
            return result()
        return launch, result, (d, st, gs, chi, pf)

    alone = []
    for inp in inputs:
        with capi.Context(width=64, height=64) as ctx:
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            alone.append(visit(ctx, inp, True))
    with capi.Lanes(2, width=64, height=64) as lanes:
        pend = [visit(ctx, inp, False) for ctx, inp in zip(lanes.ctx, inputs)]
        for launch, _, _ in pend:
            launch()                                            # both visits in flight on the two lanes' streams
        torch.cuda.synchronize()
        for (launch, result, keep), ref in zip(pend, alone):
            st, gs, states = result()
            assert np.array_equal(st, ref[0]) and np.array_equal(gs, ref[1])
            assert (gs == 0).sum() >= 8 and (gs == 3).sum() >= 3
            for (mg, Pg), (mr, Pr) in zip(states, ref[2]):
                assert np.array_equal(mg, mr) and np.array_equal(Pg, Pr)
