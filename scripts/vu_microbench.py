"""Times vu_prepare_kernel (row f3) alone and the fused prepare + gate + update call: B filters, one track each.
usage: python scripts/vu_microbench.py [B] [n_poses] [stereo 0|1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybvio_amd import capi  # noqa: E402


from hybvio_amd.synth import visual_tracks as synthetic_tracks  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    npose = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    stereo = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
    trail_len = 20
    rng = np.random.default_rng(0)
    T1, T2, means, idx, feat = synthetic_tracks(rng, B, trail_len, npose, stereo)
    vel = rng.normal(size=feat.shape) * 0.1
    y = feat.reshape(B, -1) + 1e-3 * rng.normal(size=(B, feat.shape[1] * 2))
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2) if stereo else capi.vu_default_params(imu_to_camera=T1)
    with capi.Context(width=64, height=64) as ctx:
        g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=trail_len), B)
        P = None
        for b in range(B):
            if P is None:
                _, P = g.get_state(0)
                P = P * 1e-6 + np.eye(g.n) * 1e-4
            g.set_state(b, means[b], P)
        dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
        d_idx, d_feat, d_vel, d_y = dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64), dev(y, np.float64)
        rows = 2 * npose * (2 if stereo else 1)
        H = torch.zeros((B, g.n, rows), dtype=torch.float64, device="cuda")
        v = torch.zeros((B, rows), dtype=torch.float64, device="cuda"); pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
        st = torch.zeros((B, 2), dtype=torch.int32, device="cuda"); gs = torch.zeros((B,), dtype=torch.int32, device="cuda")
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        prep = lambda: g.visual_prepare_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), d_y.data_ptr(), H.data_ptr(),
                                            v.data_ptr(), 0, pf.data_ptr(), st.data_ptr(), 0)
        for _ in range(3):
            prep()
        ctx.profile_enable(True); ctx.profile_reset()
        for _ in range(20):
            prep()
        ms, cnt = ctx.profile_read(capi.K_VU_PREPARE)
        ctx.profile_enable(False)
        ok = int((st.cpu().numpy() == 0).all(1).sum())
        print(f"vu_prepare_kernel: B={B} poses={npose} stereo={stereo} rows={rows}: {ms / cnt * 1e3:.1f} us per launch "
              f"({ms / cnt / B * 1e6:.1f} ns per track at this batch), {ok}/{B} tracks OK")
        if os.environ.get("HV_EKF_PHASE_STAMPS") == "1":
            import ctypes as C
            st32 = (C.c_longlong * 40)()
            capi.lib().hv_debug_vu_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
            capi.lib().hv_debug_vu_phase_stamps(ctx._h, st32)
            s_ = list(st32)
            print("phase ticks (100 MHz s_memtime... or cycles): setup", s_[1] - s_[0], "two-cam", s_[2] - s_[1],
                  "iters [pose, (wait), columns, update]:", [[s_[4 + 4 * k] - s_[3 + 4 * k], s_[5 + 4 * k] - s_[4 + 4 * k], s_[6 + 4 * k] - s_[5 + 4 * k],
                                                             (s_[7 + 4 * k] if k < 5 else s_[27]) - s_[6 + 4 * k]] for k in range(6) if s_[6 + 4 * k] > s_[3 + 4 * k] > 0],
                  "final", s_[28] - s_[27], "prep-pose", s_[29] - s_[28], "H", s_[30] - s_[29], "total", s_[30] - s_[0])
        # the fused prepare + column-sparse gate kernel and the update of the inliers (r03 default); y_out rejects every track, so the
        # update launch is a skip launch and the vu_prepare timer class shows the fused kernel alone
        frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25
        y_mix = y.copy(); y_mix[rng.uniform(size=B) >= frac] += 3.0
        for name, yy in (("all rejected", dev(y + 3.0, np.float64)), ("all inliers", d_y), (f"{frac:.2f} inliers", dev(y_mix, np.float64))):
            for _ in range(3):
                g.visual_track_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), yy.data_ptr(), 1.5, 0.05,
                                   st.data_ptr(), gs.data_ptr(), 0, pf.data_ptr())
            ctx.profile_enable(True); ctx.profile_reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                g.visual_track_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), yy.data_ptr(), 1.5, 0.05,
                                   st.data_ptr(), gs.data_ptr(), 0, pf.data_ptr())
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 20 * 1e6
            ms1, c1 = ctx.profile_read(capi.K_VU_PREPARE); ms2, c2 = ctx.profile_read(capi.K_EKF_UPDATE); ms3, c3 = ctx.profile_read(capi.K_EKF_GATE)
            ms4, c4 = ctx.profile_read(capi.K_VU_TRI)
            ctx.profile_enable(False)
            print(f"visual_track_dev ({name}): {wall:.1f} us per call; triangulation front {ms4 / max(c4, 1) * 1e3:.1f} us, prepare(+gate) kernel {ms1 / c1 * 1e3:.1f} us, sparse gate kernel {ms3 / max(c3, 1) * 1e3:.1f} us, update launch {ms2 / c2 * 1e3:.1f} us; "
                  f"gate inliers {int((gs.cpu().numpy() == 0).sum())}/{B}")
            if os.environ.get("HV_EKF_PHASE_STAMPS") == "1" and name == "all rejected" and ms3 > 0:
                import ctypes as C
                st16 = (C.c_longlong * 16)()
                capi.lib().hv_debug_ekf_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
                capi.lib().hv_debug_ekf_phase_stamps(g._h, st16)
                q_ = list(st16)
                print("big sparse gate phase cycles: staging", q_[13] - q_[12], "gather + products", q_[0] - q_[13], "Cholesky", q_[1] - q_[0], "chi2", q_[2] - q_[1])
            if os.environ.get("HV_EKF_PHASE_STAMPS") == "1" and name == "all inliers":
                import ctypes as C
                st16 = (C.c_longlong * 16)()
                capi.lib().hv_debug_ekf_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
                capi.lib().hv_debug_ekf_phase_stamps(g._h, st16)
                q_ = list(st16)
                print("update kernel phase cycles (workgroup 0): P loads issued + H staged", q_[9] - q_[0], "(of which up to the H requests", q_[8] - q_[0], ")", "item 0 products", q_[10] - q_[9], "item 1", q_[11] - q_[10], "H P products", q_[1] - q_[9], "S", q_[2] - q_[1],
                      "Cholesky + solves", q_[3] - q_[2], "(first diagonal block", q_[7] - q_[6], ") mean step", q_[4] - q_[3], "P -= Y'Y + store", q_[5] - q_[4],
                      "total", q_[5] - q_[0])
            if os.environ.get("HV_EKF_PHASE_STAMPS") == "1" and name == "all rejected":
                import ctypes as C
                st40 = (C.c_longlong * 40)()
                capi.lib().hv_debug_vu_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
                capi.lib().hv_debug_vu_phase_stamps(ctx._h, st40)
                s_ = list(st40)
                t64 = (C.c_longlong * 64)()
                capi.lib().hv_debug_tri_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
                capi.lib().hv_debug_tri_phase_stamps(ctx._h, t64)
                t_ = list(t64)
                if t_[57] > t_[0] > 0:
                    its = [[t_[4 + 10 * k + j + 1] - t_[4 + 10 * k + j] for j in range(6)] for k in range(5) if t_[4 + 10 * k + 6] > t_[4 + 10 * k] > t_[0]]
                    print("vu_tri_kernel phase cycles (workgroup 0): load", t_[1] - t_[0], "trail", t_[2] - t_[1], "two-cam", t_[3] - t_[2],
                          "iterations [pose, ETE+plain, L+X, pairs C/P/Q, pose-0 totals, pose-0 columns + step]:", its,
                          "status", t_[55] - t_[54], "prepare-pose + world + record", t_[56] - t_[55], "final", t_[57] - t_[56], "total", t_[57] - t_[0])
                if os.environ.get("HV_RAW_STAMPS") == "1":
                    print("structured_S stamps (deltas of g_vu_stamp[1..11]):", [s_[k + 1] - s_[k] for k in range(1, 11)])
                print("fused kernel phase cycles: up to prepare-pose", s_[29] - s_[0], "compact H + v", s_[31] - s_[29], "zero T", s_[32] - s_[31],
                      "gather + products", s_[33] - s_[32], "Cholesky", s_[34] - s_[33], "chi2", s_[35] - s_[34], "total", s_[35] - s_[0])
        g.close()


if __name__ == "__main__":
    main()
