#!/usr/bin/env python3
"""One track visit (hv_ekf_visual_track_dev: triangulation + prepareVisualUpdate + chi2 gate + update) over the track lengths a stereo
session sees (4 .. 21 poses = 16 .. 84 rows), B filters: the default path (r04: the long class's prepare + gate in one
launch) against r03's two-launch long class (knob ekf_long_fused = 0) and r02's dense kernels (knob ekf_fused_gate = 0).
Prints one JSON object: {poses: {path: {case: {wall_us, prepare_us, gate_us, update_us}}}}. Run on the GPU box.
usage: python scripts/track_length_sweep.py [B]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybvio_amd import capi  # noqa: E402
from hybvio_amd.synth import visual_tracks  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = {"filters": B, "unit": "us per call of hv_ekf_visual_track_dev (kernel classes: hipEvents on the context stream)", "poses": {}}
for npose in (4, 8, 10, 11, 12, 13, 16, 20, 21):
    rng = np.random.default_rng(npose)
    T1, T2, means, idx, feat = visual_tracks(rng, B, 20, npose, True, noise=1e-4, recent=True)
    vel = rng.normal(size=feat.shape) * 0.1
    y_in = feat.reshape(B, -1) + 1e-4 * rng.normal(size=(B, feat.shape[1] * 2))
    vp = capi.vu_default_params(imu_to_camera=T1, second_imu_to_camera=T2)
    row = {"rows": 4 * npose, "active_columns": 7 * npose + 1}
    for path, knobs in (("r04", {}), ("r03_long_two_launches", {"ekf_long_fused": 0}), ("r02_dense", {"ekf_fused_gate": 0})):
        with capi.Context(width=64, height=64) as ctx:
            for k, v in knobs.items():
                ctx.set_knob(k, v)
            g = capi.EkfBatch(ctx, capi.ekf_default_params(cameraTrailLength=20), B)
            _, P = g.get_state(0)
            P = P * 1e-6 + np.eye(g.n) * 1e-4
            for b in range(B):
                g.set_state(b, means[b], P)
            dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
            d_idx, d_feat, d_vel = dev(idx, np.int32), dev(feat, np.float64), dev(vel, np.float64)
            st = torch.zeros((B, 2), dtype=torch.int32, device="cuda"); gs = torch.zeros((B,), dtype=torch.int32, device="cuda")
            pf = torch.zeros((B, 3), dtype=torch.float64, device="cuda")
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            res = {}
            for case, yy in (("all_rejected", dev(y_in + 0.05, np.float64)), ("all_inliers", dev(y_in, np.float64))):
                call = lambda: g.visual_track_dev(vp, npose, d_idx.data_ptr(), d_feat.data_ptr(), d_vel.data_ptr(), yy.data_ptr(),
                                                  1.5 / 458.654, 0.05 / 458.654, st.data_ptr(), gs.data_ptr(), 0, pf.data_ptr())
                for _ in range(3):
                    call()
                ctx.profile_enable(True); ctx.profile_reset()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(10):
                    call()
                torch.cuda.synchronize()
                wall = (time.perf_counter() - t0) / 10 * 1e6
                t = {name: ctx.profile_read(kid) for name, kid in (("prepare_us", capi.K_VU_PREPARE), ("gate_us", capi.K_EKF_GATE), ("update_us", capi.K_EKF_UPDATE))}
                ctx.profile_enable(False)
                res[case] = {"wall_us": round(wall, 1), **{k: round(ms / 10 * 1e3, 1) for k, (ms, n_) in t.items()},
                             "gate_inliers": int((gs == 0).sum().item())}
            row[path] = res
            g.close()
    out["poses"][npose] = row
print(json.dumps(out))
