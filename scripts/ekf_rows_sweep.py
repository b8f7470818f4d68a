#!/usr/bin/env python3
"""Time the batched EKF visual gate / update over the row counts a stereo track can have (16 .. 84 rows):
hv_ekf_visual_dev mode 0 (visualTrackOutlierCheck), mode 1 (updateVisualTrack) and mode 3 (gate + update) at B filters, random
dense H. Prints one JSON object: {rows: {mode: us_per_launch}}. Run on the GPU box."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybvio_amd import capi  # noqa: E402

B = int(os.environ.get("SWEEP_B", "256"))
dev = torch.device("cuda", 0)
ctx = capi.Context(width=752, height=480)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ekf = capi.EkfBatch(ctx, capi.ekf_default_params(), B)
rng = np.random.default_rng(0)
dt = torch.full((10, B), 0.005, dtype=torch.float64, device=dev)
gyro = torch.from_numpy(rng.normal(0, 0.05, (10, B, 3))).to(dev)
acc = torch.from_numpy(rng.normal(0, 0.05, (10, B, 3)) + [0.0, 0.0, 9.819]).to(dev)
for _ in range(24):
    ekf.predict_n_dev(10, dt.data_ptr(), gyro.data_ptr(), acc.data_ptr())
    ekf._chk(capi.lib().hv_ekf_augment(ekf._h, None, None), "hv_ekf_augment")


class _DevView:                                     # zero-copy torch view of the library's device buffers
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (ptr, False), "version": 2}


mp, pp = ekf.device_pointers()
n = ekf.n
m_view, P_view = torch.as_tensor(_DevView(mp, (B, n)), device=dev), torch.as_tensor(_DevView(pp, (B, n, n)), device=dev)
m0, P0 = m_view.clone(), P_view.clone()
out = {}
chi2 = torch.zeros(B, dtype=torch.float64, device=dev)
status = torch.zeros(B, dtype=torch.int32, device=dev)
for nr in (16, 40, 48, 49, 64, 76, 80, 84):
    H = torch.from_numpy(rng.normal(size=(B, 160, nr))).to(dev)
    v = torch.from_numpy(0.02 * rng.normal(size=(B, nr))).to(dev)
    row = {}
    for mode, name in ((0, "gate"), (1, "update"), (2, "gate+update")):
        ts = []
        for rep in range(6):
            m_view.copy_(m0); P_view.copy_(P0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ekf.visual_dev(nr, 160, H.data_ptr(), v.data_ptr(), 0.05, mode, chi2.data_ptr(), status.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        row[name] = round(float(np.median(ts[1:])), 1)
    row["inliers"] = int((status == 0).sum().item())
    out[nr] = row
print(json.dumps({"filters": B, "us_per_launch": out}))
