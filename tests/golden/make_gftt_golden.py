"""Writes tests/golden/gftt_golden.npz: a seeded image + the CPU oracle's detector outputs for it.

The reference has no golden vectors for the feature detector and OpenCV cannot be imported here, so this
fixture pins the ORACLE (regression guard, and a GPU-box check that needs no oracle build); it is not an
OpenCV output.  Run from the repo root:
    python tests/golden/make_gftt_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hybvio_amd import synth  # noqa: E402
from oracle import orc        # noqa: E402

W, H = 200, 150
img = synth.render(synth.Texture.make(77), W, H, synth.Warp.make(0.0, 0.0, 0.0, W / 2, H / 2), noise_seed=3, noise_sigma=1.0)
rng = np.random.default_rng(5)
prev = rng.uniform([0, 0], [W, H], (12, 2)).astype(np.float32)
out = dict(img=img, prev=prev)
resp = orc.corner_min_eigen_val(img)
out["response_checksum"] = np.array([np.float64(resp.astype(np.float64).sum()), np.float64(np.abs(resp).astype(np.float64).max())])
for bs, md in ((8, 8.0), (16, 20.0), (32, 50.0)):
    out[f"kp{bs}"] = orc.gftt_collect_max(resp, bs, 1e-3)
    out[f"raw{bs}"] = orc.gftt_detect(img, mask_radius=0, min_distance=md)
    out[f"masked{bs}"] = orc.gftt_detect(img, prev=prev, mask_radius=int(md), min_distance=md, max_tracks=30)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gftt_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items() if k != "img"})
