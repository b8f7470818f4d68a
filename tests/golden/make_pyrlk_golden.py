"""Writes tests/golden/pyrlk_golden.npz: seeded inputs + the CPU oracle's outputs for them.

The reference has no golden vectors for the pyramid / LK path and OpenCV cannot be imported here
(SURVEY.md section 8c), so this fixture pins the ORACLE (regression guard + GPU-box check without
rebuilding anything); it is not an OpenCV output.  Run from the repo root:
    python tests/golden/make_pyrlk_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hybvio_amd import synth  # noqa: E402
from oracle import orc        # noqa: E402

W, H = 200, 150
tex = synth.Texture.make(1234)
w0 = synth.Warp.make(0.0, 0.0, 0.0, W / 2, H / 2)
w1 = synth.Warp.make(0.4, 1.7, -1.1, W / 2, H / 2)
img0 = synth.render(tex, W, H, w0)
img1 = synth.render(tex, W, H, w1, noise_seed=99, noise_sigma=1.5)
rng = np.random.default_rng(42)
pts = np.concatenate([synth.grid_points(W, H, 40, margin=4, seed=1),
                      rng.uniform([-20, -20], [W + 20, H + 20], (24, 2)).astype(np.float32)])
guess = (pts + rng.normal(0, 1.0, pts.shape)).astype(np.float32)

p0, p1 = orc.Pyramid(img0), orc.Pyramid(img1)
out = dict(img0=img0, img1=img1, pts=pts, guess=guess, levels=p0.levels)
for l in range(p0.levels):
    out[f"gray{l}"] = p0.gray(l)
    out[f"deriv{l}"] = p0.deriv(l)
out["next"], out["status"], out["err"] = orc.klt_track(p0, p1, pts)
out["flow_corners"], out["flow_status"] = orc.optical_flow_compute(p0, p1, pts, corners=guess)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyrlk_golden.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes; levels", p0.levels, "tracked", int(out["status"].sum()), "/", len(pts),
      "flow_status", np.bincount(out["flow_status"], minlength=5))
