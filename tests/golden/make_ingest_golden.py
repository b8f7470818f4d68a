"""Writes tests/golden/ingest_golden.npz: one small colour frame, its gray conversion, the remap table of a radially
distorted pinhole camera against the mono rectified camera, and the rectified gray image -- all produced by
oracle/ingest_oracle.c (the reference cannot be built here and holds no vectors for this stage: SURVEY.md 8(c)).
The fixture freezes the oracle and gives the GPU tests an input/output pair that does not depend on it at run time.
Run from the repository root:  python tests/golden/make_ingest_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hybvio_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

w, h = 188, 120
intr = np.array([114.66, 114.32, 91.8, 62.1])
coeffs = np.array([-0.28340811, 0.07395907, 0.0])
left = synth.stereo_sequence(5, w, h, 1)[0][0]
rng = np.random.default_rng(1)
rgb = np.clip(left[..., None].astype(np.int32) + rng.integers(-40, 41, (h, w, 3)), 0, 255).astype(np.uint8)
gray = orc.color_to_gray(rgb)
rect = orc.mono_rectified_camera(w, h, 100.0, zoom=0.9)
cam = orc.Camera("pinhole", *intr, coeffs=coeffs)
pix, valid = orc.undistort_map(rect, cam, w, h)
out = orc.undistort_apply(gray, pix, valid)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ingest_golden.npz"), w=w, h=h, intrinsics=intr, coeffs=coeffs,
                    rect_focal=100.0, zoom=0.9, rgb=rgb, gray=gray, pix=pix, valid=valid, rectified=out)
print("wrote ingest_golden.npz", out.mean(), (out == 0).mean())
