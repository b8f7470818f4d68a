"""Writes tests/golden/pyrlk_golden_opencv.npz from a REAL OpenCV (cv2), when one is importable.

Not runnable in the build container (no cv2, no network): see oracle/_ref/README. Same seeded inputs as
make_pyrlk_golden.py; the calls are the reference's (src/tracker/image_pyramid.cpp:42-46, optical_flow.cpp:46-49):
    cv::buildOpticalFlowPyramid(img, pyr, Size(31, 31), 3)          withDerivatives = true, REFLECT_101 / CONSTANT borders
    cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, prev, next, status, err, Size(31, 31), 3,
                             TermCriteria(COUNT | EPS, 20, 0.03), flags, 1e-3)
"""
import os
import sys

import numpy as np

try:
    import cv2
except ImportError:
    sys.exit("cv2 is not importable here: nothing written (oracle/_ref/README explains the situation)")

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
src = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyrlk_golden.npz"))
img0, img1, pts, guess = src["img0"], src["img1"], src["pts"], src["guess"]
WIN, MAXL = (31, 31), 3
crit = (cv2.TERM_CRITERIA_COUNT | cv2.TERM_CRITERIA_EPS, 20, 0.03)

n0, pyr0 = cv2.buildOpticalFlowPyramid(img0, WIN, MAXL, withDerivatives=True)
n1, pyr1 = cv2.buildOpticalFlowPyramid(img1, WIN, MAXL, withDerivatives=True)
out = dict(cv_version=cv2.__version__, cv_build=cv2.getBuildInformation(), levels=n0 + 1)
for l in range(n0 + 1):
    g, d = pyr0[2 * l], pyr0[2 * l + 1]                      # ROIs of the padded buffers: interior only
    out[f"gray{l}"] = np.ascontiguousarray(g)
    out[f"deriv{l}"] = np.ascontiguousarray(d)
nxt, st, err = cv2.calcOpticalFlowPyrLK(pyr0, pyr1, pts.reshape(-1, 1, 2), None, winSize=WIN, maxLevel=MAXL, criteria=crit,
                                        flags=0, minEigThreshold=1e-3)
out["next"], out["status"], out["err"] = nxt.reshape(-1, 2), st.reshape(-1), err.reshape(-1)
nxt2, st2, _ = cv2.calcOpticalFlowPyrLK(pyr0, pyr1, pts.reshape(-1, 1, 2), guess.reshape(-1, 1, 2).copy(), winSize=WIN,
                                        maxLevel=MAXL, criteria=crit, flags=cv2.OPTFLOW_USE_INITIAL_FLOW, minEigThreshold=1e-3)
out["next_guess"], out["status_guess"] = nxt2.reshape(-1, 2), st2.reshape(-1)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pyrlk_golden_opencv.npz")
np.savez_compressed(path, **out)
print(path, "written with OpenCV", cv2.__version__)
