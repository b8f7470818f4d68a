#!/usr/bin/env python3
"""How sensitive are the GFTT key points to the binary32 evaluation order OpenCV leaves open?

cv::cornerMinEigenVal is float arithmetic whose last bits depend on the build: the un-normalised box filter of a CV_32F image
accumulates in double (cv::createBoxFilter picks sumType CV_64F for float sources) where the oracle / kernel add three floats in a
fixed order; the SIMD Sobel column filter may contract k0*x0 + k1*(x-1 + x+1) into FMAs. The detector's output is an arg-max per
32x32 block, so it only changes where two responses tie to ~1e-7. This script evaluates the response in the plausible orders in numpy
and pushes each through the oracle's own CollectMax / sort / applyMinDistance, over frames of the BASELINE sizes.
CPU only. Writes profiles/r02/gftt_order_study.json.   python scripts/gftt_order_study.py [--quick]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybvio_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

f32, f64 = np.float32, np.float64


def response(img, box_in_double=False, fma_sobel=False, block=3):
    h, w = img.shape
    g = np.pad(img.astype(f32), 1, mode="reflect")
    scale = 1.0 / (4.0 * block * 255.0)
    k0, k1 = f32(2.0 * scale), f32(1.0 * scale)
    c = lambda dy, dx: g[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]

    def tap3(x0, xm, xp):                       # k0*x0 + k1*(xm + xp)
        if fma_sobel:                           # fma(k1, xm + xp, k0*x0) with the product k0*x0 rounded: one rounding less
            return (f64(k1) * (xm + xp).astype(f64) + (k0 * x0).astype(f64)).astype(f32)
        return k0 * x0 + k1 * (xm + xp)
    dt, dm, db = c(-1, 1) - c(-1, -1), c(0, 1) - c(0, -1), c(1, 1) - c(1, -1)
    vx = tap3(dm, dt, db)
    st, sb = tap3(c(-1, 0), c(-1, -1), c(-1, 1)), tap3(c(1, 0), c(1, -1), c(1, 1))
    vy = sb - st
    out, hb = [], block // 2
    for cov in (vx * vx, vx * vy, vy * vy):
        if box_in_double:                       # ColumnSum<double, float>: exact-ish sum, rounded to float once
            p = np.pad(cov.astype(f64), hb, mode="reflect")
            s = sum(p[j:j + h, i:i + w] for j in range(block) for i in range(block))
            out.append(s.astype(f32))
        else:
            p = np.pad(cov, hb, mode="reflect")
            rows = p[hb:hb + h, 0:w].copy()
            for i in range(1, block):
                rows = rows + p[hb:hb + h, i:i + w]
            p2 = np.pad(rows, ((hb, hb), (0, 0)), mode="reflect")
            s = p2[0:h].copy()
            for j in range(1, block):
                s = s + p2[j:j + h]
            out.append(s)
    a, b, cc = out[0] * f32(0.5), out[1], out[2] * f32(0.5)
    amc = a - cc
    return ((a + cc) - np.sqrt(amc * amc + b * b)).astype(f32)


def corners_from(resp, bs, prev, max_tracks):
    kp = orc.gftt_collect_max(resp, bs, 1e-3)
    order = np.argsort(-kp[:, 2], kind="stable")
    pts = kp[order][:, :2]
    pts = pts[kp[order][:, 2] > -1e9]
    return kp, orc.apply_min_distance(pts, prev, 30, max_tracks)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    cfgs = [(752, 480, 200, 60), (1280, 720, 400, 20)]
    if a.quick:
        cfgs = [(752, 480, 200, 6), (1280, 720, 400, 2)]
    variants = {"box_in_double": dict(box_in_double=True), "fma_sobel": dict(fma_sobel=True), "both": dict(box_in_double=True, fma_sobel=True)}
    out = {"reference_order": "oracle/gftt_oracle.c (= numpy float32, rows then columns, no FMA)", "configs": {}}
    for (w, h, mt, n) in cfgs:
        acc = {k: dict(blocks=0, keypoint_moved=0, response_rel_max=0.0, corner_lists_differ=0, corners_moved=0, frames=0) for k in variants}
        for s in range(n):
            img = synth.render(synth.Texture.make(500 + s), w, h, synth.Warp.make(0.3 * s, 1.0 * s, -0.7 * s, w / 2, h / 2),
                               noise_seed=s, noise_sigma=1.5)
            base = orc.corner_min_eigen_val(img)
            assert np.array_equal(base, response(img))                       # the numpy restatement IS the oracle order
            prev = np.zeros((0, 2), np.float32)
            kp0, c0 = corners_from(base, 32, prev, mt)
            for k, kw in variants.items():
                r = response(img, **kw)
                kp, c = corners_from(r, 32, prev, mt)
                m = acc[k]
                m["frames"] += 1; m["blocks"] += len(kp)
                m["keypoint_moved"] += int((kp[:, :2] != kp0[:, :2]).any(1).sum())
                nz = np.abs(base) > 1e-6
                m["response_rel_max"] = max(m["response_rel_max"], float((np.abs(r - base)[nz] / np.abs(base)[nz]).max()))
                same = len(c) == len(c0) and np.array_equal(c, c0)
                m["corner_lists_differ"] += int(not same)
                if not same:
                    m["corners_moved"] += int(len(set(map(tuple, c.tolist())) ^ set(map(tuple, c0.tolist()))) // 2)
        out["configs"][f"{w}x{h}"] = acc
    if not a.quick:
        with open(os.path.join(ROOT, "profiles", "r02", "gftt_order_study.json"), "w") as f:
            json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
