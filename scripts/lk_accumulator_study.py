#!/usr/bin/env python3
"""How far is the exact-integer LK accumulation (the kernel's and the oracle's default) from what an x86 OpenCV build
computes?  

OpenCV cannot be built or imported here, so the float-accumulator variants of LKTrackerInvoker are RESTATED in
oracle/pyrlk_oracle.c (ORC_ACC_F32_*: scalar float, the 4-lane universal-intrinsic code of OpenCV >= 4.1 with and without
FMA, the legacy SSE2 code, and an 8-lane sensitivity probe) and every one of them is run against the int64 mode over the
BASELINE corpus: 752x480 / 200 points and 1280x720 / 400 points, temporal and stereo pairs, with and without an initial
guess (OPTFLOW_USE_INITIAL_FLOW). Points are carried from frame to frame by the int64 tracker and re-seeded when lost, so
borders, lost tracks and low-texture cases appear at their natural rate.

CPU only (the oracle); ~5 min on 8 cores for the full corpus.  Writes profiles/r02/lk_accumulator_study.json and prints
the DESIGN.md table.   python scripts/lk_accumulator_study.py [--quick]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hybvio_amd import synth  # noqa: E402
from oracle import orc  # noqa: E402

F32_MODES = ["f32_scalar", "f32_simd128", "f32_simd128_fma", "f32_sse2_legacy", "f32_wide8"]
# the yardstick: two OpenCV float builds against EACH OTHER (the int64 mode cannot be expected to sit closer to either of
# them than they sit to one another)
CROSS = [("f32_simd128", "f32_scalar"), ("f32_simd128", "f32_sse2_legacy")]
KEYS = F32_MODES + [f"{a}_vs_{b}" for a, b in CROSS]


class Acc:
    def __init__(self):
        self.points = 0            # points compared
        self.flips = 0             # cv status differs
        self.fs_flips = 0          # Feature::Status (TRACKED / FAILED_FLOW / FLOW_OUT_OF_RANGE) differs
        self.both = 0              # tracked in both
        self.d = []                # |dxy|_inf on points tracked in both
        self.iter_diff = 0         # points whose per-level iteration counts differ

    def add(self, ref, got, w, h):
        (rxy, rst, _, rit), (gxy, gst, _, git) = ref, got
        self.points += len(rst)
        self.flips += int((rst != gst).sum())

        def fstat(xy, st):
            out = np.where(st == 0, 2, 0)
            oor = (xy[:, 0] < 0) | (xy[:, 0] >= w) | (xy[:, 1] < 0) | (xy[:, 1] >= h)
            return np.where(oor, 4, out)
        self.fs_flips += int((fstat(rxy, rst) != fstat(gxy, gst)).sum())
        ok = (rst > 0) & (gst > 0)
        self.both += int(ok.sum())
        self.d.append(np.abs(rxy - gxy).max(1)[ok])
        self.iter_diff += int((rit != git).any(0).sum())

    def summary(self):
        d = np.concatenate(self.d) if self.d else np.zeros(0)
        q = lambda p: float(np.quantile(d, p)) if len(d) else 0.0
        return {"points": self.points, "status_flips": self.flips, "status_flip_rate": self.flips / max(1, self.points),
                "feature_status_flips": self.fs_flips, "tracked_in_both": self.both,
                "dxy_max": float(d.max()) if len(d) else 0.0, "dxy_p50": q(0.5), "dxy_p99": q(0.99), "dxy_p999": q(0.999),
                "frac_dxy_gt_1e-3": float((d > 1e-3).mean()) if len(d) else 0.0,
                "frac_dxy_gt_1e-2": float((d > 1e-2).mean()) if len(d) else 0.0,
                "points_with_other_iteration_counts": self.iter_diff}


def run_config(w, h, n_pts, n_seq, frames_per_seq, seed0, accs, log):
    pairs = 0
    rng = np.random.default_rng(seed0)
    for s in range(n_seq):
        tex = synth.Texture.make(seed0 + s)
        # faster motion than the bench path: ~6 px and ~0.4 deg per frame, so the coarse levels and the iteration cap matter
        warps = synth.camera_path(frames_per_seq, w, h, radius_px=1.0 * frames_per_seq, rot_amp_deg=1.5)
        disparity = float(rng.uniform(8, 40))
        pts = synth.grid_points(w, h, n_pts, margin=2, seed=seed0 + s)
        prevL = None
        for k in range(frames_per_seq):
            wk = warps[k]
            left = synth.render(tex, w, h, wk, noise_seed=(seed0 + s) * 997 + 2 * k, noise_sigma=1.0)
            wr = synth.Warp(wk.A.copy(), wk.t + wk.A @ np.array([disparity, 0.0]))
            right = synth.render(tex, w, h, wr, noise_seed=(seed0 + s) * 997 + 2 * k + 1, noise_sigma=2.0)
            pL, pR = orc.Pyramid(left), orc.Pyramid(right)
            jobs = []
            if prevL is not None:
                flow = synth.true_flow(pts.astype(np.float64), warps[k - 1], wk).astype(np.float32)
                guess = flow + rng.normal(0, 1.0, flow.shape).astype(np.float32)
                jobs += [(prevL, pL, pts, None), (prevL, pL, pts, guess)]
            cur = pts if prevL is None else None
            # stereo: previous-frame points of the left image tracked into the right one (no guess / disparity guess)
            base = pts
            sg = base + np.array([-disparity, 0.0], np.float32) + rng.normal(0, 1.0, base.shape).astype(np.float32)
            jobs += [(pL, pR, base, None), (pL, pR, base, sg)]
            carried = None
            for (a, b, p, g) in jobs:
                ref = orc.klt_track(a, b, p, next_pts=g, want_iters=True)
                if carried is None and a is prevL:
                    carried = ref
                res = {m: orc.klt_track(a, b, p, next_pts=g, want_iters=True, acc_mode=m) for m in F32_MODES}
                for m in F32_MODES:
                    accs[m].add(ref, res[m], w, h)
                for x, y in CROSS:
                    accs[f"{x}_vs_{y}"].add(res[y], res[x], w, h)
                pairs += 1
            # carry the tracked points forward (int64 mode decides), re-seed the lost ones anywhere in the image
            if carried is not None:
                xy, st = carried[0], carried[1]
                lost = (st == 0) | (xy[:, 0] < 0) | (xy[:, 0] >= w) | (xy[:, 1] < 0) | (xy[:, 1] >= h)
                pts = xy.copy()
                pts[lost] = np.stack([rng.uniform(0, w - 1, int(lost.sum())), rng.uniform(0, h - 1, int(lost.sum()))], 1).astype(np.float32)
            prevL = pL
        log(f"  {w}x{h}: sequence {s + 1}/{n_seq} done, {pairs} pairs")
    return pairs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true", help="a 1/20 sample (what tests/test_oracle_lk_accumulators.py runs)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02", "lk_accumulator_study.json"))
    a = ap.parse_args()
    orc.lib()
    t0 = time.time()
    log = lambda s: print(s, file=sys.stderr, flush=True)
    cfgs = [("752x480_200pts", 752, 480, 200, 42, 13), ("1280x720_400pts", 1280, 720, 400, 14, 13)]
    if a.quick:
        cfgs = [("752x480_200pts", 752, 480, 200, 2, 7), ("1280x720_400pts", 1280, 720, 400, 1, 5)]
    out = {"modes_vs": "int64 (oracle default = HIP kernel)", "configs": {}}
    total = {m: Acc() for m in KEYS}
    for name, w, h, n, nseq, nf in cfgs:
        accs = {m: Acc() for m in KEYS}
        pairs = run_config(w, h, n, nseq, nf, 1000 + w, accs, log)
        out["configs"][name] = {"frame_pairs": pairs, "points_per_pair": n, "modes": {m: accs[m].summary() for m in KEYS}}
        for m in KEYS:
            total[m].points += accs[m].points; total[m].flips += accs[m].flips; total[m].fs_flips += accs[m].fs_flips
            total[m].both += accs[m].both; total[m].d += accs[m].d; total[m].iter_diff += accs[m].iter_diff
    out["all"] = {"frame_pairs": sum(c["frame_pairs"] for c in out["configs"].values()),
                  "modes": {m: total[m].summary() for m in KEYS}}
    out["seconds"] = round(time.time() - t0, 1)
    if not a.quick:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)
    print("| corpus | f32 mode | points | status flips (rate) | Feature::Status flips | max abs dxy (px, tracked in both) | p99.9 | share > 1e-3 px | points with other iteration counts |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, c in list(out["configs"].items()) + [("all", out["all"])]:
        for m, r in c["modes"].items():
            print(f"| {name} ({c['frame_pairs']} pairs) | {m} | {r['points']} | {r['status_flips']} ({r['status_flip_rate']:.2e}) | "
                  f"{r['feature_status_flips']} | {r['dxy_max']:.2e} | {r['dxy_p999']:.2e} | {r['frac_dxy_gt_1e-3']:.2e} | "
                  f"{r['points_with_other_iteration_counts']} |")
    return out


if __name__ == "__main__":
    main()
