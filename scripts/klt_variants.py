#!/usr/bin/env python3
"""A/B of the LK kernel's experiment knobs on the bench workload (C2, B sequences): HV_PAD_FROM_LEVEL (first pyramid level with a
physical border; 9 = none) x HV_KLT_TILE (staged J tile shape). Both are read when the library starts, so every combination
runs in its own process. Each run also checks LK statuses / positions of two sequences against the CPU oracle (bit-exact).
    python scripts/klt_variants.py [--sequences 1024] [--steps 10] > gpurun_out/klt_variants.json
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(B, steps):
    import numpy as np
    import torch
    import bench
    from hybvio_amd import capi
    from oracle import orc
    tb = bench.TrackerBench(B, 0, seed=0)
    for _ in range(3):
        tb.step()
    torch.cuda.synchronize()
    tb.ctx.profile_enable(True); tb.ctx.profile_reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        tb.step()
    e1.record(); torch.cuda.synchronize()
    out = {"ms_per_step": e0.elapsed_time(e1) / steps}
    for name, kid in (("pyr_l0", capi.K_PYR_L0), ("pyr_ln", capi.K_PYR_LN), ("klt", capi.K_KLT)):
        ms, n = tb.ctx.profile_read(kid)
        out[name] = {"avg_ms": ms / max(n, 1), "per_step_ms": ms / steps}
    tb.ctx.profile_enable(False)
    out["tracked_fraction"] = tb.tracked_fraction()
    # parity of this variant: the last temporal LK call of sequences 0 and B-1 against the oracle
    k = tb.k - 1
    bad = 0
    for s in (0, B - 1):
        prev = tb.frames[(k - 1) % bench.N_CYCLE, 0, s].cpu().numpy()
        cur = tb.frames[k % bench.N_CYCLE, 0, s].cpu().numpy()
        right = tb.frames[k % bench.N_CYCLE, 1, s].cpu().numpy()
        rng = np.random.default_rng(s)
        pts = np.concatenate([rng.uniform([-8, -8], [bench.W + 8, bench.H + 8], (150, 2)),
                              rng.uniform([0, 0], [40, bench.H], (50, 2))]).astype(np.float32)     # borders on purpose
        with capi.Context(width=bench.W, height=bench.H, pool_size=4) as c2:
            a, b, r = c2.acquire(), c2.acquire(), c2.acquire()
            c2.build(a, prev); c2.build(b, cur); c2.build(r, right)
            pa, pb, pr = orc.Pyramid(prev), orc.Pyramid(cur), orc.Pyramid(right)
            for l in range(c2.levels):
                g, d = c2.download(b, l)
                bad += int(not (np.array_equal(g, pb.gray(l)) and np.array_equal(d, pb.deriv(l))))
            for (sa, sb, oa, ob, guess) in ((a, b, pa, pb, None), (b, r, pb, pr, pts - np.float32([20, 0]))):
                xy, st, _ = c2.klt_track(sa, sb, pts, next_xy=guess)
                oxy, ost, _ = orc.klt_track(oa, ob, pts, next_pts=guess)
                bad += int(not np.array_equal(st, ost)) + int(not np.array_equal(xy[st > 0], oxy[ost > 0]))
    out["parity_failures"] = bad
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--pads", default="9,2,1")
    ap.add_argument("--tiles", default="0,1,2,3")
    a = ap.parse_args()
    if a.child:
        return child(a.sequences, a.steps)
    res = {}
    for pad in a.pads.split(","):
        for tile in a.tiles.split(","):
            env = dict(os.environ, HV_PAD_FROM_LEVEL=pad, HV_KLT_TILE=tile)
            p = subprocess.run([sys.executable, __file__, "--child", "--sequences", str(a.sequences), "--steps", str(a.steps)],
                               env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            res[f"pad{pad}_tile{tile}"] = json.loads(line[-1][7:]) if line else {"error": (p.stderr or p.stdout)[-600:]}
            print(f"pad{pad}_tile{tile}", json.dumps(res[f"pad{pad}_tile{tile}"]), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
