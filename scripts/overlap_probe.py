"""Does the HBM-bound pyramid of one half of the batch overlap the VALU-bound LK of the other half when the two halves run on
two streams? usage: python scripts/overlap_probe.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = 20


def run_single():
    tb = bench.TrackerBench(B, 0, seed=0)
    for _ in range(3):
        tb.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        tb.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tb.ctx.close()
    return dt / steps


def run_split(parts):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    tbs = []
    for s in streams:
        with torch.cuda.stream(s):
            tbs.append(bench.TrackerBench(B // parts, 0, seed=len(tbs)))
    def one():
        for s, tb in zip(streams, tbs):
            with torch.cuda.stream(s):
                tb.step()
    for _ in range(3):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for tb in tbs:
        tb.ctx.close()
    return dt / steps


a = run_single()
print(f"one stream,  B={B}: {a * 1e3:.3f} ms per step = {B / a:.0f} frames/s")
for parts in (2, 4):
    b = run_split(parts)
    print(f"{parts} streams x B={B // parts}: {b * 1e3:.3f} ms per step = {B / b:.0f} frames/s ({a / b:.3f}x)")
