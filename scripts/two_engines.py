#!/usr/bin/env python3
"""Experiment: the realistic C3 step of bench.py run by E independent engine instances (each its own hv context, stream and HIP graphs)
on ONE GPU, S sequences each -- do the latency-bound launch chains of several engines fill each other's idle CUs?
usage: two_engines.py E S [steps]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench

E, S = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
OFFSET = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # extra eager steps of engine 0 before the capture (frame-cycle offset between the engines)
SHIFT = len(sys.argv) > 4 and sys.argv[4] == "shift"       # odd engines run half a frame out of phase: EKF half first, tracker half second
dev = 0
torch.cuda.set_device(dev)
mode = os.environ.get("TE_PRELOAD", "")
if mode == "1":                                             # a third, idle engine with some history (what bench.py's C2 leg leaves behind)
    tb0 = bench.TrackerBench(S, dev, seed=7)
    for _ in range(40):
        tb0.step()
    torch.cuda.synchronize()
elif mode == "2":                                           # ... constructed only
    tb0 = bench.TrackerBench(S, dev, seed=7)
    torch.cuda.synchronize()
elif mode == "3":                                           # ... only its memory footprint
    ballast = torch.empty(10 * 1024 ** 3, dtype=torch.uint8, device="cuda")
elif mode == "4":                                           # ... used, then destroyed
    tb0 = bench.TrackerBench(S, dev, seed=7)
    for _ in range(40):
        tb0.step()
    torch.cuda.synchronize()
    tb0.ctx.close() if hasattr(tb0.ctx, "close") else None
    del tb0
    torch.cuda.empty_cache()
elif mode == "5":                                           # ... a few extra torch streams that ran something
    xs = [torch.cuda.Stream() for _ in range(int(os.environ.get("TE_NSTREAMS", "3")))]
    for x in xs:
        with torch.cuda.stream(x):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
engines = []
for i in range(E):
    s = torch.cuda.Stream()
    sd = 1000 * i
    tb = bench.TrackerBench(S, dev, seed=sd)
    REAL = os.environ.get('TE_UNIFORM') is None
    tb.enable_chain(sd); tb.predicted_flow = REAL; tb.overlap = False
    eb = bench.VisualEkfBench(tb.ctx, S, dev, seed=sd, realistic=REAL)
    tb.tracked_fraction()
    torch.cuda.synchronize()
    tb.ctx.set_stream(s.cuda_stream)
    graphs = []
    with torch.cuda.stream(s):
        for _ in range(bench.N_CYCLE + (OFFSET if i == 0 else 0)):
            tb.step(); eb.step()
        s.synchronize()
        for _ in range(bench.N_CYCLE):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                if SHIFT and i % 2 == 1:
                    eb.step(); tb.step()
                else:
                    tb.step(); eb.step()
            graphs.append(g)
        s.synchronize()
    engines.append((s, tb, eb, graphs))
torch.cuda.synchronize()

def step(k):
    for s, tb, eb, graphs in engines:
        with torch.cuda.stream(s):
            graphs[k % bench.N_CYCLE].replay()

for k in range(bench.N_CYCLE):
    step(k)
torch.cuda.synchronize()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    torch.cuda.synchronize()
    res.append((time.perf_counter() - t0) / K * 1e3)
ms = sorted(res)[1]
print(f"engines {E} x {S} sequences{' (odd engines phase-shifted)' if SHIFT else ''}: {ms:.3f} ms per step of {E * S} frames -> {E * S / ms:.1f} k frames/s   (runs {['%.3f' % r for r in res]})")
