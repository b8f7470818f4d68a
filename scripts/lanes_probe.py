#!/usr/bin/env python3
"""Experiment: the realistic C3 step of bench.py on E lanes of one hv_lanes set (library-owned high-priority streams, r04) or -- LP_MODE=torch,
r03's arrangement -- on E engines with torch streams, S sequences each, after different process histories (LP_PRELOAD): does the
step time depend on what the process did before the engines were created?
usage: [LP_MODE=lanes|torch] [LP_PRELOAD=0..5] [HV_<KNOB>=v ...] lanes_probe.py E S [steps]
LP_PRELOAD: 0 nothing; 1 an idle tracker object with history (what bench.py's C2 leg leaves behind); 5 three torch streams that ran
a kernel; 6 = 1 + 5 + a captured and replayed HIP graph with a forked branch (every kind of stream the runtime creates)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from hybvio_amd import capi

E, S = int(sys.argv[1]), int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
MODE = os.environ.get("LP_MODE", "lanes")
PRE = int(os.environ.get("LP_PRELOAD", "0"))
EAGER = os.environ.get("LP_EAGER", "0") == "1"              # time eager launches instead of HIP-graph replay
dev = 0
torch.cuda.set_device(dev)
keep = []
if PRE in (1, 6):
    tb0 = bench.TrackerBench(S, dev, seed=7)
    for _ in range(40):
        tb0.step()
    torch.cuda.synchronize()
    keep.append(tb0)
if PRE in (5, 6):
    xs = [torch.cuda.Stream() for _ in range(3)]
    for x in xs:
        with torch.cuda.stream(x):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
    keep.append(xs)
if PRE == 6:
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros(1 << 20, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s1):
        a.add_(1); s1.synchronize()
        with torch.cuda.graph(g, stream=s1):
            a.add_(1)
            ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
            with torch.cuda.stream(s2):
                b = a * 2
            ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
            a.add_(b)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    keep.append(g)

lanes = capi.Lanes(E, width=bench.W, height=bench.H, levels=bench.LEVELS, max_tracks=bench.NPTS, pool_size=3 * S, max_pairs=S, device=dev) if MODE == "lanes" else None
engines = []
for i in range(E):
    sd = 1000 * i
    if lanes is not None:
        ctx = lanes.ctx[i]
        s = torch.cuda.ExternalStream(ctx.get_stream())
        tb = bench.TrackerBench(S, dev, seed=sd, ctx=ctx)
    else:
        s = torch.cuda.Stream()
        tb = bench.TrackerBench(S, dev, seed=sd)
    tb.enable_chain(sd); tb.predicted_flow = True; tb.overlap = False
    if lanes is None:
        tb.tracked_fraction()
        torch.cuda.synchronize()
        tb.ctx.set_stream(s.cuda_stream)
    graphs = []
    with torch.cuda.stream(s):
        eb = bench.VisualEkfBench(tb.ctx, S, dev, seed=sd, realistic=True)
        for _ in range(bench.N_CYCLE):
            tb.step(); eb.step()
        s.synchronize()
        for _ in range(bench.N_CYCLE):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                tb.step(); eb.step()
            graphs.append(g)
        s.synchronize()
    engines.append((s, tb, eb, graphs))
torch.cuda.synchronize()


def step(k):
    for s, tb, eb, graphs in engines:
        with torch.cuda.stream(s):
            if EAGER:
                tb.step(); eb.step()
            else:
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
knobs = {k: v for k, v in os.environ.items() if k.startswith("HV_")}
print(f"LANES_PROBE mode={MODE}{' eager' if EAGER else ''} preload={PRE} engines={E}x{S} knobs={knobs}: {ms:.3f} ms per step of {E * S} frames -> {E * S / ms:.1f} k frames/s   (runs {['%.3f' % r for r in res]})")
