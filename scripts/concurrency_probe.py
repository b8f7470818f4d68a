#!/usr/bin/env python3
"""How many kernels of DIFFERENT streams does the device run at once? k streams each run one spin kernel of ~1 ms (one thread);
wall time of the batch = 1 ms x ceil(k / concurrent kernels). Streams: torch's (default priority) and the high-priority streams of an
hv_lanes set (what the bench's lanes replay on). usage: concurrency_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from hybvio_amd import capi

torch.cuda.set_device(0)
CYC = 2_000_000


def run(streams, label):
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    out = []
    for k in range(1, len(streams) + 1):
        t0 = time.perf_counter()
        for s in streams[:k]:
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    print(f"CONCURRENCY {label}: ms for k = 1..{len(streams)} streams with one ~{out[0]:.2f} ms spin kernel each: " + ", ".join(f"{x:.2f}" for x in out))


run([torch.cuda.Stream() for _ in range(8)], "torch streams (default priority)")
run([torch.cuda.Stream(priority=-1) for _ in range(8)], "torch streams (high priority)")
lanes = capi.Lanes(4, width=64, height=64, levels=2, max_tracks=8, pool_size=4, max_pairs=1, device=0)
run([torch.cuda.ExternalStream(c.get_stream()) for c in lanes.ctx], "hv_lanes streams")
lanes8 = capi.Lanes(8, width=64, height=64, levels=2, max_tracks=8, pool_size=4, max_pairs=1, device=0)
run([torch.cuda.ExternalStream(c.get_stream()) for c in lanes8.ctx], "hv_lanes streams (8 lanes)")
