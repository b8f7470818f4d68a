#!/usr/bin/env python3
"""A/B of two builds of the LK kernel IN ONE PROCESS (knob klt_tile: 5 = the default build, 6 = the KLT_ALT build of klt_alt.hip):
alternating blocks of steps of the C2 tracker leg at B sequences, klt launch time from hipEvents, plus a bit-exact comparison of the
two builds' outputs on the same inputs (statuses and positions of both LK calls of a step).
usage: klt_ab.py [B] [steps] [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from hybvio_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 4
A, Bk = int(os.environ.get("KLT_A", "5")), int(os.environ.get("KLT_B", "6"))
tb = bench.TrackerBench(B, 0, seed=0)
tb.overlap = False
for _ in range(bench.N_CYCLE):
    tb.step()
torch.cuda.synchronize()
# ---- parity: the same step from the same state with both builds ----
state = {k: getattr(tb, k).clone() for k in ("pts_left", "flow", "dispvec", "cur_left", "cur_right", "st1", "st2", "tracked")}
k0, pend = tb.k, tb.pending
outs = []
for knob in (A, Bk):
    for k, v in state.items():
        getattr(tb, k).copy_(v)
    tb.k, tb.pending = k0, pend
    tb.ctx.set_knob("klt_tile", knob)
    tb.step()
    torch.cuda.synchronize()
    outs.append([getattr(tb, k).clone() for k in ("cur_left", "cur_right", "st1", "st2")])
same = all(torch.equal(x.view(torch.uint8) if x.dtype != torch.uint8 else x, y.view(torch.uint8) if y.dtype != torch.uint8 else y) for x, y in zip(*outs))
tracked = int(outs[0][2].sum().item())
# ---- timing: alternating blocks ----
res = {A: [], Bk: []}
for r in range(ROUNDS):
    for knob in (A, Bk) if r % 2 == 0 else (Bk, A):
        tb.ctx.set_knob("klt_tile", knob)
        for _ in range(2):
            tb.step()
        torch.cuda.synchronize()
        tb.ctx.profile_enable(True); tb.ctx.profile_reset()
        for _ in range(STEPS):
            tb.step()
        torch.cuda.synchronize()
        ms, n = tb.ctx.profile_read(capi.K_KLT)
        tb.ctx.profile_enable(False)
        res[knob].append(ms / n)
fmt = lambda v: "[" + ", ".join(f"{x:.4f}" for x in v) + "]"
ma, mb = sum(res[A]) / len(res[A]), sum(res[Bk]) / len(res[Bk])
print(f"KLT_AB B={B}: knob {A}: {ma:.4f} ms per launch {fmt(res[A])}; knob {Bk}: {mb:.4f} ms {fmt(res[Bk])}; ratio {mb / ma:.4f}; "
      f"outputs bit-identical: {same} ({tracked} tracked points in the compared step)")
