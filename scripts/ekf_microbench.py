#!/usr/bin/env python3
"""Time hv_ekf_visual_dev (gate-only / update) and the other EKF kernels per launch at batch B."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hybvio_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = capi.Context(width=64, height=64, levels=1, pool_size=1)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ekf = capi.EkfBatch(ctx, capi.ekf_default_params(), B)
rng = np.random.default_rng(0)
for _ in range(24):
    ekf.predict(0.005, rng.normal(0, .05, (B, 3)), rng.normal(0, .05, (B, 3)) + [0, 0, 9.8])
    ekf.augment(None)
chi2 = torch.zeros(B, dtype=torch.float64, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")


def bench(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for nr, l in ((8, 76), (16, 160), (40, 160), (64, 160), (84, 160)):
    H = torch.from_numpy(rng.normal(size=(B, l, nr))).cuda()
    v = torch.from_numpy(0.02 * rng.normal(size=(B, nr))).cuda()
    tg = bench(lambda: ekf.visual_dev(nr, l, H.data_ptr(), v.data_ptr(), 0.05, 0, chi2.data_ptr(), st.data_ptr()))
    import ctypes as C
    capi.lib().hv_debug_ekf_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    sg = (C.c_longlong * 16)()
    capi.lib().hv_debug_ekf_phase_stamps(ekf._h, sg)
    gph = [sg[i + 1] - sg[i] for i in range(3)]
    tu = bench(lambda: ekf.visual_dev(nr, l, H.data_ptr(), v.data_ptr(), 0.05, 1, chi2.data_ptr(), st.data_ptr()))
    st8 = (C.c_longlong * 16)()
    capi.lib().hv_debug_ekf_phase_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    capi.lib().hv_debug_ekf_phase_stamps(ekf._h, st8)
    ph = [st8[i + 1] - st8[i] for i in range(5)]
    print(f"B={B} nr={nr:3d} l={l:3d}: gate {tg:7.1f} us [A,B,C]={gph}   update {tu:7.1f} us   update phases [A HP, B S, C chol, D chi2, E+F] ticks={ph} D0={st8[7]-st8[6]} A:[issue,zero+stage,item0,item1]={[st8[8]-st8[0], st8[9]-st8[8], st8[10]-st8[9], st8[11]-st8[10]]}")
dt = torch.full((B,), 0.005, dtype=torch.float64, device="cuda")
gy = torch.zeros((B, 3), dtype=torch.float64, device="cuda"); ac = torch.tensor([[0, 0, 9.8]] * B, dtype=torch.float64, device="cuda")
print(f"predict  {bench(lambda: ekf.predict_dev(dt.data_ptr(), gy.data_ptr(), ac.data_ptr())):7.1f} us")
dtn = torch.full((10, B), 0.005, dtype=torch.float64, device="cuda"); gyn = torch.zeros((10, B, 3), dtype=torch.float64, device="cuda")
acn = torch.tensor([[[0, 0, 9.8]] * B] * 10, dtype=torch.float64, device="cuda")
print(f"predict x10 in one launch {bench(lambda: ekf.predict_n_dev(10, dtn.data_ptr(), gyn.data_ptr(), acn.data_ptr())):7.1f} us")
sp = (C.c_longlong * 16)(); capi.lib().hv_debug_ekf_phase_stamps(ekf._h, sp)
print("predict phases [serial mean/F/L, P00 block, trailing rows] ticks =", [sp[13]-sp[12], sp[14]-sp[13], sp[15]-sp[14]])
if sp[5] > sp[0] > 0:
    print("predict chain kernel, first sample / first chunk (cycles): S0", sp[1]-sp[0], "stage B", sp[2]-sp[1], "q", sp[3]-sp[2], "R", sp[4]-sp[3], "mean entries", sp[5]-sp[4],
          "| start -> end of F / L", sp[6]-sp[12], "| sample 2: products 1", sp[8]-sp[7], "barrier", sp[9]-sp[8], "products 2 + stores", sp[10]-sp[9], "| chunk's five recursions", sp[11]-sp[6])
print(f"augment  {bench(lambda: ekf._chk(capi.lib().hv_ekf_augment(ekf._h, None, None), 'aug')):7.1f} us")
print(f"symmetr  {bench(lambda: ekf.symmetrize()):7.1f} us")
ekf.close(); ctx.close()
