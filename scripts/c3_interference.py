#!/usr/bin/env python3
"""Why are the tracker kernels slower inside the C3 step than in the C2 step? Event-timed klt / pyramid kernels with different
neighbours on the stream. Prints one JSON object."""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from hybvio_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tb = bench.TrackerBench(B, 0, seed=0)
eb = bench.EkfBench(tb.ctx, B, 0, seed=0)
clk = []
stop = False
def poll():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
            clk.append((time.time(), o[:600]))
        except Exception as e:
            clk.append((time.time(), repr(e)))
        time.sleep(0.3)

def measure(name, fn, steps=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    tb.ctx.profile_enable(True); tb.ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r = {"ms_per_step": dt * 1e3}
    for k, kid in (("pyr_l0", capi.K_PYR_L0), ("pyr_ln", capi.K_PYR_LN), ("klt", capi.K_KLT), ("ekf_update", capi.K_EKF_UPDATE),
                   ("ekf_predict", capi.K_EKF_PREDICT), ("ekf_augment", capi.K_EKF_AUGMENT)):
        ms, n = tb.ctx.profile_read(kid)
        if n:
            r[k] = round(ms / n, 4)
    tb.ctx.profile_enable(False)
    return r

def ekf_gates_only():
    s = eb.k % eb.H.shape[0]
    for j in range(20):
        eb.ekf.visual_dev(bench.EKF_ROWS, bench.EKF_COLS, eb.H[s].data_ptr(), eb.v_out[(s + j) % 4].data_ptr(), 0.05, 0, eb.chi2.data_ptr(), eb.status.data_ptr())

def ekf_light():
    eb._predicts(); eb.ekf.symmetrize()

out = {}
th = threading.Thread(target=poll); th.start()
out["tracker_only"] = measure("t", tb.step)
out["tracker_plus_ekf"] = measure("te", lambda: (tb.step(), eb.step()))
out["tracker_plus_ekf_sync_sleep"] = measure("tes", lambda: (tb.step(), eb.step(), torch.cuda.synchronize(), time.sleep(0.01)))
out["tracker_plus_20_gates"] = measure("tg", lambda: (tb.step(), ekf_gates_only()))
out["tracker_plus_predict_sym"] = measure("tp", lambda: (tb.step(), ekf_light()))
out["ekf_only"] = measure("e", eb.step)
out["tracker_only_again"] = measure("t2", tb.step)
stop = True; th.join()
out["clock_samples"] = [c[1] for c in clk[:: max(1, len(clk) // 12)]]
print(json.dumps(out, indent=1))
