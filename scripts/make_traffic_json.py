#!/usr/bin/env python3
"""Turn the per-(kernel, grid) PMC summaries written by scripts/collect_profile.sh (pmc1..pmc4.csv) into
profiles/rNN/traffic.json, the file bench.py reads for roofline.traffic. Dispatches are selected BY GRID SIZE: the launch
of the profiled C2 / C3 step at the profiled B only, so the figures reproduce from the committed pmc*.csv."""
import csv
import json
import os
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r06"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06/traffic.json"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NPTS = bench.NPTS
B = 1024
try:                                                   # sequences per GPU of the profiled bench run
    with open(os.path.join(src, "bench_under_kernel_trace.log")) as f:
        B = [json.loads(l) for l in f if l.startswith("{")][-1]["config"]["sequences_per_gpu"]
except Exception:
    pass
val, disp = {}, {}
for i in range(1, 7):
    path = os.path.join(src, f"pmc{i}.csv")
    if not os.path.exists(path):
        continue
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["kernel"]
            for short in ("pyr_down_l0_kernel", "pyr_tail_kernel", "gftt_march_kernel"):      # summaries written with an older name pattern
                if short in name:
                    name = short
            if name == "ekf_update_dual_kernel":          # r03: the visit's two update launches (short class + long block 1 / rest + long block 2)
                name = "ekf_update_kernel"
            val[(name, int(r["grid"]), r["counter"])] = float(r["mean_per_dispatch"])
            disp[(name, int(r["grid"]))] = int(r["dispatches"])
alg = bench.algorithmic_bytes()
tiles_l0 = ((bench.W + 127) // 128) * ((bench.H + 31) // 32)
W1, H1 = (bench.W + 1) // 2, (bench.H + 1) // 2
wgs_l0 = (((W1 + 3) // 4) * ((H1 + 1) // 2) + 255) // 256          # pyr_down_l0_kernel: work items of one image / 256
if bench.L0_GRADIENTS_STORED:
    L0K, L0GRID = "pyr_level_kernel<true>", 2 * B * tiles_l0 * 256
else:
    L0K, L0GRID = "pyr_down_l0_kernel", 2 * B * wgs_l0 * 256
W2, H2 = (W1 + 1) // 2, (H1 + 1) // 2
wgs_l1 = (((W2 + 3) // 4) * ((H2 + 1) // 2) + 255) // 256          # the level-1 launch of the same kernel
GRID = {"klt_kernel": B * NPTS * 64, L0K: L0GRID, "ekf_update_kernel": B * 512, "pyr_tail_kernel": 2 * B * 512,
        "vu_gate_kernel_2percu": B * 384, "ekf_sparse_gate_kernel": B * 256, "vu_compact_kernel_2percu": B * 384,
        "vu_gate_long_kernel": B * 768,
        # r06, the split form of a visit: vu_tri_kernel_x2 (short class, 128 threads per track) / _x4 (long class, 256) share the summary
        # name "vu_tri_kernel" (scripts/pmc_summary.py) and differ by grid; the record-fed gates
        "vu_tri_kernel": B * 128, "vu_gate_rec_kernel": B * 256, "vu_gate_long_rec_kernel": B * 768}
GRID_TRI_LONG = B * 256
GRID_L1 = 2 * B * wgs_l1 * 256


def g(k, c):
    return val.get((k, GRID[k], c))


def hbm(k):      # KB -> bytes; gfx950 FETCH_SIZE counts 64 B per 128 B request (MI355X_MICROARCH.md, HBM section): x2
    f, w = g(k, "FETCH_SIZE"), g(k, "WRITE_SIZE")
    return None if f is None or w is None else (2 * f + w) * 1024.0


def frac(num, den, scale=1.0):
    return None if num is None or not den else num * scale / den


out = {
    "note": f"rocprofv3 --pmc (separate passes, scripts/collect_profile.sh) on `bench.py --only-headline` at B={B}; every figure is the mean "
            "over the dispatches of ONE (kernel, grid size): the launch of the C2 / C3 step itself. gfx950: FETCH_SIZE counts 64 B per "
            "128 B request -> x2; WRITE_SIZE in KB.",
    "sequences_per_gpu": B,
    "selected_grids": {k: {"grid_threads": v, "dispatches": disp.get((k, v))} for k, v in GRID.items()},
    "klt_kernel": {
        "fetch_kb_raw": g("klt_kernel", "FETCH_SIZE"), "write_kb": g("klt_kernel", "WRITE_SIZE"),
        "hbm_bytes_per_launch": hbm("klt_kernel"), "algorithmic_bytes_per_launch": alg["klt_call"] * B,
        "valu_insts_per_feature": frac(g("klt_kernel", "SQ_INSTS_VALU"), B * NPTS),
        "salu_insts_per_feature": frac(g("klt_kernel", "SQ_INSTS_SALU"), B * NPTS),
        "lds_insts_per_feature": frac(g("klt_kernel", "SQ_INSTS_LDS"), B * NPTS),
    },
    "pyr_level_kernel_L0": {
        "kernel": L0K,
        "fetch_kb_raw": g(L0K, "FETCH_SIZE"), "write_kb": g(L0K, "WRITE_SIZE"),
        "hbm_bytes_per_launch": hbm(L0K), "algorithmic_bytes_per_launch": alg["pyr_l0"] * 2 * B,
        "how": f"the level-0 launch is the {L0K} dispatch with grid {L0GRID} threads (2B images)",
    },
    "pyr_down_l0_kernel": {
        "hbm_bytes_per_launch": hbm(L0K), "algorithmic_bytes_per_launch": alg["pyr_l0"] * 2 * B,
        "how": f"level 0: the dispatch with grid {L0GRID} threads (2B images)",
    },
    "pyr_down_l0_kernel_L1": {
        "hbm_bytes_per_launch": (lambda f, w: None if f is None or w is None else (2 * f + w) * 1024.0)(
            val.get((L0K, GRID_L1, "FETCH_SIZE")), val.get((L0K, GRID_L1, "WRITE_SIZE"))),
        "how": f"level 1: the dispatch of the same kernel with grid {GRID_L1} threads",
    },
    "vu_gate_kernel_2percu": {
        "hbm_bytes_per_launch": hbm("vu_gate_kernel_2percu"),
        "mfma_busy_frac": frac(g("vu_gate_kernel_2percu", "SQ_VALU_MFMA_BUSY_CYCLES"), 32.0 * (g("vu_gate_kernel_2percu", "SQ_BUSY_CYCLES") or 0)),
        "wave_parked_frac": frac(g("vu_gate_kernel_2percu", "SQ_WAIT_ANY"), g("vu_gate_kernel_2percu", "SQ_WAVE_CYCLES")),
        "valu_insts_per_track": frac(g("vu_gate_kernel_2percu", "SQ_INSTS_VALU"), B),
        "mfma_insts_per_track": frac(g("vu_gate_kernel_2percu", "SQ_INSTS_MFMA"), B),
        "how": "fused triangulation + prepareVisualUpdate + column-sparse chi2 gate, one launch per track visit (short-track class of the ragged frame loop)",
    },
    "vu_gate_long_kernel": {
        "hbm_bytes_per_launch": hbm("vu_gate_long_kernel"),
        "mfma_busy_frac": frac(g("vu_gate_long_kernel", "SQ_VALU_MFMA_BUSY_CYCLES"), 32.0 * (g("vu_gate_long_kernel", "SQ_BUSY_CYCLES") or 0)),
        "wave_parked_frac": frac(g("vu_gate_long_kernel", "SQ_WAIT_ANY"), g("vu_gate_long_kernel", "SQ_WAVE_CYCLES")),
        "how": "r04: triangulation + prepareVisualUpdate + the chi2 gate on the factors of the Jacobian (structured_S) of the long-track class (12 .. 21 stereo poses) in one launch per visit; ~21 % of the records of a launch are live",
    },
    # r06: the split form's kernels (short class: triangulation then record-fed gate; long class likewise)
    **{k: {"hbm_bytes_per_launch": hbm(k),
           "mfma_busy_frac": frac(g(k, "SQ_VALU_MFMA_BUSY_CYCLES"), 32.0 * (g(k, "SQ_BUSY_CYCLES") or 0)),
           "wave_parked_frac": frac(g(k, "SQ_WAIT_ANY"), g(k, "SQ_WAVE_CYCLES")),
           "valu_insts_per_track": frac(g(k, "SQ_INSTS_VALU"), B), "mfma_insts_per_track": frac(g(k, "SQ_INSTS_MFMA"), B),
           "lds_bank_conflict_frac": frac(g(k, "SQ_LDS_BANK_CONFLICT"), g(k, "SQ_LDS_IDX_ACTIVE")), "how": how}
       for k, how in (("vu_tri_kernel", "short class: pose trail + two-camera start + Gauss-Newton with derivatives + per-pose part of prepareVisualUpdate -> factor record (128 threads per track)"),
                      ("vu_gate_rec_kernel", "short class: compact Jacobian from the factor record + column-sparse chi2 gate, three workgroups per CU"),
                      ("vu_gate_long_rec_kernel", "long class: the gate on the Jacobian's factors (structured_S) from the factor record"))},
    "vu_tri_kernel_long": {
        "hbm_bytes_per_launch": (lambda f, w: None if f is None or w is None else (2 * f + w) * 1024.0)(
            val.get(("vu_tri_kernel", GRID_TRI_LONG, "FETCH_SIZE")), val.get(("vu_tri_kernel", GRID_TRI_LONG, "WRITE_SIZE"))),
        "wave_parked_frac": frac(val.get(("vu_tri_kernel", GRID_TRI_LONG, "SQ_WAIT_ANY")), val.get(("vu_tri_kernel", GRID_TRI_LONG, "SQ_WAVE_CYCLES"))),
        "valu_insts_per_track": frac(val.get(("vu_tri_kernel", GRID_TRI_LONG, "SQ_INSTS_VALU")), B),
        "how": f"long class (12 .. 21 stereo poses): the same front with 256 threads per track, grid {GRID_TRI_LONG} threads",
    },
    "pyr_tail_kernel": {
        "fetch_kb_raw": g("pyr_tail_kernel", "FETCH_SIZE"), "write_kb": g("pyr_tail_kernel", "WRITE_SIZE"),
        "hbm_bytes_per_launch": hbm("pyr_tail_kernel"),
        "how": "levels 2.. of 2B images: reads the level-2 gray interior, writes gradients of levels 2.., the next gray levels and the physical borders",
    },
    "ekf_update_kernel": {
        "hbm_bytes_per_launch": hbm("ekf_update_kernel"),
        "algorithmic_bytes_per_launch_gate": B * (160 * 160 + 40 * 160) * 8,
        "algorithmic_bytes_per_launch_update": B * (2 * 160 * 160 + 40 * 160) * 8,
        "mix": "updateVisualTrack of the visits the gate accepted (compact Jacobian): ekf_update_dual_kernel, two launches per visit -- inliers of the short class + first block of the long class, then the rest + second block (~25 % of the filters per launch are live)",
        "mfma_busy_frac": frac(g("ekf_update_kernel", "SQ_VALU_MFMA_BUSY_CYCLES"), 32.0 * (g("ekf_update_kernel", "SQ_BUSY_CYCLES") or 0)),
        "mfma_insts_per_filter": frac(g("ekf_update_kernel", "SQ_INSTS_MFMA"), B),
    },
    "measured_hbm_ceilings_GBs": {"read": 6509.5, "write": 4755.2, "copy": 4746.5, "pyramid_mix_4r_17w_nt": 4931.3,
                                  "source": "scripts/hbm_ceiling.hip, profiles/r01/hbm_ceiling_measured.txt"},
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
