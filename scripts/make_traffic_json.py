#!/usr/bin/env python3
"""Turn the per-kernel PMC summaries written by scripts/collect_profile.sh (pmc1..pmc5.csv, means per
dispatch at B = 256) into profiles/rNN/traffic.json, the file bench.py reads for roofline.traffic."""
import csv, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_final"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01/traffic.json"
NPTS, LEVEL_SHARE_L0 = 200, 16.0 / 21.0               # <true> launches L0, L1, L2 scale 16 : 4 : 1
B = 256
try:                                                   # sequences per GPU of the profiled bench run
    with open(os.path.join(src, "bench_under_kernel_trace.log")) as f:
        B = [json.loads(l) for l in f if l.startswith("{")][-1]["config"]["sequences_per_gpu"]
except Exception:
    pass
val = {}
for i in range(1, 6):
    with open(os.path.join(src, f"pmc{i}.csv")) as f:
        for r in csv.DictReader(f):
            val[(r["kernel"], r["counter"])] = float(r["mean_per_dispatch"])
g = lambda k, c: val[(k, c)]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
alg = bench.algorithmic_bytes()

def hbm(k, fetch_x2=True):      # KB -> bytes; gfx950 FETCH_SIZE counts 64 B per 128 B request
    return (g(k, "FETCH_SIZE") * (2 if fetch_x2 else 1) + g(k, "WRITE_SIZE")) * 1024.0

pyr = "pyr_level_kernel<true>"
out = {
    "note": f"rocprofv3 --pmc (separate passes, scripts/collect_profile.sh), means per dispatch of bench.py at B={B}. "
            "gfx950: FETCH_SIZE counts 64 B per 128 B request -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE "
            "calibrated against the pyramid's known write bytes (1.03x).",
    "sequences_per_gpu": B,
    "klt_kernel": {
        "fetch_kb_raw": g("klt_kernel", "FETCH_SIZE"), "write_kb": g("klt_kernel", "WRITE_SIZE"),
        "hbm_bytes_per_launch": hbm("klt_kernel"),
        "algorithmic_bytes_per_launch": alg["klt_call"] * B,
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 32 SQ instances SQ_BUSY_CYCLES is reported for
        "valu_busy_frac": g("klt_kernel", "SQ_ACTIVE_INST_VALU") * 4 / (g("klt_kernel", "SQ_BUSY_CYCLES") * 32.0),
        "valu_insts_per_feature": g("klt_kernel", "SQ_INSTS_VALU") / (B * NPTS),
    },
    "pyr_level_kernel_L0": {
        "fetch_kb_raw_est": g(pyr, "FETCH_SIZE") * 3 * LEVEL_SHARE_L0, "write_kb_est": g(pyr, "WRITE_SIZE") * 3 * LEVEL_SHARE_L0,
        "hbm_bytes_per_launch": hbm(pyr) * 3 * LEVEL_SHARE_L0,
        "algorithmic_bytes_per_launch": alg["pyr_l0"] * 2 * B,
        "how": "the three <true> launches (L0,L1,L2) share one kernel name; their bytes scale 16:4:1, so L0 = 16/21 of the per-name sum",
    },
    "ekf_update_kernel": {
        "hbm_bytes_per_launch": hbm("ekf_update_kernel"),
        "algorithmic_bytes_per_launch_gate": B * (160 * 160 + 40 * 160) * 8,
        "algorithmic_bytes_per_launch_update": B * (2 * 160 * 160 + 40 * 160) * 8,
        "mix": "15 chi2 gates + 5 fused gate+update launches per frame",
        "mfma_busy_frac": g("ekf_update_kernel", "SQ_VALU_MFMA_BUSY_CYCLES") / (32.0 * g("ekf_update_kernel", "SQ_BUSY_CYCLES")),
        "mfma_insts_per_filter": g("ekf_update_kernel", "SQ_INSTS_MFMA") / B,
    },
    "ekf_augment_kernel": {"hbm_bytes_per_launch": hbm("ekf_augment_kernel"), "algorithmic_bytes_per_launch": B * 2 * 160 * 160 * 8},
    "ekf_symmetrize_kernel": {"hbm_bytes_per_launch": hbm("ekf_symmetrize_kernel"), "algorithmic_bytes_per_launch": B * 2 * 160 * 160 * 8},
    "measured_hbm_ceilings_GBs": {"read": 6509.5, "write": 4755.2, "copy": 4746.5, "pyramid_mix_4r_17w_nt": 4931.3,
                                  "source": "scripts/hbm_ceiling.hip, profiles/r01/hbm_ceiling_measured.txt"},
}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
