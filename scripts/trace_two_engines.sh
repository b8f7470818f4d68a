#!/bin/bash
# kernel-trace of the two-engine replay (scripts/two_engines.py): start / end / queue of the dispatches of a few steps, to show the
# two engines' launch chains running beside each other
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_two_engines
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/t2e
TE_PRELOAD=5 TE_NSTREAMS=2 rocprofv3 --kernel-trace --output-format csv -d /tmp/t2e -o t -- python $R/scripts/two_engines.py 2 1024 4 > $OUT/log.txt 2>&1
F=$(find /tmp/t2e -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT/timeline.txt" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "hv::" in r["Kernel_Name"]][-700:-300]
t0 = int(sel[0]["Start_Timestamp"])
busy = 0
with open(sys.argv[2], "w") as f:
    for r in sel:
        m_ = re.search(r"(\w+_kernel\w*)", r["Kernel_Name"])
        name = m_.group(1) if m_ else r["Kernel_Name"][:40]
        s_, e_ = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f.write(f"{s_/1e3:10.1f} {e_/1e3:10.1f} {(e_-s_)/1e3:8.1f} q{r.get('Queue_Id','?')} {name}\n")
PY
tail -2 $OUT/log.txt | cut -c1-200
