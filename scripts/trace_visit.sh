#!/bin/bash
# kernel-trace of two C3 steps (timestamps per dispatch): shows how the launches of a track visit overlap / queue
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-trace_visit}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tv
rocprofv3 --kernel-trace --output-format csv -d /tmp/tv -o t -- python $R/bench.py --only-headline --steps 2 --warmup 1 --repeats 1 --verify 0 > $OUT/log.txt 2>&1
F=$(find /tmp/tv -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT/timeline.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 400 dispatches of our kernels
sel = [r for r in rows if "hv::" in r["Kernel_Name"]][-420:]
t0 = int(sel[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    for r in sel:
        import re
        m_ = re.search(r"(\w+_kernel\w*)", r["Kernel_Name"])
        name = m_.group(1) if m_ else r["Kernel_Name"][:40]
        s_, e_ = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f.write(f"{s_/1e3:10.1f} {e_/1e3:10.1f} {(e_-s_)/1e3:8.1f} q{r.get('Queue_Id','?')} {name} grid={r.get('Grid_Size','?')}\n")
PY
tail -5 $OUT/log.txt | cut -c1-200
