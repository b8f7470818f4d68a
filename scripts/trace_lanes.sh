#!/bin/bash
# kernel trace of the lanes replay (scripts/lanes_probe.py E S 4): start / end / queue of the dispatches of a few steps + overlap statistics
# usage: trace_lanes.sh [E] [S] [outdir-name]     (HV_* knobs are passed through)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
E=${1:-2}; S=${2:-1024}; OUT=$R/gpurun_out/${3:-trace_lanes}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
LP_MODE=lanes rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python $R/scripts/lanes_probe.py $E $S 4 > $OUT/log.txt 2>&1
F=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hv = [r for r in rows if "hv::" in r["Kernel_Name"]]
def name(r):
    m_ = re.search(r"(\w+_kernel\w*)", r["Kernel_Name"])
    return m_.group(1) if m_ else r["Kernel_Name"][:40]
# the replayed steps = the tail of the trace (3 x 4 timed steps + warm-up): take the last third of the dispatches
sel = hv[-(len(hv) // 3):]
t0 = int(sel[0]["Start_Timestamp"])
with open(sys.argv[2] + "/timeline.txt", "w") as f:
    for r in sel[:1200]:
        s_, e_ = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        f.write(f"{s_/1e3:10.1f} {e_/1e3:10.1f} {(e_-s_)/1e3:8.1f} q{r.get('Queue_Id','?')} {name(r)} grid={r.get('Grid_Size','?')}\n")
# statistics over the selection: per-kernel mean duration, time with >= k kernels in flight, per-queue busy share
ev = []
for r in sel:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[depth] += t - last; last = t; depth += d
tot = sum(hist.values())
per = collections.defaultdict(list)
for r in sel:
    per[name(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
q = collections.Counter()
for r in sel:
    q[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
with open(sys.argv[2] + "/stats.txt", "w") as f:
    f.write(f"window {tot/1e6:.3f} ms, {len(sel)} dispatches\n")
    f.write("kernels in flight -> share of the window: " + ", ".join(f"{k}: {v/tot:.3f}" for k, v in sorted(hist.items())) + "\n")
    f.write("queue busy share: " + ", ".join(f"q{k}: {v/tot:.3f}" for k, v in sorted(q.items())) + "\n")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        f.write(f"{k:34s} n={len(v):5d} mean {sum(v)/len(v):8.1f} us  total {sum(v)/1e3:8.2f} ms  share of window {sum(v)*1e3/tot:.3f}\n")
print(open(sys.argv[2] + "/stats.txt").read())
PY
grep -a LANES_PROBE $OUT/log.txt | cut -c1-220
