#!/bin/bash
# r04 GPU session 17: latency legs of bench.py (speculative long loop, forked frame DAG) on a small headline
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_17; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --sequences 64 --engines 1 --steps 3 --warmup 1 --repeats 1 --verify 0 --no-cpu-baseline --no-c4 --no-gftt --no-ingest --no-visual-track --no-ransac --no-pcie > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_17/bench.json").read().strip().splitlines()[-1])
for k in ("latency_mode", "latency_mode_uniform", "latency_mode_graph", "latency_mode_forked"):
    print(k, json.dumps(d.get(k))[:400])
print("c2", json.dumps(d["c2"].get("latency_mode"))[:200], json.dumps(d["c2"].get("latency_mode_graph"))[:200])
PY
