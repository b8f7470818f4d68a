#!/bin/bash
# r04 GPU session 4: track-length sweep (the 48 / 49-row class boundary with the one-launch long class), full GPU test suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_4; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/track_length_sweep.py 256 2>/dev/null | tail -1 > $O/track_length_sweep_256.json
timeout 600 python scripts/track_length_sweep.py 1024 2>/dev/null | tail -1 > $O/track_length_sweep_1024.json
python - <<'PY'
import json
for B in (256, 1024):
    d = json.load(open(f"gpurun_out/r04_4/track_length_sweep_{B}.json"))
    print("B", B)
    for n, r in d["poses"].items():
        print(n, r["rows"], {p: (r[p]["all_rejected"]["wall_us"], r[p]["all_inliers"]["wall_us"]) for p in ("r04", "r03_long_two_launches", "r02_dense")})
PY
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/tests.log 2>&1
tail -6 $O/tests.log
