#!/bin/bash
# r04 GPU session 24: the default bench line + profile collection of the final code on another box (session 23's box ran every f64 / HBM-bound
# kernel 14 - 25 % slower than the other boxes of the round: ekf_predict 0.172 against 0.151 ms with unchanged code)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_24; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.err; head -c 300 $O/bench.json; echo
PROF_DIR=r04_24/prof bash scripts/collect_profile.sh 2>&1 | tail -3
