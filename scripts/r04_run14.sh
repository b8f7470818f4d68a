#!/bin/bash
# r04 GPU session 14 (final code, second box): default bench line + profile collection
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_14; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.err; head -c 300 $O/bench.json; echo
PROF_DIR=r04_14/prof bash scripts/collect_profile.sh 2>&1 | tail -3
