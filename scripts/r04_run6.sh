#!/bin/bash
# r04 GPU session 6 (final code): smoke, default bench line, profile collection (kernel stats + PMC)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_6; mkdir -p $O
export TMPDIR=/tmp
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err; head -c 600 $O/bench.json; echo
PROF_DIR=r04_6/prof bash scripts/collect_profile.sh 2>&1 | tail -12
