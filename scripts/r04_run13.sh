#!/bin/bash
# r04 GPU session 13 (final code): full GPU test suite, smoke, default bench line, profile collection
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_13; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/tests.log 2>&1
tail -6 $O/tests.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; grep -a "smoke ok" $O/smoke.log | cut -c1-200
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.err; head -c 300 $O/bench.json; echo
PROF_DIR=r04_13/prof bash scripts/collect_profile.sh 2>&1 | tail -3
