#!/bin/bash
# r04 GPU session 23 (final code of the round): full GPU suite, smoke(), the default bench line, profile collection
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_23; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee $O/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 300 $O/bench.err; head -c 400 $O/bench.json; echo
PROF_DIR=r04_23/prof bash scripts/collect_profile.sh 2>&1 | tail -3
