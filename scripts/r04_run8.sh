#!/bin/bash
# r04 GPU session 8: full GPU test suite on the final code (incl. the threaded and four-lane tests), smoke
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_8; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/tests.log 2>&1
tail -8 $O/tests.log
( time timeout 300 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; tail -4 $O/smoke.log | cut -c1-300
