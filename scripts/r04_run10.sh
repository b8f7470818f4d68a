#!/bin/bash
# r04 GPU session 10: ragged / lanes tests after the fork arrangement became the default, then the final bench line + profiles
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_10; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_visual_prepare.py tests/test_gpu_lanes.py tests/test_gpu_large_grid.py tests/test_gpu_frame_chain.py -q --maxfail=10 ) > $O/tests.log 2>&1
tail -5 $O/tests.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 200 $O/bench.err; head -c 400 $O/bench.json; echo
PROF_DIR=r04_10/prof bash scripts/collect_profile.sh 2>&1 | tail -3
