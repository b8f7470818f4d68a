#!/bin/bash
# r04 GPU session 5: the new adaptive-threshold tests first, then the full suite
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_5; mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python -m pytest tests/test_gpu_visual_prepare.py -q -k "adaptive" --maxfail=5 ) > $O/tests_adaptive.log 2>&1
tail -30 $O/tests_adaptive.log
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=10 ) > $O/tests.log 2>&1
tail -8 $O/tests.log
