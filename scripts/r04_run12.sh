#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_12; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/klt_ab.py 1024 10 6 2>&1 | grep -a "KLT_AB\|Error\|error" | tee -a $O/klt_ab.log
KLT_A=1 KLT_B=7 timeout 600 python scripts/klt_ab.py 1024 10 4 2>&1 | grep -a "KLT_AB\|Error\|error" | tee -a $O/klt_ab.log
timeout 600 python -m pytest tests/test_gpu_pyrlk.py -q 2>&1 | tail -3
