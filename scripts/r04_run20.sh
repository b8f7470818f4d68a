#!/bin/bash
# r04 GPU session 20: the fused gates on the factors of the Jacobian (structured_S) -- parity tests, per-visit microbench, single-sequence latency
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_20; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_visual_prepare.py -m gpu -x -q 2>&1 | tail -12 | tee $O/tests.log
for np in 21 16 12 10 6; do timeout 120 python scripts/vu_microbench.py 1 $np 1 2>&1 | grep "all rejected"; done | tee $O/visit_b1.txt
for np in 21 10; do timeout 120 python scripts/vu_microbench.py 1024 $np 1 2>&1 | grep "all rejected\|0.25"; done | tee $O/visit_b1024.txt
timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1 | tee $O/latency.txt
