#!/bin/bash
# r04 GPU session 15: symmetrise folded into the augmentation -- parity test, then A/B on 4 lanes x 1024 and on one lane (same box)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_15; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ekf.py -m gpu -x -q 2>&1 | tail -3 | tee $O/tests.log
for r in 1 2; do
  for split in 1 0; do
    echo "split=$split lanes=4"; HV_BENCH_SPLIT_SYM=$split timeout 300 python scripts/lanes_probe.py 4 1024 24 2>&1 | tail -2
  done
done | tee $O/ab4.txt
for split in 1 0; do
  echo "split=$split lanes=1 eager"; HV_BENCH_SPLIT_SYM=$split LP_EAGER=1 timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | tail -2
done | tee $O/ab1.txt
