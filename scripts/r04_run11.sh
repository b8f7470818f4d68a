#!/bin/bash
# r04 GPU session 11: batch sizes that fill the rounds of the one-workgroup-per-CU kernels (long class <= 256 records per visit)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_11; mkdir -p $O
export TMPDIR=/tmp
P() { timeout 500 python scripts/lanes_probe.py "$@" 2>&1 | grep -a "LANES_PROBE\|Error\|error" | tail -2 >> $O/probe.log; }
P 4 1200 16
P 4 1152 16
P 2 1200 24
P 3 1200 16
cat $O/probe.log
