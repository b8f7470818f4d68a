#!/bin/bash
# r04 GPU session 1: full GPU test suite, lanes placement probes, default bench line. Outputs under gpurun_out/r04_1/
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --maxfail=15 -x --deselect tests/test_gpu_lanes.py ) > $O/tests.log 2>&1
tail -5 $O/tests.log
( time timeout 600 python -m pytest tests/test_gpu_lanes.py -q --maxfail=5 ) > $O/tests_lanes.log 2>&1
tail -5 $O/tests_lanes.log
for cfg in "lanes 0" "lanes 6" "torch 0" "torch 6"; do
  set -- $cfg
  LP_MODE=$1 LP_PRELOAD=$2 timeout 300 python scripts/lanes_probe.py 2 1024 24 2>&1 | grep -a "LANES_PROBE\|Error\|error" | tail -3 >> $O/probe.log
done
HV_EKF_SIDE_STREAM=0 LP_MODE=lanes timeout 300 python scripts/lanes_probe.py 2 1024 24 2>&1 | grep -a "LANES_PROBE\|Error" | tail -2 >> $O/probe.log
HV_EKF_LONG_FUSED=0 LP_MODE=lanes timeout 300 python scripts/lanes_probe.py 2 1024 24 2>&1 | grep -a "LANES_PROBE\|Error" | tail -2 >> $O/probe.log
LP_MODE=lanes timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | grep -a "LANES_PROBE\|Error" | tail -2 >> $O/probe.log
HV_EKF_LONG_FUSED=0 LP_MODE=lanes timeout 300 python scripts/lanes_probe.py 1 1024 24 2>&1 | grep -a "LANES_PROBE\|Error" | tail -2 >> $O/probe.log
cat $O/probe.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.err; head -c 1500 $O/bench.json
