#!/bin/bash
# r04 GPU session 18 (experiment, NOT in the tree any more): the two block-update launches of a speculative pass over long records as ONE launch
# (knob ekf_spec_one_update, a kernel that ran the body twice behind a fence: 18 spilled VGPRs) -- 0.787 ms per frame against 0.777 with two
# launches (profiles/r04/spec_long_pair_update_ab.txt); reverted. The knob below no longer exists.
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_18; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_visual_prepare.py -m gpu -x -q -k "whole_frame_loop or contention" 2>&1 | tail -5 | tee $O/tests.log
for r in 1 2; do for one in 0 1; do
  echo "one_update=$one graph"; HV_EKF_SPEC_ONE_UPDATE=$one timeout 300 python scripts/lanes_probe.py 1 1 200 2>&1 | tail -1
done; done | tee $O/latency.txt
