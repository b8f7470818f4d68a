#!/bin/bash
# r04 GPU session 3: default bench line (4 lanes), 2-rank smoke on one device, profile collection (kernel stats + PMC), 4-lane trace
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r04_3; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.err; head -c 700 $O/bench.json; echo
( time HV_BENCH_FORCE_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --sequences 256 --engines 2 --no-cpu-baseline --no-c4 --no-pcie ) > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err
tail -c 300 $O/bench_2ranks.err; head -c 500 $O/bench_2ranks_one_device.json; echo
PROF_DIR=r04_3/prof bash scripts/collect_profile.sh 2>&1 | tail -15
bash scripts/trace_lanes.sh 4 1024 r04_3/trace_lanes4 2>&1 | tail -22
