"""Worker for tests/test_bench_distributed.py: the N>1 path of bench.py (one process per rank,
replicas, barrier + MAX-over-ranks timing) on CPU with the gloo backend."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

env = bench.DistEnv("gloo")
work = {"calls": 0}


def step():                      # rank r is slower than rank 0: the reported time must be the slowest rank's
    work["calls"] += 1
    time.sleep(0.01 * (1 + 2 * env.rank))


seconds = env.timed(step, 5)
if env.rank == 0:
    print(json.dumps({"world": env.world, "seconds": seconds, "calls": work["calls"],
                      "value": bench.aggregate_value(7, env.world, 5, seconds)}))
env.close()
