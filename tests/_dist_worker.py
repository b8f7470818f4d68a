"""Worker for tests/test_bench_distributed.py: the N>1 path of bench.py (one process per rank,
replicas, barrier + MAX-over-ranks timing) on CPU with the gloo backend."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

env = bench.DistEnv("gloo", use_cuda=False)
work = {"calls": 0}


def step():                      # rank r is slower than rank 0: the reported time must be the slowest rank's
    work["calls"] += 1
    time.sleep(0.01 * (1 + 2 * env.rank))


seconds = env.timed(step, 5)
# every replica works on its own inputs: the per-rank seed must give different crops (gathered over the host group)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
mine = torch.from_numpy(bench.crop_offsets(16, env.rank).astype("int64"))
allo = [torch.zeros_like(mine) for _ in range(env.world)]
dist.all_gather(allo, mine)
if env.rank == 0:
    print(json.dumps({"world": env.world, "seconds": seconds, "calls": work["calls"],
                      "value": bench.aggregate_value(7, env.world, 5, seconds),
                      "inputs_differ": bool(any((allo[0] != a).any() for a in allo[1:])),
                      "max_barrier_wait_s": env.max_barrier_wait_s, "cores": env.cores}))
env.close()
