"""The multi-GPU contract of bench.py without a GPU: world_size 2 over gloo (replicas
only -- no data-path collective; the only collectives are the timing barrier and the MAX reduce)."""
import json
import os
import socket
import subprocess
import sys

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_timing_contract_over_gloo():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "_dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["world"] == 2 and d["calls"] == 5
    assert d["seconds"] >= 5 * 0.03 * 0.9                      # rank 1 sleeps 30 ms per step: MAX over ranks
    assert abs(d["value"] - 2 * 7 * 5 / d["seconds"]) < 1e-9   # whole-job aggregate: all ranks' units / max time
    assert d["inputs_differ"]                                   # rank-seeded inputs: replicas do not repeat each other
    assert d["max_barrier_wait_s"] < 1.0                        # no rank sits at a barrier while another does extra legs


def test_single_process_needs_no_rendezvous():
    env = bench.DistEnv("gloo", use_cuda=False)
    assert env.world == 1 and env.dist is None
    assert env.max_over_ranks(1.5) == 1.5
    assert bench.aggregate_value(256, 1, 10, 2.0) == 1280.0


def test_algorithmic_byte_accounting():
    ab = bench.algorithmic_bytes()
    # SURVEY.md section 8(d)
    assert ab["pyr_image"] == 2_397_000 and ab["klt_call"] == 4_915_200 and ab["stereo_frame"] == 14_624_400
    assert ab["pyr_l0_ref"] + ab["pyr_ln"] == ab["pyr_image"]
    # the level-0 launch's own bytes: the level-0 gradient plane is not stored by default (the LK kernel forms it)
    assert ab["pyr_l0"] == (360_960 + 90_240 if not bench.L0_GRADIENTS_STORED else ab["pyr_l0_ref"])
    big = bench.algorithmic_bytes(1280, 720, 400)
    assert big["pyr_image"] == 6_120_000 and big["stereo_frame"] == 31_900_800


def test_realistic_c3_workload_generator():
    """The headline's EKF workload: track lengths 5 + Geometric(0.2) capped at the trail, GAP-like pose sets (the newest poses + one
    older), padded records for the ragged API (bench.sample_track_lengths / make_visual_frame_realistic)."""
    import numpy as np
    rng = np.random.default_rng(0)
    lens = bench.sample_track_lengths(rng, 200000)
    assert lens.min() == 5 and lens.max() == 21 and 8.6 < lens.mean() < 9.1
    assert 0.19 < (lens > 11).mean() < 0.23                      # the long class of the two-class visit loop (> 44 rows in stereo)
    T1, T2, means, lens, idx, feat, vel, y = bench.make_visual_frame_realistic(np.random.default_rng(1), 96, distinct=32)
    assert lens.shape == (bench.VISITS, 96) and idx.shape == (bench.VISITS, 96, 21) and feat.shape == (bench.VISITS, 96, 42, 2)
    assert y.shape == (bench.VISITS, 96, 84) and means.shape == (96, 160)
    k, b = 3, 5
    n = int(lens[k, b])
    assert idx[k, b, :n - 1].tolist() == list(range(n - 1)) and n - 1 <= idx[k, b, n - 1] <= 20      # newest poses + an older one
    assert not idx[k, b, n:].any() and not feat[k, b, 2 * n:].any() and not y[k, b, 4 * n:].any()   # padding
    assert np.array_equal(lens[:, :32], lens[:, 32:64])                                               # `distinct` filters, tiled
