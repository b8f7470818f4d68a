#!/usr/bin/env python3
"""Model of two schedules of a frame's visit loop (no GPU needed): the lock-step the library runs (visit k of EVERY filter, two length
classes beside each other, shared update grids) against the per-filter-cursor scheme VERDICT r03 item 1 proposes (every filter has its
own visit cursor; the short and the long class are two independent launch chains, each round of a chain takes the next pending visit
of every filter of its class). Launch times are the measured ones of r04 (profiles/r04/kernel_stats.csv, track_length_sweep_*.json),
as functions of the records a launch holds; the cursor scheme gets every benefit of the doubt: its two chains never slow each other
down, a filter changes chain without delay, rounds start the moment a filter is ready. Workload: bench.py's realistic frame
(sample_track_lengths, inliers p = 0.25, quota 5, 20 visits).
usage: visit_schedule_model.py [filters] [frames]"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
FRAMES = int(sys.argv[2]) if len(sys.argv) > 2 else 20
V, QUOTA, NP_SHORT = bench.VISITS, bench.QUOTA, 11


def t_short(n):      # fused prepare + gate, two workgroups per CU: 55 us for one round of <= 512 records, 82 us at 810 (measured alone)
    return 0.0 if n == 0 else 55.0 + 27.0 * (math.ceil(n / 512) - 1)


def t_long(n):       # prepare + big gate, one workgroup per CU: 95 us per round of <= 256 records
    return 95.0 * math.ceil(n / 256)


def t_upd(n):        # one block update launch, one workgroup per CU: 57 us per round
    return 57.0 * math.ceil(n / 256)


def frame_inputs(rng):
    lens = bench.sample_track_lengths(rng, (V, B))
    inl = rng.uniform(size=(V, B)) < 0.25
    return lens > NP_SHORT, inl


def lockstep(is_long, inl):
    done = np.zeros(B, int)
    t = 0.0
    for k in range(V):
        act = done < QUOTA
        nS, nL = int((act & ~is_long[k]).sum()), int((act & is_long[k]).sum())
        iS, iL = int((act & ~is_long[k] & inl[k]).sum()), int((act & is_long[k] & inl[k]).sum())
        t += max(t_short(nS), t_long(nL))                              # the two classes beside each other
        t += t_upd(iS + iL) if iS + iL else 0.0                        # short inliers + long block 1 in one grid
        t += t_upd(iL) if iL else 0.0                                  # long block 2
        done += (act & inl[k])
    return t


def cursors(is_long, inl):
    cur = np.zeros(B, int)               # next visit of every filter
    done = np.zeros(B, int)
    free_at = np.zeros(B)                # time the filter's previous visit completed
    clock = {"S": 0.0, "L": 0.0}         # the two chains
    def pending(cls):
        alive = (cur < V) & (done < QUOTA)
        nxt_long = np.zeros(B, bool)
        nxt_long[alive] = is_long[cur[alive], np.nonzero(alive)[0]]
        return alive & (nxt_long == (cls == "L"))
    while True:
        alive = (cur < V) & (done < QUOTA)
        if not alive.any():
            break
        # the chain that can start a round first does so
        best = None
        for cls in ("S", "L"):
            p = pending(cls)
            if not p.any():
                continue
            start = max(clock[cls], free_at[p].min())
            if best is None or start < best[1]:
                best = (cls, start, p)
        cls, start, p = best
        take = p & (free_at <= start)
        idx = np.nonzero(take)[0]
        k = cur[idx]
        acc = inl[k, idx]
        n, ni = len(idx), int(acc.sum())
        if cls == "S":
            t_gate = t_short(n); t_done = t_gate + (t_upd(ni) if ni else 0.0)
        else:
            t_gate = t_long(n); t_done = t_gate + (2 * t_upd(ni) if ni else 0.0)
        free_at[idx[~acc]] = start + t_gate
        free_at[idx[acc]] = start + t_done
        done[idx[acc]] += 1
        cur[idx] += 1
        clock[cls] = start + t_done
    return float(free_at.max())


rng = np.random.default_rng(0)
ls, cs = [], []
for _ in range(FRAMES):
    a, b = frame_inputs(rng)
    ls.append(lockstep(a, b)); cs.append(cursors(a, b))
print(f"filters {B}, {FRAMES} frames of {V} visits (quota {QUOTA}): lock-step {np.mean(ls) / 1e3:.2f} ms per frame (min {min(ls) / 1e3:.2f}, max {max(ls) / 1e3:.2f}); "
      f"per-filter cursors, two independent chains {np.mean(cs) / 1e3:.2f} ms (min {min(cs) / 1e3:.2f}, max {max(cs) / 1e3:.2f})")
