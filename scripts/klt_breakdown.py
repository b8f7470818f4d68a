#!/usr/bin/env python3
"""Where the LK kernel's time goes: runs the bench's temporal LK call with the iteration cap forced to
0, 1, 2, 3 and the default (20). cap 0 = template patches + structure tensor only; (cap 1) - (cap 0) = one
J-tile staging + one iteration per level; (cap k+1) - (cap k) = one more iteration where still needed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hybvio_amd import capi

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tb = bench.TrackerBench(B, 0)
for _ in range(3):
    tb.step()
torch.cuda.synchronize()
prev, cur = tb.L[(tb.k - 2) % 2], tb.L[(tb.k - 1) % 2]
out = torch.empty_like(tb.cur_left)
for cap in (1, 2, 3, 5, -1):
    def run():
        tb.ctx.klt_track_batch_dev(B, prev.data_ptr(), cur.data_ptr(), bench.NPTS, tb.pts_left.data_ptr(), out.data_ptr(),
                                   tb.st1.data_ptr(), 0, False, cap)
    # cap 0 is clamped by the ABI? (max_iter_override <= 0 means "use the default"): use hv params instead
    for _ in range(3):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    print(f"iteration cap {cap:3d}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per LK call (B = {B}, {bench.NPTS} points)")
