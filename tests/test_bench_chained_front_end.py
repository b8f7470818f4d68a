"""CPU check of bench.py's `c3_chained` front end (VisualEkfBench._regenerate_tracks): the torch code that re-projects every track from
the CURRENT device mean each frame must produce observations the reference's triangulation accepts -- the oracle triangulates the
regenerated tracks from the same mean (backend.cpp:1063-1148 conventions: camera pose = imuToCamera applied to the trail pose, second
camera displaced by its baseline) and re-projects them to within the front end's own noise."""
import types

import numpy as np
import torch

import bench


def test_regenerated_tracks_triangulate_from_the_same_mean(oracle):
    rng = np.random.default_rng(1)
    B = 12
    T1, T2, means, lens, idx, feat, vel, y = bench.make_visual_frame_realistic(rng, B)
    o = types.SimpleNamespace(torch=torch)
    to = lambda a, dt: torch.from_numpy(np.array(a, dt, order="C"))
    o.idx, o.feat, o.vel, o.y, o.lens = to(idx, np.int32), to(feat, np.float64), to(vel, np.float64), to(y, np.float64), to(lens, np.int32)
    for name in ("_init_chain", "_regenerate_tracks", "_scatter_cam"):
        setattr(o, name, types.MethodType(getattr(bench.VisualEkfBench, name), o))
    o._init_chain(rng, torch.device("cpu"), T1, T2)
    # a DIFFERENT state than the one the tracks were generated for: every pose moved and rotated a little
    m2 = means.copy()
    m2[:, 0:3] += 0.05 * rng.normal(size=(B, 3))
    for k in range(20):
        m2[:, 20 + 7 * k:23 + 7 * k] += 0.05 * rng.normal(size=(B, 3))
        q = m2[:, 23 + 7 * k:27 + 7 * k] + 0.01 * rng.normal(size=(B, 4))
        m2[:, 23 + 7 * k:27 + 7 * k] = q / np.linalg.norm(q, axis=1, keepdims=True)
    o._regenerate_tracks(to(m2, np.float64))
    par = oracle.tri_default_params()
    ok = worst = 0
    for k in range(bench.VISITS):
        for b in range(B):
            n = int(lens[k, b])
            f_new = o.feat[k, b, :2 * n].numpy()
            assert float(o.feat[k, b, 2 * n:].abs().max()) == 0.0 if 2 * n < o.feat.shape[2] else True      # padding stays zero
            st, ps, pf, H, f = oracle.visual_track_prepare(par, m2[b], idx[k, b, :n], T1, T2, f_new, vel[k, b, :2 * n])
            if (st, ps) == (0, 0):
                ok += 1
                worst = max(worst, float(np.abs(f - f_new.reshape(-1)).max()))
    assert ok >= 0.9 * bench.VISITS * B, ok                 # (a few random points fall behind a camera of the moved trail)
    assert worst < 2e-3, worst                              # re-projection of the triangulated point vs the regenerated observations (noise 1e-4)
    # the measurement keeps the offsets the start state's y had (observation noise + the gross errors of the outliers)
    assert float((o.y - o.feat.reshape(bench.VISITS, B, -1) - o.c_yoff).abs().max()) < 1e-12
