"""CPU check of the structure `structured_S` (hybvio_amd/csrc/ekf_device.hpp) relies on, on the ORACLE's Jacobians -- i.e. on
prepareVisualUpdate itself (triangulation.cpp:940-980), independently of any device code: on the active columns
    H = Dp + O4 F4
with Dp the observation's own-pose block (7 columns), O4 = [dip R | -feature velocity] (rows x 4) and F4 = [d pf / d column; unit row of
the time-shift column]. Consequence tested here: the block (rows of the observations NOT taken at pose k) x (the 7 columns of pose k)
is O F_k, so its rank is at most 3, and the time-shift column plus the feature velocities is O f_t: it lies in the column space of O."""
import numpy as np
import pytest

from hybvio_amd import synth
from oracle import orc as oracle

SFT, CAM, POSE = 19, 20, 7


def _pose_cols(pidx):
    # getPosOriIndices (triangulation.cpp:989-998): pose 0 is the current pose (POS 0..2, ORI 6..9), pose i >= 1 trail slot i - 1
    if pidx == 0:
        return [0, 1, 2, 6, 7, 8, 9]
    b = CAM + POSE * (pidx - 1)
    return list(range(b, b + 7))


@pytest.mark.parametrize("npose,stereo", [(6, True), (10, True), (16, True), (21, True), (9, False)])
def test_prepare_visual_update_jacobian_is_block_diagonal_plus_rank_four(npose, stereo):
    rng = np.random.default_rng(40 + npose)
    B, trail = 6, 20
    T1, T2, means, idx, feat = synth.visual_tracks(rng, B, trail, npose, stereo)
    vel = rng.normal(size=feat.shape) * 0.1
    par = oracle.tri_default_params()
    checked = 0
    for b in range(B):
        st, ps, pf, H, f = oracle.visual_track_prepare(par, means[b], idx[b], T1, T2 if stereo else None, feat[b], vel[b])
        if (st, ps) != (0, 0):
            continue
        nt = npose * (2 if stereo else 1)
        assert H.shape == (2 * nt, len(means[b]))
        active = sorted(set(c for k in range(npose) for c in _pose_cols(int(idx[b][k]))) | {SFT})
        assert np.all(H[:, [c for c in range(H.shape[1]) if c not in active]] == 0.0)       # nothing outside the 7 n + 1 active columns
        scale = np.abs(H).max()
        O_rows = []
        for k in range(npose):
            cols = _pose_cols(int(idx[b][k]))
            others = [i for i in range(nt) if i % npose != k]
            rows = [2 * i + s for i in others for s in (0, 1)]
            blk = H[np.ix_(rows, cols)]                                                    # = O[rows] F_k: rank <= 3
            sv = np.linalg.svd(blk, compute_uv=False)
            assert sv[3] < 1e-11 * max(sv[0], scale), (npose, b, k, sv[:5])
            if k == 0:
                O_rows = (rows, np.linalg.svd(blk, full_matrices=False)[0][:, :3])          # an orthonormal basis of span(O[rows])
        # the time-shift column is O f_t - velocity (triangulation.cpp:965-967): adding the velocities back leaves a vector in span(O)
        rows, Q = O_rows
        col = H[rows, SFT] + vel[b].reshape(-1)[rows]
        resid = col - Q @ (Q.T @ col)
        assert np.abs(resid).max() < 1e-9 * max(1.0, np.abs(col).max()), (npose, b, np.abs(resid).max())
        checked += 1
    assert checked >= 3
