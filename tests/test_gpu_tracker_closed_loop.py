"""Closed-loop tracker parity (north star: "tracked-feature IDs/status bit-exact"): a miniature of
TrackerImplementation::add (src/tracker/tracker.cpp:178-239, 378-559, 675-703) -- detect, temporal LK,
stereo LK, failure merge, track bookkeeping with the reference's ID rule (nextTrackId = frameNum * maxTracks
+ 1, tracker.cpp:199) and re-detection masked by the live tracks -- driven once through the HIP library
and once through the CPU oracle on the same synthetic stereo sequence. Because every frame's inputs are the
previous frame's outputs, any single differing status or sub-pixel position would fork the two runs: IDs,
statuses and positions must stay identical for the whole sequence. (RANSAC, the epipolar check and the
flow predictor are host code outside SURVEY.md section 8 and are left out of both runs.)"""
import numpy as np
import pytest

from hybvio_amd import capi, synth

pytestmark = pytest.mark.gpu
MAX_TRACKS, MIN_DIST = 120, 20


class HipBackend:
    def __init__(self, w, h):
        self.ctx = capi.Context(width=w, height=h, max_tracks=4 * MAX_TRACKS, pool_size=8)
        self.gp = capi.gftt_default_params(gfttMinDistance=float(MIN_DIST), maxTracks=MAX_TRACKS)

    def build(self, img):
        s = self.ctx.acquire(); self.ctx.build(s, img); return s

    def release(self, s):
        self.ctx.release(s)

    def flow(self, prev, cur, pts, guess=None):
        return self.ctx.optical_flow_compute(prev, cur, pts, corners=guess)

    def detect(self, handle, mask, r):
        return self.ctx.gftt_detect(handle, prev=mask, mask_radius=r, params=self.gp)


class OracleBackend:
    def __init__(self, oracle):
        self.o = oracle

    def build(self, img):
        return (self.o.Pyramid(img), img)

    def release(self, s):
        pass

    def flow(self, prev, cur, pts, guess=None):
        return self.o.optical_flow_compute(prev[0], cur[0], pts, corners=guess)

    def detect(self, handle, mask, r):
        return self.o.gftt_detect(handle[1], prev=mask, mask_radius=r, min_distance=float(MIN_DIST), max_tracks=MAX_TRACKS)


def run_tracker(be, left, right):
    """Returns per frame: (ids, left points, right points, per-track status of the temporal+stereo merge)."""
    log, tracks = [], []                       # track = [id, (x, y) left, (x, y) right]
    prev_l = prev_r = None
    for frame in range(len(left)):
        cur_l, cur_r = be.build(left[frame]), be.build(right[frame])
        next_id = frame * MAX_TRACKS + 1                                               # tracker.cpp:199
        status = np.zeros(0, np.int32)
        if frame > 0 and len(tracks) >= 5:                                             # tracker.cpp:209
            p0 = np.array([t[1] for t in tracks], np.float32)
            xy, st = be.flow(prev_l, cur_l, p0)                                        # temporal, zero-flow start
            xr, st2 = be.flow(cur_l, cur_r, xy, guess=np.array([t[2] for t in tracks], np.float32) + (xy - p0))
            status = np.where(st2 == 2, 2, st)                                         # FAILED_FLOW merge, tracker.cpp:441-447
            tracks = [[t[0], tuple(xy[i]), tuple(xr[i])] for i, t in enumerate(tracks) if status[i] == 0]
        missing = MAX_TRACKS - len(tracks)
        if frame == 0 or missing >= MAX_TRACKS // 10:                                  # tracker.cpp:683-700
            mask = np.array([t[1] for t in tracks], np.float32).reshape(-1, 2)
            corners = be.detect(cur_l, mask, MIN_DIST)
            if len(corners):
                cr, sts = be.flow(cur_l, cur_r, corners)                               # detectFeatures: stereo LK for new corners
                for i in range(len(corners)):
                    if sts[i] == 0 and missing > 0:
                        tracks.append([next_id, tuple(corners[i]), tuple(cr[i])]); next_id += 1; missing -= 1
        log.append((np.array([t[0] for t in tracks]), np.array([t[1] for t in tracks], np.float32),
                    np.array([t[2] for t in tracks], np.float32), status.copy()))
        for s in (prev_l, prev_r):
            if s is not None:
                be.release(s)
        prev_l, prev_r = cur_l, cur_r
    return log


def test_track_ids_statuses_and_positions_identical_over_a_sequence(oracle):
    w, h, frames = 376, 240, 14
    left, right, _ = synth.stereo_sequence(77, w, h, frames)
    hip = HipBackend(w, h)
    try:
        got = run_tracker(hip, left, right)
    finally:
        hip.ctx.close()
    ref = run_tracker(OracleBackend(oracle), left, right)
    born, lost = set(), 0
    for f, ((gi, gl, gr, gs), (oi, ol, orr, os_)) in enumerate(zip(got, ref)):
        np.testing.assert_array_equal(gs, os_, err_msg=f"frame {f}: status")
        np.testing.assert_array_equal(gi, oi, err_msg=f"frame {f}: track ids")
        np.testing.assert_array_equal(gl, ol, err_msg=f"frame {f}: left positions")
        np.testing.assert_array_equal(gr, orr, err_msg=f"frame {f}: right positions")
        born |= set(gi.tolist()); lost += int((gs != 0).sum())
    ids_last = got[-1][0]
    assert len(ids_last) >= MAX_TRACKS // 2 and len(born) > len(ids_last)          # tracks were lost and re-detected on the way
    assert ids_last.min() <= MAX_TRACKS and ids_last.max() > MAX_TRACKS             # survivors from frame 0 next to later births
    assert lost > 0
