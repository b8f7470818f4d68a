"""GPU parity tests of the GFTT feature detector through the C ABI against
oracle/gftt_oracle.c: key points (position AND binary32 response) and the final corner lists must be
bit-identical -- the kernel evaluates the oracle's float sequence without FMA contraction."""
import numpy as np
import pytest

from hybvio_amd import capi, synth

pytestmark = pytest.mark.gpu


def _ctx(w, h, **kw):
    return capi.Context(width=w, height=h, **kw)


def _device_keypoints(ctx, slots, params=None):
    import torch
    nk = ctx.gftt_keypoint_count(params)
    sl = torch.tensor(slots, dtype=torch.int32, device="cuda")
    kp = torch.full((len(slots), max(nk, 1), 3), -7.0, dtype=torch.float32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.gftt_keypoints_batch_dev(len(slots), sl.data_ptr(), kp.data_ptr(), params)
    torch.cuda.synchronize()
    return kp.cpu().numpy()[:, :nk]


@pytest.mark.parametrize("kernel", ["marching", "tiled"])
@pytest.mark.parametrize("shape", [(480, 752), (720, 1280), (479, 641), (97, 130), (70, 101), (64, 64)])
@pytest.mark.parametrize("min_distance", [50.0, 20.0, 8.0])
def test_keypoints_bit_exact(oracle, shape, min_distance, kernel, monkeypatch):
    # two kernels serve the detector: the register-marching one (many images) and the LDS-tiled one (a few); the library
    # picks by image count, the test forces each
    monkeypatch.setenv("HV_GFTT_TILED", "1" if kernel == "tiled" else "0")
    rng = np.random.default_rng(shape[0] + int(min_distance))
    imgs = [rng.integers(0, 256, shape, dtype=np.uint8),
            synth.stereo_sequence(3, shape[1], shape[0], 1)[0][0],
            np.full(shape, 90, np.uint8)]                              # flat: every block reports "no corner"
    imgs[2][5:9, 5:9] = 255                                            # ... except around one bright square
    gp = capi.gftt_default_params(gfttMinDistance=min_distance)
    bs = oracle.gftt_block_size(min_distance)
    with _ctx(shape[1], shape[0], pool_size=4) as ctx:
        slots = []
        for im in imgs:
            s = ctx.acquire(); ctx.build(s, im); slots.append(s)
        kp = _device_keypoints(ctx, slots, gp)
        for i, im in enumerate(imgs):
            ref = oracle.gftt_collect_max(oracle.corner_min_eigen_val(im), bs, 1e-3)
            assert kp[i].shape == ref.shape
            assert np.array_equal(kp[i], ref), (i, np.nonzero((kp[i] != ref).any(1))[0][:5])


@pytest.mark.parametrize("block_size", [5, 7])
@pytest.mark.parametrize("shape,min_distance", [((480, 752), 50.0), ((97, 130), 20.0), ((70, 101), 8.0), ((64, 64), 50.0), ((67, 35), 8.0)])
def test_keypoints_bit_exact_with_other_box_sizes(oracle, shape, min_distance, block_size):
    """gfttBlockSize 5 / 7 (parameter_definitions.c: odometry gfttBlockSize, cv::cornerMinEigenVal's blockSize): the plain box kernel
    against the oracle, incl. images whose ragged right / bottom strip is narrower than the box (products mirrored at the border)."""
    rng = np.random.default_rng(shape[0] + block_size)
    imgs = [rng.integers(0, 256, shape, dtype=np.uint8), synth.stereo_sequence(5, shape[1], shape[0], 1)[0][0]]
    gp = capi.gftt_default_params(gfttMinDistance=min_distance, gfttBlockSize=block_size)
    bs = oracle.gftt_block_size(min_distance)
    with _ctx(shape[1], shape[0], pool_size=2) as ctx:
        slots = []
        for im in imgs:
            s = ctx.acquire(); ctx.build(s, im); slots.append(s)
        kp = _device_keypoints(ctx, slots, gp)
        for i, im in enumerate(imgs):
            ref = oracle.gftt_collect_max(oracle.corner_min_eigen_val(im, block_size), bs, 1e-3)
            assert np.array_equal(kp[i], ref), (i, np.nonzero((kp[i] != ref).any(1))[0][:5])


def test_detect_matches_reference_flow_incl_zero_prefix_and_mask(oracle):
    left = synth.stereo_sequence(21, 752, 480, 2)[0]
    with _ctx(752, 480) as ctx:
        s = ctx.acquire(); ctx.build(s, left[0])
        raw = ctx.gftt_detect(s)                                        # maskRadius 0: no filter, no cap
        assert np.array_equal(raw, oracle.gftt_detect(left[0], mask_radius=0))
        first = ctx.gftt_detect(s, mask_radius=50)
        assert np.array_equal(first, oracle.gftt_detect(left[0], mask_radius=50))
        assert first[0].tolist() == [0.0, 0.0]                          # feature_detector.cpp:629-631 quirk
        # next frame: existing tracks mask the detector (tracker.cpp:683-700 via Image::findKeypoints)
        s2 = ctx.acquire(); ctx.build(s2, left[1])
        prev = first[1:120]
        for r, cap in ((50, 200), (30, 200), (50, 25)):
            gp = capi.gftt_default_params(maxTracks=cap)
            got = ctx.gftt_detect(s2, prev=prev, mask_radius=r, params=gp)
            assert np.array_equal(got, oracle.gftt_detect(left[1], prev=prev, mask_radius=r, max_tracks=cap))
            assert len(got) <= cap
        assert np.array_equal(capi.apply_min_distance(raw, prev, 50, 200), oracle.apply_min_distance(raw, prev, 50, 200))


def test_detector_on_the_batched_device_path_and_errors(oracle, seq752):
    import torch
    left, right = seq752[0], seq752[1]
    B = 6
    with _ctx(752, 480, pool_size=B) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        frames = torch.from_numpy(np.stack([left[i % len(left)] if i % 2 == 0 else right[i % len(right)] for i in range(B)])).cuda()
        slots = [ctx.acquire() for _ in range(B)]
        sl = torch.tensor(slots, dtype=torch.int32, device="cuda")
        ctx.build_batch_dev(B, sl.data_ptr(), frames.data_ptr(), 752 * 480, 752)    # level 0 is the caller's buffer, used in place
        kp = _device_keypoints(ctx, slots)
        for i in range(B):
            ref = oracle.gftt_collect_max(oracle.corner_min_eigen_val(frames[i].cpu().numpy()), 32, 1e-3)
            assert np.array_equal(kp[i], ref)
        # box sizes beyond 7 (and even ones) are not served
        gp = capi.gftt_default_params(gfttBlockSize=9)
        out = np.zeros((2 * ctx.gftt_keypoint_count(), 2), np.float32)
        n = capi.C.c_int(0)
        rc = capi.lib().hv_gftt_detect(ctx._h, capi.C.byref(gp), slots[0], None, 0, 0, out.ctypes.data_as(capi.f32p), len(out), capi.C.byref(n))
        assert rc == -2                                                     # HV_ERR_UNSUPPORTED
        gp = capi.gftt_default_params()
        rc = capi.lib().hv_gftt_detect(ctx._h, capi.C.byref(gp), slots[0], None, 0, 0, out.ctypes.data_as(capi.f32p), 3, capi.C.byref(n))
        assert rc == -1                                                     # capacity too small


def test_device_detector_against_committed_golden_fixture():
    """No oracle involved: the HIP detector against tests/golden/gftt_golden.npz."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gftt_golden.npz"))
    img = g["img"]
    with _ctx(img.shape[1], img.shape[0]) as ctx:
        s = ctx.acquire(); ctx.build(s, img)
        for bs, md in ((8, 8.0), (16, 20.0), (32, 50.0)):
            gp = capi.gftt_default_params(gfttMinDistance=md, maxTracks=30)
            assert np.array_equal(_device_keypoints(ctx, [s], gp)[0], g[f"kp{bs}"])
            assert np.array_equal(ctx.gftt_detect(s, params=gp), g[f"raw{bs}"])
            assert np.array_equal(ctx.gftt_detect(s, prev=g["prev"], mask_radius=int(md), params=gp), g[f"masked{bs}"])
