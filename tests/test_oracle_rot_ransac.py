"""CPU checks of the rotation-RANSAC oracle (oracle/rot_ransac_oracle.c). The reference holds no
test for RotRansac ("parity unpinned": cv::SVD is absent), so the restatement is checked against the generator's standard
known answer, an independent numpy SVD evaluation of Kabsch, and the control flow the reference code implies."""
import numpy as np
import pytest


def _rot(rng, angle=None):
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = rng.uniform(0, 0.2) if angle is None else angle
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _kabsch_numpy(p1, p2):
    H = (p1[:, :, None] * p2[:, None, :]).sum(0)                 # sum p1 p2^T
    U, S, Vt = np.linalg.svd(H)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        R = Vt.T @ np.diag([1, 1, -1]) @ U.T                     # rot_ransac.cpp:155-158
    return R


def test_mt19937_known_answers(oracle):
    assert oracle.mt19937_draws(5489, 1, skip=9999)[0] == 4123659995          # the C++ standard's check value
    assert oracle.mt19937_draws(5489, 3).tolist() == [3499211612, 581869302, 3890346734]
    a = oracle.mt19937_draws(4649, 700)                                        # ransacRngSeed, across the 624-word refill
    assert np.array_equal(a[100:700], oracle.mt19937_draws(4649, 600, skip=100))
    bg = np.random.MT19937(); bg._legacy_seeding(4649)                         # numpy's init_genrand seeding = std::mt19937(seed)
    assert np.array_equal(a, bg.random_raw(700).astype(np.uint32))


@pytest.mark.parametrize("n", [2, 3, 10, 200])
def test_solve_rotation_equals_svd_kabsch(oracle, n):
    rng = np.random.default_rng(n)
    for trial in range(20):
        R = _rot(rng, rng.uniform(0, 3.0))
        p1 = rng.normal(size=(n, 3)); p1 /= np.linalg.norm(p1, axis=1, keepdims=True)
        p2 = p1 @ R.T + (0.0 if trial % 2 == 0 else 1e-3) * rng.normal(size=(n, 3))
        p1f, p2f = p1.astype(np.float32), p2.astype(np.float32)
        got = oracle.solve_rotation(p1f, p2f, np.arange(n))
        want = _kabsch_numpy(p1f.astype(np.float64), p2f.astype(np.float64))
        assert np.abs(got - want).max() < 2e-6, (trial, np.abs(got - want).max())
        assert abs(np.linalg.det(got.astype(np.float64)) - 1) < 1e-5 and np.abs(got @ got.T - np.eye(3)).max() < 1e-5


def test_solve_rotation_reflection_and_degenerate_inputs(oracle):
    rng = np.random.default_rng(1)
    # coplanar points mirrored through their plane: the best orthogonal map is a reflection, the answer must be a rotation
    p1 = np.c_[rng.normal(size=(30, 2)), np.zeros(30)]; p2 = p1.copy(); p2[:, 0] = -p2[:, 0]
    R = oracle.solve_rotation(p1, p2, np.arange(30)).astype(np.float64)
    assert abs(np.linalg.det(R) - 1) < 1e-5 and np.abs(R - _kabsch_numpy(p1, p2)).max() < 1e-5
    # a single direction used twice (rank 1): still a proper rotation that maps it correctly
    a, b = np.array([[0.1, 0.2, 0.97]]), np.array([[0.15, 0.18, 0.97]])
    a /= np.linalg.norm(a); b /= np.linalg.norm(b)
    R1 = oracle.solve_rotation(np.r_[a, a], np.r_[b, b], [0, 1]).astype(np.float64)
    assert abs(np.linalg.det(R1) - 1) < 1e-5 and np.abs(R1 @ a[0] - b[0]).max() < 1e-6


def _scene(oracle, rng, n, outliers, cam_kind="pinhole"):
    w, h = 752, 480
    if cam_kind == "pinhole":
        cam = oracle.Camera("pinhole", 458.654, 457.296, 367.215, 248.375, coeffs=[-0.28340811, 0.07395907, 0.0])
    else:
        cam = oracle.Camera("fisheye", 280.0, 280.0, 376.0, 240.0, coeffs=[0.0035, 0.0007, -0.002, 0.0002], max_valid_fov_deg=170.0)
    R = _rot(rng, 0.03)
    c1 = rng.uniform([60, 60], [w - 60, h - 60], (n, 2)).astype(np.float32)
    c2 = np.zeros_like(c1)
    for i in range(n):
        ok, ray = cam.pixel_to_ray(*c1[i])
        ok2, pix = cam.ray_to_pixel(R @ ray)
        c2[i] = pix + rng.normal(size=2) * 0.3
    bad = rng.choice(n, outliers, replace=False)
    c2[bad] += rng.uniform(8, 40, (outliers, 2)).astype(np.float32) * rng.choice([-1, 1], (outliers, 2))
    return cam, c1, c2, R, set(bad.tolist())


@pytest.mark.parametrize("cam_kind", ["pinhole", "fisheye"])
def test_fit_finds_the_rotation_and_the_outliers(oracle, cam_kind):
    rng = np.random.default_rng(3)
    cam, c1, c2, R, bad = _scene(oracle, rng, 200, 40, cam_kind)
    thr = float(np.float32((4.0 * 480 / 720.0) ** 2))                      # ransac_pipeline.cpp:91-93
    st, Rf, best, used = oracle.rot_ransac_fit(c1, c2, cam, cam, oracle.mt19937_draws(4649, 200), thr)
    assert used == 200 and 150 <= best <= 160
    assert set(np.nonzero(st == 3)[0].tolist()) == bad and set(np.unique(st).tolist()) == {0, 3}
    assert np.abs(Rf.astype(np.float64) - R).max() < 2e-3


def test_fit_control_flow_draw_consumption_and_first_maximum(oracle):
    rng = np.random.default_rng(4)
    cam, c1, c2, R, _ = _scene(oracle, rng, 50, 0)
    thr = 7.1
    # all points inliers: the loop stops at the first hypothesis that is not a repeated index (rot_ransac.cpp:84,104)
    draws = np.array([7, 7 + 50] + [3, 9] + [1, 2] * 99, np.uint32)        # 7 % 50 == 57 % 50 -> skipped, consumes 2 draws
    st, Rf, best, used = oracle.rot_ransac_fit(c1, c2, cam, cam, draws, thr)
    assert used == 4 and best == 50 and (st == 0).all()
    # only "continue" iterations: no hypothesis is ever evaluated, bestInds stays {0, 1}, bestInlierCount 0
    same = np.repeat(np.arange(100, dtype=np.uint32), 2)
    st2, R2, best2, used2 = oracle.rot_ransac_fit(c1, c2, cam, cam, same, thr)
    assert used2 == 200 and best2 == 0 and np.array_equal(R2, oracle_refit(oracle, c1, c2, cam, thr, [0, 1]))
    # n = 2 (the smallest the pipeline passes, ransac_pipeline.cpp:209)
    st3, _, best3, _ = oracle.rot_ransac_fit(c1[:2], c2[:2], cam, cam, oracle.mt19937_draws(1, 200), thr)
    assert best3 == 2 and (st3 == 0).all()


def oracle_refit(oracle, c1, c2, cam, thr, inds):
    """rot_ransac.cpp:107-118 done by hand: rotation of the pair, its inliers, rotation of the inliers."""
    p1 = np.array([cam.pixel_to_ray(*c)[1] for c in c1], np.float32)
    p2 = np.array([cam.pixel_to_ray(*c)[1] for c in c2], np.float32)
    R = oracle.solve_rotation(p1, p2, inds)

    def inl(R):
        out = []
        for i in range(len(c1)):
            q = np.zeros(3, np.float32)
            for r in range(3):
                s = np.float32(0)
                for k in range(3):
                    s = np.float32(s + np.float32(R[r, k] * p1[i, k]))
                q[r] = s
            ok, pix = cam.ray_to_pixel(q.astype(np.float64))
            d = (c2[i] - pix.astype(np.float32)).astype(np.float64)
            if ok and d @ d <= thr:
                out.append(i)
        return out
    ii = inl(R)
    return oracle.solve_rotation(p1, p2, ii) if len(ii) >= 2 else R
