"""GPU tests of the vo_pipeline drop-in classes (reference method names and return
conventions) on top of the C-ABI."""
import os

import numpy as np
import pytest

from visual_odometry_amd import params, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pose_estimation_klt_signature_and_parity(oracle, ctx):
    from visual_odometry_amd.vo_pipeline.poseEstimation import PoseEstimation
    img0, img1, _ = synth.frame_pair((240, 320), seed=21)
    pts = synth.uniform_keypoints((240, 320), 300, seed=5).astype(np.float64)   # the reference casts to float32
    out, st = PoseEstimation.KLT(img0, img1, pts)
    assert out.shape == (300, 2) and out.dtype == np.float32 and st.shape == (300, 1) and st.dtype == np.uint8
    ro, rs, _ = oracle.klt(img0, img1, np.float32(pts), win=params.KLT_RADIUS, max_level=params.KLT_NUM_PYRAMIDS,
                           max_iters=params.KLT_N_ITERS, eps=0.001, min_eig=params.KLT_MIN_EIGEN_THRESHOLD)
    assert (st == rs).mean() >= 0.995
    ok = (st.ravel() == 1) & (rs.ravel() == 1)
    assert np.abs(out[ok] - ro[ok]).max() <= 0.05
    o, s = PoseEstimation.KLT(img0, img1, np.zeros((0, 2)))
    assert o.shape == (0, 2) and s.shape == (0, 1)
    pe = PoseEstimation(synth.K_KITTI, first_kps=np.float32(pts))
    p1 = pe.match_key_points(None, img0, img1)
    assert p1.shape[0] == int((st == 1).sum()) and pe.prev_kpts is p1


def test_trackpoints_forward_backward(ctx):
    from visual_odometry_amd.vo_pipeline.trackPoints import TrackPoints
    img0, img1, A = synth.frame_pair((240, 320), seed=22)
    pts = synth.uniform_keypoints((240, 320), 200, seed=6, margin=25)
    tp = TrackPoints(params.KLT_RADIUS, params.KLT_N_ITERS, params.KLT_LAMBDA)
    kp1, mask = tp.trackKLT(img0, img1, pts)
    assert kp1.shape == (200, 2) and kp1.dtype == np.float64 and mask.shape == (200, 1) and mask.dtype == bool
    assert mask.mean() > 0.9
    truth = pts @ A[:, :2].T + A[:, 2]
    m = mask.ravel()
    assert np.median(np.linalg.norm(kp1[m] - truth[m], axis=1)) < 0.05
    assert (kp1[~m] == 0).all()


def test_harris_dropin(ctx):
    from visual_odometry_amd.vo_pipeline.featureExtraction import ExtractorType, FeatureExtractor
    img = np.full((80, 100), 40, np.uint8)
    img[20:50, 30:70] = 200
    c = FeatureExtractor.harris(img)
    assert c.dtype == np.float32 and c.ndim == 2 and c.shape[1] == 2 and c.shape[0] >= 4
    corners = np.array([(30, 20), (69, 20), (30, 49), (69, 49)], np.float32)
    for cx in corners:
        assert np.abs(c - cx).sum(1).min() <= 2
    kps, des = FeatureExtractor(ExtractorType.SIFT).get_kp(img)          # the rectangle's corners are SIFT keypoints too
    assert des.shape == (len(kps), 128) and all(hasattr(k, "pt") for k in kps)


def test_pnp_recovers_planted_pose(ctx):
    from visual_odometry_amd.vo_pipeline.poseEstimation import AlgoMethod, PoseEstimation
    sc = synth.two_view_scene(n=600, K=synth.K_KITTI, shape=synth.KITTI_SHAPE, inlier_frac=0.7, noise=0.3, seed=31)
    pe = PoseEstimation(synth.K_KITTI, algo_method_type=AlgoMethod.P3P)
    M, inliers = pe.PnP(sc["X"], sc["m2"])
    assert M.shape == (4, 4) and M.dtype == np.float64 and inliers.ndim == 2 and inliers.shape[1] == 1 and inliers.dtype == np.int32
    np.testing.assert_allclose(M[:3, :3], sc["R"], atol=2e-3)
    np.testing.assert_allclose(M[:3, 3], sc["t"], atol=3e-2)
    got = np.zeros(600, bool); got[inliers.ravel()] = True
    assert (got & ~sc["inliers"]).sum() <= 3           # outliers stay out
    assert got[sc["inliers"]].mean() > 0.9             # 1 px gate at 0.3 px noise
    with pytest.raises(AssertionError):
        pe.PnP(sc["X"][:10], sc["m2"][:9])


def test_find_fundamental_mat_recovers_planted_geometry(ctx):
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    sc = synth.two_view_scene(n=800, inlier_frac=0.75, noise=0.02, seed=33)
    F, mask = BootstrapInitializer.findFundamentalMat(sc["m1"], sc["m2"], 0.1, 0.9999, 100)
    assert F.shape == (3, 3) and F.dtype == np.float64 and mask.shape == (800, 1) and mask.dtype == np.uint8
    m = mask.ravel() == 1
    assert (m & ~sc["inliers"]).sum() <= 4
    assert m[sc["inliers"]].mean() > 0.8
    # two-view geometry after F: rotation and translation direction of the planted motion
    b = object.__new__(BootstrapInitializer)
    b.K, b.max_point_dist = sc["K"].astype(np.float32), 100
    h1 = np.hstack([sc["m1"][m], np.ones((m.sum(), 1), np.float32)])
    h2 = np.hstack([sc["m2"][m], np.ones((m.sum(), 1), np.float32)])
    T, cloud, ok = b._transform_matrix(F, h1, h2)
    np.testing.assert_allclose(T[:3, :3], sc["R"], atol=5e-3)
    tdir = sc["t"] / np.linalg.norm(sc["t"])
    assert T[:3, 3] @ tdir > 0.999


@pytest.mark.parametrize("name", ["ba_small", "ba_mid"])
def test_bundle_adjustment_fun_matches_reference(ctx, name):
    from visual_odometry_amd.vo_pipeline.bundleAdjustment import BundleAdjustment
    d = np.load(os.path.join(GOLD, name + ".npz"))
    ba = BundleAdjustment(d["K"])
    r = ba._fun(d["x"], int(d["F"]), int(d["P"]), d["ci"], d["pi"], d["obs"])
    assert r.shape == d["r"].shape and r.dtype == np.float64
    np.testing.assert_allclose(r, d["r"], rtol=0, atol=1e-8)
    cost = 0.5 * float(r @ r)
    assert abs(cost - float(d["cost"])) <= 1e-6 * float(d["cost"])
    F = int(d["F"])
    proj = ba._reproject(d["x"][6 * F:].reshape(-1, 3)[d["pi"]], d["x"][:6 * F].reshape(F, 6)[d["ci"]])
    np.testing.assert_allclose((proj - d["obs"]).ravel(), d["r"], atol=1e-8)


def test_bundle_adjustment_solve_against_scipy(oracle, ctx):
    """Same problem definition as the reference's least_squares call (frozen first cameras through the
    sparsity, ftol 1e-4): the final cost must be as low as scipy's TRF on the same residual."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    from visual_odometry_amd.vo_pipeline.bundleAdjustment import BundleAdjustment
    pb = synth.ba_problem(F=6, P=120, visibility=0.8, seed=5, obs_noise=0.3)
    F, P, ci, pi, obs, K = pb["F"], pb["P"], pb["ci"], pb["pi"], pb["obs"], pb["K"].astype(np.float32)
    rng = np.random.default_rng(0)
    cams = pb["x"][:6 * F].reshape(F, 6).copy()
    cams[2:, 3:] += rng.normal(0, 0.02, (F - 2, 3))              # perturb the free cameras
    M = np.zeros((F, 3, 4), np.float32)
    for i in range(F):
        M[i, :, :3] = Rotation.from_rotvec(cams[i, :3]).as_matrix()
        M[i, :, 3] = cams[i, 3:]
    landmarks = (pb["x"][6 * F:].reshape(P, 3) + rng.normal(0, 0.1, (P, 3))).astype(np.float32)
    ba = BundleAdjustment(K)
    kp2, ci2, pi2 = ba._preprocess_data(M, landmarks, ci, pi, obs)
    x0 = np.hstack([ba._camera_params(M).ravel(), landmarks.astype(np.float64).ravel()])
    fun = lambda x: oracle.ba_residual(x, F, P, ci2, pi2, kp2, K)[0]
    c0 = 0.5 * float(fun(x0) @ fun(x0))
    A = ba._bundle_adjustment_sparsity(F, P, ci2, pi2)
    res = least_squares(fun, x0, jac_sparsity=A, x_scale="jac", ftol=1e-4, method="trf")
    adj_M, adj_L = ba.bundle_adjustment(M, landmarks, ci, pi, obs)
    assert adj_M.shape == (F, 3, 4) and adj_L.shape == (P, 3) and adj_M.dtype == np.float64
    x1 = np.hstack([ba._camera_params(adj_M).ravel(), adj_L.ravel()])
    c1 = 0.5 * float(fun(x1) @ fun(x1))
    assert c1 < 0.5 * c0
    assert c1 <= res.cost * 1.05
    # frozen cameras (first int(F/4)+1 = 2) stay put
    np.testing.assert_allclose(adj_M[:2], M[:2].astype(np.float64), atol=1e-6)


def test_pose_graph_optimizer_motion_only(ctx):
    from visual_odometry_amd.vo_pipeline.poseOptimization import PoseGraphOptimizer
    sc = synth.two_view_scene(n=300, K=synth.K_KITTI, shape=synth.KITTI_SHAPE, inlier_frac=1.0, noise=0.2, seed=41)

    class FS:
        pass
    fs = FS(); fs.idx = 3
    T = np.eye(4); T[:3, :3] = synth.rodrigues([0.012, -0.037, 0.01]); T[:3, 3] = sc["t"] + [0.03, -0.02, 0.04]
    fs.pose = T

    class Tracks:
        def at_frame(self, idx):
            assert idx == 3
            return sc["m2"], np.arange(300, dtype=np.int32), sc["X"]
    assert PoseGraphOptimizer(None).optimize() == 0          # the reference's stub signature still works
    opt = PoseGraphOptimizer([fs], K=synth.K_KITTI, keypoint_trajectories=Tracks())
    assert opt.optimize() == 1
    np.testing.assert_allclose(fs.pose[:3, :3], sc["R"], atol=1e-3)
    np.testing.assert_allclose(fs.pose[:3, 3], sc["t"], atol=2e-2)


def test_two_view_geometry_matches_reference(ctx):
    """GPU triangulation and the cheirality selection built on it against the reference's own
    outputs (tests/golden/two_view.npz, written by tools/gen_golden.py from the reference functions)."""
    from visual_odometry_amd import hip
    from visual_odometry_amd.vo_pipeline import bootstrap
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "two_view.npz"))
    tri = bootstrap.BootstrapInitializer._linear_triangulation(d["p1"], d["p2"], d["M1"], d["M2"])
    assert tri.dtype == np.float32 and tri.shape == d["tri"].shape
    np.testing.assert_allclose(tri, d["tri"], rtol=1e-5, atol=1e-5)
    b = object.__new__(bootstrap.BootstrapInitializer)
    b.K, b.max_point_dist = d["K"], float(d["max_point_dist"])
    T, cloud, mask = b._transform_matrix(d["F"], d["p1"], d["p2"])
    np.testing.assert_allclose(T, d["T"], atol=1e-6)
    np.testing.assert_allclose(cloud, d["cloud"], rtol=1e-5, atol=1e-4)
    np.testing.assert_array_equal(mask, d["mask"])
    np.testing.assert_allclose(T[:3, :3], d["R_true"], atol=1e-3)
    np.testing.assert_allclose(T[:3, 3], d["t_true"], atol=1e-3)
    # device-resident inputs, float64 projection matrices, empty input
    import torch
    t1, t2 = torch.from_numpy(d["p1"]).cuda(), torch.from_numpy(d["p2"]).cuda()
    tri_d = hip.triangulate(ctx, t1, t2, d["M1"], d["M2"]).cpu().numpy()
    np.testing.assert_array_equal(tri_d, tri)
    from oracle import solvers_np
    M1d = d["M1"].astype(np.float64)
    ref = solvers_np.linear_triangulation(d["p1"], d["p2"], M1d, d["M2"])
    np.testing.assert_allclose(hip.triangulate(ctx, d["p1"], d["p2"], M1d, d["M2"]), ref, rtol=1e-5, atol=1e-5)
    assert hip.triangulate(ctx, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), d["M1"], d["M2"]).shape == (0, 4)
    with pytest.raises(ValueError):
        hip.triangulate(ctx, d["p1"], d["p2"][:-1], d["M1"], d["M2"])


def test_cheirality_vote_matches_oracle(ctx):
    """All four motion hypotheses in one launch: clouds and in-front counts against the numpy restatement
    (per-hypothesis reference triangulation + the reference's two np.sum counts), on the golden pair set and
    on a noisy scene with outliers; host and device memory; the winner reproduces the reference's T and cloud."""
    import torch
    from oracle import solvers_np
    from visual_odometry_amd import hip
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    d = np.load(os.path.join(GOLD, "two_view.npz"))
    cases = [(d["K"], d["F"], d["p1"], d["p2"], 1e-5, 1e-4)]
    sc = synth.two_view_scene(n=700, inlier_frac=0.8, noise=0.3, seed=17)
    Kf = sc["K"].astype(np.float32)
    Fm = sc["F"]
    h1 = np.hstack([sc["m1"], np.ones((700, 1), np.float32)]).astype(np.float32)
    h2 = np.hstack([sc["m2"], np.ones((700, 1), np.float32)]).astype(np.float32)
    # 20 % of these pairs are outliers whose rays nearly miss each other: the null vector of such a pair is
    # ill-conditioned, and the Jacobi eigenvector of A^T A and numpy's SVD of A part ways in the fourth digit
    cases.append((Kf, Fm, h1, h2, 2e-3, 1e-3))
    for K, F, p1, p2, rtol, atol in cases:
        motions = BootstrapInitializer.motion_hypotheses(K.T @ F @ K)
        assert motions.shape == (4, 3, 4)
        for m in motions:
            np.testing.assert_allclose(m[:, :3] @ m[:, :3].T, np.eye(3), atol=1e-9)
            assert np.linalg.det(m[:, :3]) > 0
        M1 = K @ np.eye(3, 4, dtype=np.float32)
        M2 = np.stack([K @ m for m in motions])
        ref_clouds, ref_votes = solvers_np.cheirality_votes(p1, p2, M1, M2, motions)
        clouds, votes = hip.cheirality_vote(ctx, p1, p2, M1, M2, motions)
        assert clouds.dtype == np.float32 and votes.dtype == np.int32
        np.testing.assert_array_equal(votes, ref_votes)
        np.testing.assert_allclose(clouds, ref_clouds, rtol=rtol, atol=atol)
        rel = np.abs(clouds - ref_clouds)[..., :3] / np.maximum(np.abs(ref_clouds[..., :3]), 1.0)
        assert np.median(rel) < 1e-6
        dc, dv = hip.cheirality_vote(ctx, torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda(), M1, M2, motions)
        np.testing.assert_array_equal(dv.cpu().numpy(), votes)
        np.testing.assert_array_equal(dc.cpu().numpy(), clouds)
        # single-hypothesis call equals the plain triangulation
        one, _ = hip.cheirality_vote(ctx, p1, p2, M1, M2[1:2], motions[1:2])
        np.testing.assert_array_equal(one[0], hip.triangulate(ctx, p1, p2, M1, M2[1]))
    e, v = hip.cheirality_vote(ctx, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), M1, M2, motions)
    assert e.shape == (4, 0, 4) and not v.any()
    with pytest.raises(ValueError):
        hip.cheirality_vote(ctx, p1, p2, M1, np.zeros((9, 3, 4)), np.zeros((9, 3, 4)))


def test_klt_pyramid_reuse_is_transparent(ctx):
    """reuse_pyramids keeps the previous call's next-frame pyramid only for the identical array object;
    results equal the rebuild-everything path, and a different or modified-then-copied frame is rebuilt."""
    from visual_odometry_amd.vo_pipeline.poseEstimation import PoseEstimation
    f = [synth.frame_pair((240, 320), seed=s)[0] for s in (1, 2, 3)]
    f[1] = synth.frame_pair((240, 320), seed=1)[1]
    f[2] = np.roll(f[1], 2, axis=1)
    pts = synth.uniform_keypoints((240, 320), 300, seed=4, margin=25)
    plain = [PoseEstimation.KLT(f[i], f[i + 1], pts) for i in range(2)]
    chained = [PoseEstimation.KLT(f[i], f[i + 1], pts, reuse_pyramids=True) for i in range(2)]   # second call reuses f[1]
    for (a, sa), (b, sb) in zip(plain, chained):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(sa, sb)
    # an equal copy is another object: rebuilt, same answer; so is a frame whose predecessor was something else
    c, sc = PoseEstimation.KLT(f[1].copy(), f[2], pts, reuse_pyramids=True)
    np.testing.assert_array_equal(c, plain[1][0])
    d, sd = PoseEstimation.KLT(f[0], f[1], pts, reuse_pyramids=True)
    np.testing.assert_array_equal(d, plain[0][0])
    np.testing.assert_array_equal(sd, plain[0][1])
