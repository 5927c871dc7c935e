"""vo_knn_match2 (through the C-ABI) against the CPU oracle: bit-exact indices, distances and ratio mask
on byte-valued (SIFT-shaped) descriptors, which take the matrix-core kernel, and on general float
descriptors, which take the float kernel; both kernels agree on byte-valued input."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


def _check(ctx, oracle, d1, d2, ratio=0.8):
    from visual_odometry_amd import hip
    idx, dist, keep = hip.knn_match2(ctx, d1, d2, ratio)
    ri, rd, rk = oracle.knn_match2(d1, d2, ratio)
    assert idx.dtype == np.int32 and dist.dtype == np.float32 and keep.dtype == np.uint8
    assert (idx == ri).all()
    assert (dist.view(np.uint32) == rd.view(np.uint32)).all()
    assert (keep == rk).all()
    return idx, dist, keep


@pytest.mark.parametrize("n1,n2", [(1, 2), (31, 33), (64, 64), (127, 129), (300, 1000), (1000, 257), (2000, 2000)])
def test_byte_descriptors_bit_exact(ctx, oracle, n1, n2):
    d1, d2, truth = synth.sift_like_descriptors(n1, n2, seed=n1 + n2)
    idx, _, keep = _check(ctx, oracle, d1, d2)
    if n1 >= 300:
        hit = truth >= 0
        assert (idx[hit, 0] == truth[hit]).mean() > 0.95 and keep[hit].mean() > 0.9


@pytest.mark.parametrize("n1,n2", [(1, 1), (5, 3), (65, 130), (500, 777), (1500, 1500)])
def test_float_descriptors_bit_exact(ctx, oracle, n1, n2):
    rng = np.random.default_rng(n1 * 7 + n2)
    d1 = rng.normal(0, 50, (n1, 128)).astype(np.float32)
    d2 = rng.normal(0, 50, (n2, 128)).astype(np.float32)
    _check(ctx, oracle, d1, d2, 0.9)


def test_extreme_bytes_and_ties(ctx, oracle):
    # all-255 against all-0 rows is the largest sum the matrix-core path has to hold exactly (2^23.99)
    d1 = np.full((3, 128), 255, np.float32)
    d2 = np.zeros((70, 128), np.float32)
    d2[40] = 255
    d2[7] = 255            # equal distance: the lower row comes first
    d2[50, :64] = 255
    idx, dist, _ = _check(ctx, oracle, d1, d2)
    assert idx[0].tolist() == [7, 40] and dist[0].tolist() == [0.0, 0.0]
    d1[1] = 0
    idx, dist, _ = _check(ctx, oracle, d1, d2)
    assert idx[1].tolist() == [0, 1]
    # one non-integer entry anywhere sends the whole call down the float path, same answers
    d2[69, 127] = 0.5
    _check(ctx, oracle, d1, d2)
    d2[69, 127] = 256.0
    _check(ctx, oracle, d1, d2)
    d2[69, 127] = -1.0
    _check(ctx, oracle, d1, d2)


def test_missing_neighbours_and_empty_sets(ctx, oracle):
    from visual_odometry_amd import hip
    d1, d2, _ = synth.sift_like_descriptors(10, 5, seed=2)
    idx, dist, keep = _check(ctx, oracle, d1, d2[:1])
    assert (idx[:, 1] == -1).all() and np.isinf(dist[:, 1]).all() and not keep.any()
    idx, dist, keep = _check(ctx, oracle, d1, d2[:0])
    assert (idx == -1).all() and np.isinf(dist).all()
    idx, dist, keep = hip.knn_match2(ctx, d1[:0], d2)
    assert idx.shape == (0, 2) and dist.shape == (0, 2) and keep.shape == (0,)
    with pytest.raises(Exception):
        hip.knn_match2(ctx, np.zeros((4, 64), np.float32), np.zeros((4, 64), np.float32))


def test_device_resident_call(ctx, oracle):
    import torch
    from visual_odometry_amd import hip
    d1, d2, _ = synth.sift_like_descriptors(3000, 3000, seed=9)
    tctx = hip.Context.from_torch(0)
    a, b = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda()
    idx, dist, keep = hip.knn_match2(tctx, a, b, 0.8)
    idx2, dist2, keep2 = hip.knn_match2(tctx, a[:100], b[:50], 0.8)     # smaller call reuses the buffers
    tctx.sync()
    ri, rd, rk = oracle.knn_match2(d1, d2, 0.8)
    assert (idx.cpu().numpy() == ri).all() and (dist.cpu().numpy() == rd).all() and (keep.cpu().numpy() == rk).all()
    ri, rd, rk = oracle.knn_match2(d1[:100], d2[:50], 0.8)
    assert (idx2.cpu().numpy() == ri).all() and (dist2.cpu().numpy() == rd).all() and (keep2.cpu().numpy() == rk).all()


def test_feature_matcher_mirror(ctx, oracle):
    from visual_odometry_amd.vo_pipeline.featureMatching import FeatureMatcher, MatcherType
    d1, d2, truth = synth.sift_like_descriptors(400, 500, seed=4)
    ri, rd, rk = oracle.knn_match2(d1, d2, 0.8)
    for mt in (MatcherType.BF, MatcherType.FLANN):
        m = FeatureMatcher(mt, k=2, matching_threshold=0.8)
        good = m.match_descriptors(d1, d2)
        rows = np.nonzero(rk)[0]
        assert [g.queryIdx for g in good] == rows.tolist()
        assert [g.trainIdx for g in good] == ri[rows, 0].tolist()
        assert [g.distance for g in good] == [float(x) for x in rd[rows, 0]]
        every = m.match_descriptors(d1, d2, filter=False)
        assert len(every) == 400 and [g.trainIdx for g in every] == ri[:, 0].tolist()
    # the reference's own filter generator works on pairs of these objects
    pairs = [(g, g) for g in every[:3]]
    assert list(m.match_filter(pairs)) == []


def test_bootstrap_from_descriptors(ctx):
    """BootstrapInitializer(features=...) takes the reference's descriptor route (bootstrap.py:188-211):
    keypoints of the second image shuffled, found again through their descriptors, planted motion recovered."""
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    sc = synth.two_view_scene(n=800, inlier_frac=1.0, noise=0.02, seed=21)
    des0, des1, truth = synth.sift_like_descriptors(800, 900, shared=1.0, seed=8)
    kp1 = np.zeros((900, 2), np.float32)
    kp1[:] = 5.0
    kp1[truth] = sc["m2"]

    class Kp:
        def __init__(self, xy):
            self.pt = (float(xy[0]), float(xy[1]))

    for kp0_in, kp1_in in ((sc["m1"], kp1), ([Kp(p) for p in sc["m1"]], [Kp(p) for p in kp1])):
        b = BootstrapInitializer(None, None, sc["K"].astype(np.float32), features=(kp0_in, des0, kp1_in, des1))
        n = b.pts1.shape[0]
        assert n > 600 and b.pts2.shape == (n, 3) and b.pts_des1.shape == (n, 128) and b.point_cloud.shape == (n, 4)
        assert b.kps.shape == (900, 2)
        np.testing.assert_allclose(b.T[:3, :3], sc["R"], atol=5e-3)
        assert b.T[:3, 3] @ (sc["t"] / np.linalg.norm(sc["t"])) > 0.999
        # descriptors travel with their points
        row = np.nonzero((sc["m1"] == b.pts1[0, :2]).all(1))[0][0]
        assert (b.pts_des1[0] == des0[row]).all() and (b.pts_des2[0] == des1[truth[row]]).all()


def test_properties_at_full_size(ctx):
    """Size-independent checks at 8000 x 8000 (beyond what the CPU checker finishes in seconds): a set matched
    against itself finds every row at distance 0; shuffling the train rows relabels the matches and leaves
    the distances bit-identical; both kernels (byte-valued and float input) obey it."""
    from visual_odometry_amd import hip
    d1, d2, _ = synth.sift_like_descriptors(8000, 8000, seed=12)
    rng = np.random.default_rng(0)
    for scale in (np.float32(1.0), np.float32(1.0 / 512)):       # 1/512 makes the entries non-integers: float kernel
        a, b = d1 * scale, d2 * scale
        idx, dist, keep = hip.knn_match2(ctx, a, a, 0.8)
        # distinct rows in the synthetic set: the nearest neighbour of a row is itself
        assert (idx[:, 0] == np.arange(8000)).all() and (dist[:, 0] == 0).all() and (dist[:, 1] > 0).all()
        assert keep.all()
        idx, dist, keep = hip.knn_match2(ctx, a, b, 0.8)
        perm = rng.permutation(8000)
        idx_p, dist_p, keep_p = hip.knn_match2(ctx, a, b[perm], 0.8)
        assert (dist_p.view(np.uint32) == dist.view(np.uint32)).all() and (keep_p == keep).all()
        untied = dist[:, 0] != dist[:, 1]
        assert untied.mean() > 0.99
        assert (perm[idx_p[untied, 0]] == idx[untied, 0]).all()
        # the first neighbour is never farther than the second, and both are real rows
        assert (dist[:, 0] <= dist[:, 1]).all() and (idx >= 0).all() and (idx < 8000).all()
