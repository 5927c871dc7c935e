"""The matching oracle against an independent numpy search, and the host-side argument checks of the
FeatureMatcher mirror (no GPU calls)."""
import numpy as np
import pytest

from visual_odometry_amd import synth


def _numpy_knn2(a, b):
    D = ((a[:, None, :].astype(np.float64) - b[None].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(D, axis=1, kind="stable")[:, :2]
    return order, np.sqrt(np.take_along_axis(D, order, 1))


def test_oracle_byte_descriptors_exact(oracle):
    d1, d2, truth = synth.sift_like_descriptors(300, 350, seed=1)
    idx, dist, keep = oracle.knn_match2(d1, d2, 0.8)
    ref_i, ref_d = _numpy_knn2(d1, d2)
    assert (idx == ref_i).all()
    # integer entries: the float32 sum of squares is exact, so the distance is the rounded square root
    assert (dist == ref_d.astype(np.float32)).all()
    assert (keep == (dist[:, 0].astype(np.float64) < 0.8 * dist[:, 1].astype(np.float64))).all()
    hit = truth >= 0
    assert (idx[hit, 0] == truth[hit]).mean() > 0.95 and keep[hit].mean() > 0.9 and keep[~hit].mean() < 0.1


def test_oracle_float_descriptors(oracle):
    rng = np.random.default_rng(3)
    d1 = rng.normal(0, 1, (200, 128)).astype(np.float32)
    d2 = rng.normal(0, 1, (260, 128)).astype(np.float32)
    idx, dist, _ = oracle.knn_match2(d1, d2)
    ref_i, ref_d = _numpy_knn2(d1, d2)
    assert (idx == ref_i).mean() > 0.995            # float32 vs float64 sums may swap near-ties
    assert np.abs(dist - ref_d).max() < 1e-4


def test_oracle_ties_and_missing_neighbours(oracle):
    row = np.arange(128, dtype=np.float32)
    d2 = np.stack([row + 1, row, row, row + 1])      # rows 1 and 2 coincide with the query, 0 and 3 tie
    idx, dist, keep = oracle.knn_match2(row[None], d2)
    assert idx.tolist() == [[1, 2]] and dist.tolist() == [[0.0, 0.0]] and keep.tolist() == [0]
    idx, dist, keep = oracle.knn_match2(row[None], d2[[0, 3]])
    assert idx.tolist() == [[0, 1]]
    idx, dist, keep = oracle.knn_match2(row[None], d2[:1])
    assert idx.tolist() == [[0, -1]] and np.isinf(dist[0, 1]) and keep.tolist() == [0]
    idx, dist, keep = oracle.knn_match2(row[None], np.zeros((0, 128), np.float32))
    assert idx.tolist() == [[-1, -1]] and np.isinf(dist).all()


def test_matcher_mirror_argument_checks():
    from visual_odometry_amd.vo_pipeline.featureMatching import FeatureMatcher, MatcherType
    with pytest.raises(ValueError):
        FeatureMatcher(MatcherType.BF, k=3)
    with pytest.raises(ValueError):
        FeatureMatcher("flann")
    m = FeatureMatcher(MatcherType.FLANN, k=2, matching_threshold=0.8)
    assert (m.k, m.nbr_tress, m.nbr_search, m.matching_threshold) == (2, 5, 50, 0.8)
    one = np.zeros((1, 128), np.float32)
    with pytest.raises(IndexError):                  # the reference indexes match[1] of a one-element list
        m.match_descriptors(one, one)
    with pytest.raises(IndexError):
        m.match_descriptors(one, np.zeros((0, 128), np.float32), filter=False)
