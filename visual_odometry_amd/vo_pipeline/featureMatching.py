"""FeatureMatcher -- API of the reference's vo_pipeline/featureMatching.py:12-76."""
import enum
from typing import List, Tuple

import numpy as np

from .. import hip
from . import _runtime


class MatcherType(enum.Enum):
    # brute force matching
    BF = 0,
    # Fast KNN matching
    FLANN = 1,


class DMatch:
    """The fields of cv2.DMatch that the pipeline reads (bootstrap.py:207-211)."""
    __slots__ = ("queryIdx", "trainIdx", "imgIdx", "distance")

    def __init__(self, queryIdx: int, trainIdx: int, distance: float, imgIdx: int = 0):
        self.queryIdx, self.trainIdx, self.imgIdx, self.distance = queryIdx, trainIdx, imgIdx, distance

    def __repr__(self):
        return "DMatch(queryIdx=%d, trainIdx=%d, distance=%r)" % (self.queryIdx, self.trainIdx, self.distance)


class FeatureMatcher:
    """Two-nearest-neighbour descriptor matching with Lowe's ratio test.

    Both matcher types run the same exhaustive float32 L2 search on the GPU (vo_knn_match2).
    For MatcherType.BF that is the reference's cv.BFMatcher() (featureMatching.py:45); for
    MatcherType.FLANN it is the exact answer that the reference's randomised kd-tree search
    (featureMatching.py:46-50) approximates, so `nbr_tress` and `nbr_search` are kept but unused.
    """

    def __init__(self, matcher_type: MatcherType,
                 k: int = 2,
                 nbr_tress: int = 5,
                 nbr_search: int = 50,
                 matching_threshold: float = 0.8):
        """k: neighbours per query (1 or 2).  matching_threshold: the ratio of the ratio test (a match is kept when its
        distance is below this fraction of the runner-up's).  nbr_tress, nbr_search: the kd-tree settings of the
        reference's FLANN matcher, accepted and ignored -- the search here is exhaustive."""
        self.k = k
        self.nbr_tress = nbr_tress
        self.nbr_search = nbr_search
        self.matching_threshold = matching_threshold
        self.matcher_type = matcher_type
        self.matcher = None
        self.get_matcher()

    def get_matcher(self):
        if not isinstance(self.matcher_type, MatcherType):
            raise ValueError("unknown matcher type %r" % (self.matcher_type,))
        if self.k not in (1, 2):
            raise ValueError("FeatureMatcher: k = %r; the GPU search returns the one or two nearest neighbours" % (self.k,))
        self.matcher = self.match_arrays

    def match_arrays(self, des1, des2) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """The search without the Python objects: (idx (n1,2) int32, dist (n1,2) float32, keep (n1) bool) as
        numpy arrays, or torch tensors on the device when the descriptors are."""
        idx, dist, keep = hip.knn_match2(_runtime.context(), des1, des2, self.matching_threshold)
        return idx, dist, (keep != 0)

    def match_descriptors(self, des1: np.ndarray, des2: np.ndarray, filter: bool = True) -> List[DMatch]:
        """
        :param des1:    Descriptors of image 1 / point cloud, (n1, 128)
        :param des2:    Descriptors of image 2, (n2, 128)
        :param filter:  if True, only the "good" matches are returned
        :return: the nearest match of every (kept) query row, in query order (featureMatching.py:58-63)
        """
        des1 = np.asarray(des1)
        des2 = np.asarray(des2)
        if des2.shape[0] == 0 and des1.shape[0] > 0:
            raise IndexError("list index out of range")      # match[0] of an empty knn list
        if filter and des1.shape[0] > 0 and min(self.k, des2.shape[0]) < 2:
            raise IndexError("list index out of range")      # match[1] of a one-element knn list
        idx, dist, keep = self.match_arrays(des1, des2)
        rows = np.nonzero(keep)[0] if filter else np.arange(idx.shape[0])
        return [DMatch(int(q), int(idx[q, 0]), float(dist[q, 0])) for q in rows]

    def match_filter(self, matches):
        # ratio test from Lowe's paper: Distinctive Image Features from Scale-Invariant Keypoints
        return (match for match in matches if
                match[0].distance < self.matching_threshold * match[1].distance)
