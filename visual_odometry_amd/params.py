"""Pipeline constants.  The values are part of the drop-in contract: they are the
ones the reference's params.py (params.py:4-46) resolves at import time from
DATASET; here `for_dataset()` returns the same table for any dataset name."""
from types import SimpleNamespace


def for_dataset(dataset: str = "KITTI") -> SimpleNamespace:
    parking = dataset == "PARKING"
    return SimpleNamespace(
        DATASET=dataset,
        FRAME_QUEUE_SIZE=100,
        MATCHING_THRESHOLD=0.8, MATCHING_RATIO=0.4, MIN_FRAME_DIST=5,
        RANSAC_REPROJ_THRESHOLD=0.5 if parking else 0.1,
        RANSAC_CONFIDENCE=0.999 if parking else 0.9999,
        RANSAC_MAX_ITERS=10000 if parking else 100,
        PNP_RANSAC_REPROJ_THRESHOLD=1,
        PNP_RANSAC_CONFIDENCE=0.99 if parking else 0.999,
        PNP_RANSAC_MAX_ITERS=10000 if parking else 100,
        KLT_RADIUS=21, KLT_LAMBDA=0.1, KLT_N_ITERS=10, KLT_NUM_PYRAMIDS=8, KLT_MIN_EIGEN_THRESHOLD=1e-2,
        MAX_BASELINE_UNCERTAINTY=0.15, MIN_INLIER_RATIO=0.5, BA_DISTANCE_TH=1e2,
        BUNDLE_ADJUSTMENT_KEYFRAME_LOOK_BACK=20 if parking else 4,
    )


def use_dataset(dataset: str) -> None:
    """Switch the module-level constants to `dataset` (the reference edits DATASET in params.py)."""
    globals().update(vars(for_dataset(dataset)))


use_dataset("KITTI")
