"""CPU restatement of cv2's SIFT (oracle/vo_oracle_sift.c): pieces against independent scipy arithmetic and
known-answer properties (blob scale, rotation covariance, repeatability under a known shift).  cv2 itself is
absent: parity with it is unpinned."""
import numpy as np
import pytest
from scipy import ndimage

from visual_odometry_amd import synth


def test_gaussian_kernel_and_blur_against_ndimage(oracle):
    k = oracle.sift_gauss_kernel(1.6)
    assert k.size == 15 and abs(k.sum() - 1) < 1e-6 and np.allclose(k, k[::-1])
    x = np.arange(15) - 7
    ref = np.exp(-0.5 * x * x / 1.6 ** 2); ref /= ref.sum()
    assert np.allclose(k, ref, atol=1e-7)
    img = synth.frame_pair((60, 80), seed=1)[0]
    g0 = oracle.sift_pyramid_image(img, 0, 0)
    g1 = oracle.sift_pyramid_image(img, 0, 1)
    assert g0.shape == (120, 160)
    sig1 = 1.6 * np.sqrt(2 ** 0.5 - 1)                       # sigma * sqrt(k^2 - 1), k = 2^(1/4)
    k1 = oracle.sift_gauss_kernel(sig1)
    ref = ndimage.correlate1d(ndimage.correlate1d(g0.astype(np.float64), k1.astype(np.float64), axis=1, mode="mirror"),
                              k1.astype(np.float64), axis=0, mode="mirror")
    assert np.abs(g1 - ref).max() < 1e-3
    # the base image: centre-aligned bilinear doubling, then a blur to sigma 1.6 from the assumed 2 x 0.5
    up = ndimage.zoom(img.astype(np.float64), 2, order=1, mode="nearest", grid_mode=True)
    kb = oracle.sift_gauss_kernel(np.sqrt(1.6 ** 2 - 1.0)).astype(np.float64)
    ref0 = ndimage.correlate1d(ndimage.correlate1d(up, kb, axis=1, mode="mirror"), kb, axis=0, mode="mirror")
    assert np.abs(g0 - ref0).max() < 2e-3
    # next octave starts from every second pixel of the image with twice the base sigma
    assert np.array_equal(oracle.sift_pyramid_image(img, 1, 0), oracle.sift_pyramid_image(img, 0, 4)[::2, ::2])
    d2 = oracle.sift_pyramid_image(img, 0, 2, dog=True)
    assert np.array_equal(d2, oracle.sift_pyramid_image(img, 0, 3) - oracle.sift_pyramid_image(img, 0, 2))


def _blobs(shape=(160, 200), sig=4.0):
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
    img = np.full(shape, 40.0)
    centres = [(50.0, 60.0), (110.3, 140.6), (60.5, 150.2)]
    for cy, cx in centres:
        img += 180.0 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sig * sig))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), centres


def test_gaussian_blobs_are_found_at_their_scale(oracle):
    img, centres = _blobs(sig=4.0)
    kp, des = oracle.sift(img)
    assert des.shape == (len(kp), 128) and des.dtype == np.float32
    assert np.array_equal(des, np.rint(des)) and des.min() >= 0 and des.max() <= 255
    for cy, cx in centres:
        d = np.hypot(kp["x"] - cx, kp["y"] - cy)
        near = kp[d < 1.0]
        assert len(near) >= 1, (cy, cx)
        best = near[np.argmax(near["response"])]
        assert 6.0 < best["size"] < 14.0            # diameter 2 sigma' with sigma' ~ blob sigma (x sqrt 2 for a DoG peak)
    # sorted as cv2 returns them: by x, then y
    assert np.all(np.diff(kp["x"]) >= 0)


def test_rotation_by_90_degrees_is_covariant(oracle):
    img = synth.frame_pair((120, 150), seed=4)[0]
    kp, des = oracle.sift(img)
    rot = np.ascontiguousarray(np.rot90(img))                # (x, y) -> (y, W - 1 - x)
    kp_r, des_r = oracle.sift(rot)
    assert abs(len(kp) - len(kp_r)) <= 0.03 * len(kp)
    W = img.shape[1]
    # OpenCV's SIFT halves the coordinates of the doubled image without the quarter-pixel shift of a centre-aligned
    # doubling, so reported positions sit 0.25 px right/below the true ones: a true (x, y) -> (y, W - 1 - x) becomes
    # reported (x, y) -> (y, W - 1 - x + 0.5)
    want = np.stack([kp["y"], W - 1 - kp["x"] + 0.5], axis=1)
    got = np.stack([kp_r["x"], kp_r["y"]], axis=1)
    matched = 0
    for i in range(len(kp)):
        d = np.hypot(got[:, 0] - want[i, 0], got[:, 1] - want[i, 1])
        for j in np.nonzero(d < 0.05)[0]:
            # KeyPoint.angle grows clockwise on the screen (y points down): a counter-clockwise turn lowers it by 90
            da = (kp_r["angle"][j] - kp["angle"][i] + 90.0) % 360.0
            if min(da, 360 - da) < 2.0 and abs(kp_r["size"][j] - kp["size"][i]) < 0.02 * kp["size"][i]:
                if np.linalg.norm(des_r[j] - des[i]) < 60:
                    matched += 1
                    break
    assert matched >= 0.9 * len(kp)


def test_repeatability_under_a_known_shift(oracle):
    rng = np.random.default_rng(3)
    tex = synth.texture((200, 260), rng, 0)
    a = np.clip(np.rint(tex[20:180, 30:230]), 0, 255).astype(np.uint8)
    b = np.clip(np.rint(tex[13:173, 41:241]), 0, 255).astype(np.uint8)     # content moves by (-11, +7)
    ka, da = oracle.sift(a)
    kb, db = oracle.sift(b)
    idx, dist, keep = oracle.knn_match2(da, db, 0.8)
    q = np.nonzero(keep)[0]
    assert q.size > 0.3 * len(ka)
    flow = np.stack([kb["x"][idx[q, 0]] - ka["x"][q], kb["y"][idx[q, 0]] - ka["y"][q]], axis=1)
    good = (np.abs(flow[:, 0] + 11) < 0.2) & (np.abs(flow[:, 1] - 7) < 0.2)
    assert good.mean() > 0.95


def test_keypoint_fields_and_limits(oracle):
    img = synth.frame_pair((90, 120), seed=8)[0]
    kp, des = oracle.sift(img)
    assert len(kp) > 50
    octave = (kp["octave"] & 255).astype(np.int8)
    layer = (kp["octave"] >> 8) & 255
    assert octave.min() >= -1 and layer.min() >= 1 and layer.max() <= 4
    assert np.all((kp["angle"] >= 0) & (kp["angle"] < 360)) and np.all(kp["response"] * 4 >= 0.03 - 1e-6)
    few, d2 = oracle.sift(img, max_keypoints=10)
    assert len(few) == 10 and np.array_equal(few, kp[:10]) and np.array_equal(d2, des[:10])
    flat, dflat = oracle.sift(np.full((40, 50), 128, np.uint8))
    assert len(flat) == 0 and dflat.shape == (0, 128)


def test_descriptor_rotation_matches_libm(oracle):
    """The written-out cos / sin of the descriptor's rotation against numpy's (libm) float64 cos / sin rounded to
    float32 -- what cosf / sinf return -- over the whole range of KeyPoint.angle and at the quadrant borders."""
    rng = np.random.default_rng(11)
    deg = np.concatenate([rng.uniform(0, 360, 40000).astype(np.float32),
                          np.array([0, 45, 90, 135, 180, 225, 270, 315, 359.99997, 360, 1e-6, 89.99999, 90.00001], np.float32)])
    t = deg * np.float32(np.pi / 180.0)
    c, s = oracle.sift_sincos(t)
    ref_c, ref_s = np.cos(t.astype(np.float64)).astype(np.float32), np.sin(t.astype(np.float64)).astype(np.float32)
    assert np.array_equal(c, ref_c) and np.array_equal(s, ref_s)
