"""vo_sift_detect_compute / vo_sift_scale_space_image (through the C-ABI) against the CPU oracle: scale-space
images, keypoints (all six fields) and descriptors bit for bit; the FeatureExtractor.get_kp mirror; the bootstrap
on SIFT correspondences."""
import numpy as np
import pytest

from visual_odometry_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,seed", [((97, 131), 3), ((60, 80), 1)])
def test_scale_space_bit_exact(ctx, oracle, shape, seed):
    from visual_odometry_amd import hip
    img = synth.frame_pair(shape, seed=seed)[0]
    n_oct = oracle.lib().vo_or_sift_num_octaves(shape[1], shape[0])
    for o in range(n_oct):
        for i in range(7):
            got = hip.sift_scale_space_image(ctx, img, o, i)
            ref = oracle.sift_pyramid_image(img, o, i)
            assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (o, i)
        for i in range(6):
            got = hip.sift_scale_space_image(ctx, img, o, i, dog=True)
            assert np.array_equal(got, oracle.sift_pyramid_image(img, o, i, dog=True)), (o, i)
    with pytest.raises(ValueError):
        hip.sift_scale_space_image(ctx, img, n_oct, 0)


def _same(kp, des, rk, rd):
    assert len(kp) == len(rk)
    for f in ("x", "y", "size", "angle", "response"):
        assert np.array_equal(kp[f].view(np.uint32), rk[f].view(np.uint32)), f
    assert np.array_equal(kp["octave"], rk["octave"])
    if des is not None:
        assert des.dtype == np.float32 and des.shape == rd.shape and np.array_equal(des, rd)


@pytest.mark.parametrize("shape,layers,sigma", [((30, 400), 5, 2.0), ((400, 26), 3, 1.2), ((64, 64), 8, 2.4), ((150, 210), 2, 1.6)])
def test_scale_space_small_octaves_other_parameters(ctx, oracle, shape, layers, sigma):
    """The last octaves come from one launch that keeps the layers in LDS with reflected borders as wide as the largest blur:
    thin images (a border wider than the image, octaves that do and do not fit that launch), other layer counts and sigmas."""
    from visual_odometry_amd import hip
    img = synth.frame_pair(shape, seed=11)[0]
    n_oct = oracle.lib().vo_or_sift_num_octaves(shape[1], shape[0])
    for o in range(max(0, n_oct - 5), n_oct):
        for i in range(layers + 3):
            got = hip.sift_scale_space_image(ctx, img, o, i, n_octave_layers=layers, sigma=sigma)
            ref = oracle.sift_pyramid_image(img, o, i, n_octave_layers=layers, sigma=sigma)
            assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (o, i)
        for i in range(layers + 2):
            got = hip.sift_scale_space_image(ctx, img, o, i, dog=True, n_octave_layers=layers, sigma=sigma)
            ref = oracle.sift_pyramid_image(img, o, i, dog=True, n_octave_layers=layers, sigma=sigma)
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (o, i)


@pytest.mark.parametrize("shape,seed", [((240, 320), 3), ((97, 131), 5), ((376, 1241), 2021), ((33, 300), 7)])
def test_detect_compute_bit_exact(ctx, oracle, shape, seed):
    from visual_odometry_amd import hip
    img = synth.frame_pair(shape, seed=seed)[0]
    rk, rd = oracle.sift(img)
    kp, des = hip.sift_detect_compute(ctx, img)
    assert len(kp) > 20
    _same(kp, des, rk, rd)
    k2, none = hip.sift_detect_compute(ctx, img, want_descriptors=False)
    assert none is None
    _same(k2, None, rk, rd)


def test_other_parameters_blobs_and_edge_cases(ctx, oracle):
    import torch
    from visual_odometry_amd import hip
    yy, xx = np.mgrid[0:160, 0:200].astype(np.float64)
    img = np.full((160, 200), 40.0)
    for cy, cx, s in [(50.0, 60.0, 4.0), (110.3, 140.6, 7.0), (60.5, 150.2, 2.5)]:
        img += 180.0 * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    img = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    for args in [dict(), dict(n_octave_layers=3, contrast_threshold=0.04, edge_threshold=8.0, sigma=1.2),
                 dict(n_octave_layers=5, contrast_threshold=0.01, edge_threshold=20.0, sigma=2.0)]:
        rk, rd = oracle.sift(img, **args)
        kp, des = hip.sift_detect_compute(ctx, img, **args)
        _same(kp, des, rk, rd)
    tex = synth.frame_pair((120, 160), seed=9)[0]
    rk, rd = oracle.sift(tex)
    few, dfew = hip.sift_detect_compute(ctx, tex, max_keypoints=17)            # truncated output = the first rows
    _same(few, dfew, rk[:17], rd[:17])
    tctx = hip.Context.from_torch(0)
    big = torch.from_numpy(np.pad(tex, ((2, 3), (5, 7)))).cuda()
    kd, dd = hip.sift_detect_compute(tctx, big[2:122, 5:165])                  # device image with padded rows
    _same(kd, dd.cpu().numpy(), rk, rd)
    flat, dflat = hip.sift_detect_compute(ctx, np.full((40, 50), 128, np.uint8))
    assert len(flat) == 0 and dflat.shape == (0, 128)
    with pytest.raises(ValueError):
        hip.sift_detect_compute(ctx, tex, n_octave_layers=0)


def test_get_kp_mirror_and_sift_bootstrap(ctx, oracle):
    """FeatureExtractor.get_kp returns cv2-style keypoints + descriptors; BootstrapInitializer's default route
    (SIFT -> two-nearest-neighbour matching -> F-RANSAC -> two-view geometry) recovers a planted camera motion."""
    from visual_odometry_amd import params
    from visual_odometry_amd.vo_pipeline.bootstrap import BootstrapInitializer
    from visual_odometry_amd.vo_pipeline.featureExtraction import ExtractorType, FeatureExtractor
    scene = synth.BoxScene(n_frames=12, seed=5)
    frames = [img for _, img in scene.frames()]
    kps, des = FeatureExtractor(ExtractorType.SIFT).get_kp(frames[0])
    rk, rd = oracle.sift(frames[0])
    assert len(kps) == len(rk) and np.array_equal(des, rd)
    assert kps[3].pt == (float(rk["x"][3]), float(rk["y"][3])) and kps[3].octave == int(rk["octave"][3])
    # default (KITTI) RANSAC parameters: 0.1 px gate.  The matches are good to ~0.05 px; with PARKING's 0.5 px gate nearly
    # every minimal model collects every match and the first of them wins, whatever its translation
    assert params.RANSAC_REPROJ_THRESHOLD == 0.1
    b = BootstrapInitializer(frames[0], frames[10], np.asarray(scene.K, np.float32), max_point_dist=100)
    assert b.pts_des1.shape[1] == 128 and b.pts_des1.any() and b.pts1.shape[0] > 100
    rel = np.linalg.inv(scene.poses[10]) @ scene.poses[0]           # poses are camera-to-world; T maps camera 0 to camera 10
    np.testing.assert_allclose(b.T[:3, :3], rel[:3, :3], atol=2e-2)
    t_true = rel[:3, 3] / np.linalg.norm(rel[:3, 3])
    assert b.T[:3, 3] @ t_true > 0.97


def test_buffers_grow_with_the_image(oracle, monkeypatch):
    """A frame with more scale-space extrema or raw keypoints than the library's room (set tiny here): the room grows to what the counters ask for and the passes run again -- same result as
    the checker, no bound but memory, as cv2's nfeatures = 0.  More keypoints than the caller's first room: the
    Python call asks again for all of them instead of clipping the right-hand side of the image."""
    from visual_odometry_amd import hip
    monkeypatch.setenv("VO_SIFT_CAND_CAP", "1024")        # 16 per list
    monkeypatch.setenv("VO_SIFT_KP_CAP", "512")           # 8 per list
    small = hip.Context(0)                                # a fresh context reads the starting room
    try:
        img = synth.frame_pair((120, 160), seed=3)[0]
        rk, rd = oracle.sift(img)
        assert len(rk) > 500                              # ~13 per keypoint list, more extrema than that per candidate list
        kp, des = hip.sift_detect_compute(small, img)
        _same(kp, des, rk, rd)
        kp, des = hip.sift_detect_compute(small, img, max_keypoints=100)      # an explicit bound: the first rows
        _same(kp, des, rk[:100], rd[:100])
        monkeypatch.setattr(hip, "SIFT_FIRST_ROOM", 300, raising=False)
        kp, des = hip.sift_detect_compute(small, img, device_descriptors=True)
        assert isinstance(des, hip.DeviceArray) and des.shape == (len(rk), 128)
        _same(kp, des.numpy(), rk, rd)
        # a smaller image afterwards, in the grown buffers
        img2 = synth.frame_pair((60, 80), seed=1)[0]
        _same(*hip.sift_detect_compute(small, img2), *oracle.sift(img2))
    finally:
        small.close()


def test_device_descriptors_feed_the_matcher(ctx, oracle):
    """Descriptors left on the GPU in library-owned memory (no tensor library involved) and matched there equal
    the host route."""
    from visual_odometry_amd import hip
    a, b, _ = synth.frame_pair((200, 260), seed=9, max_shift=4.0)
    k0, d0 = hip.sift_detect_compute(ctx, a)
    k1, d1 = hip.sift_detect_compute(ctx, b)
    g0 = hip.sift_detect_compute(ctx, a, device_descriptors=True)[1]
    g1 = hip.sift_detect_compute(ctx, b, device_descriptors=True)[1]
    assert np.array_equal(g0.numpy(), d0) and np.array_equal(g1.numpy(), d1)
    ref = hip.knn_match2(ctx, d0, d1, 0.8)
    got = hip.knn_match2(ctx, g0, g1, 0.8)
    for x, y in zip(got, ref):
        assert np.array_equal(x, y)
    assert np.array_equal(g0[:10].numpy(), d0[:10])
    with pytest.raises(IndexError):
        g0[5:]
