"""Library-owned device arrays (vo_dev_alloc / _upload / _download, hip.DeviceArray): round trips through every size
class of the host <-> device paths -- one piece of the page-locked ring (up to 1 MiB), several pieces (up to 16 MiB, the
ring going round more than once), the runtime's copy beyond; the result area (up to 1 MiB) and the runtime's copy back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nbytes", [0, 1, 15, 4096 + 3, (1 << 20) - 8, (1 << 20) + 16, 3 * (1 << 20) + 5, 7 * (1 << 20), (16 << 20) + 64])
def test_round_trip_through_every_upload_path(ctx, nbytes):
    from visual_odometry_amd import hip
    a = np.random.default_rng(nbytes).integers(0, 256, nbytes, dtype=np.uint8)
    d = hip.DeviceArray.from_numpy(ctx, a)
    assert d.shape == a.shape and d.dtype == a.dtype and d.nbytes == nbytes
    np.testing.assert_array_equal(d.numpy(), a)
    if nbytes >= 32:
        np.testing.assert_array_equal(d[:17].numpy(), a[:17])        # a leading view shares the memory
    d.close()


def test_views_shapes_and_pool_reuse(ctx):
    from visual_odometry_amd import hip
    a = np.arange(24 * 128, dtype=np.float32).reshape(24, 128)
    d = hip.DeviceArray.from_numpy(ctx, a)
    assert len(d) == 24 and d.ndim == 2
    np.testing.assert_array_equal(d[:5].numpy(), a[:5])
    ptr = d.ptr
    d.close()
    e = hip.DeviceArray(ctx, (20, 128), np.float32)                  # same power-of-two class: the pooled block comes back
    assert e.ptr == ptr
    e.close()
    with pytest.raises((ValueError, TypeError, IndexError)):
        hip.DeviceArray.from_numpy(ctx, a)[3:5]                       # only leading views exist
