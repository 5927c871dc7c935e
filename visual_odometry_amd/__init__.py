"""MI355X-native (gfx950) data-parallel front end of the visual-odometry pipeline.

`visual_odometry_amd.vo_pipeline` mirrors the reference's `vo_pipeline` module
API; its hot-path methods forward through ctypes to `csrc/libvo_hip.so`
(hand-written HIP kernels).  There is no CPU fallback: without the library the
hot-path calls raise.
"""
__version__ = "0.1.0"
