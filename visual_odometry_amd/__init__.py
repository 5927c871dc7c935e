"""MI355X-native (gfx950) data-parallel front end of the visual-odometry pipeline.

`visual_odometry_amd.vo_pipeline` mirrors the reference's `vo_pipeline` module
API; its hot-path methods forward through ctypes to `csrc/libvo_hip.so`
(hand-written HIP kernels).  There is no CPU fallback: without the library the
hot-path calls raise.
"""
__version__ = "0.1.0"


def install_reference_names() -> None:
    """Make the reference's absolute module names resolve to this package: after the call
    `from vo_pipeline.poseEstimation import PoseEstimation`, `from utils.matrix import hom_inv`,
    `from utils.frameQueue import FrameQueue` and `import params` import the mirror, so a script
    written against the reference runs unchanged (for the modules the mirror has)."""
    import importlib
    import pkgutil
    import sys
    for alias, name in (("vo_pipeline", __name__ + ".vo_pipeline"), ("utils", __name__ + ".utils"),
                        ("params", __name__ + ".params")):
        mod = importlib.import_module(name)
        sys.modules[alias] = mod
        for info in pkgutil.iter_modules(getattr(mod, "__path__", [])):
            if not info.name.startswith("_"):
                sys.modules[alias + "." + info.name] = importlib.import_module(name + "." + info.name)
