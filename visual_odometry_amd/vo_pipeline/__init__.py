"""Drop-in mirror of the reference's `vo_pipeline` hot-path modules.

Same module, class and method names, argument meaning and return conventions as
the reference; the arithmetic the reference delegates to cv2 / scipy runs in
libvo_hip.so on the GPU instead.  `visual_odometry_amd.install_reference_names()`
makes them importable as `vo_pipeline.<module>`.
"""
