"""ctypes binding of csrc/libvo_hip.so (C-ABI declared in include/vo_hip.h)."""
import ctypes
import os

_DIR = os.path.dirname(os.path.abspath(__file__))
# VO_HIP_LIB names another build of the same library (A/B timing of two kernel versions on one box)
LIB_PATH = os.environ.get("VO_HIP_LIB") or os.path.join(_DIR, "csrc", "libvo_hip.so")

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_vp = ctypes.c_void_p

VO_MEM_HOST = 0
VO_MEM_DEVICE = 1


class BaLmStats(ctypes.Structure):
    _fields_ = [("cost_start", ctypes.c_double), ("cost_final", ctypes.c_double), ("lam", ctypes.c_double),
                ("trial_steps", ctypes.c_int32), ("accepted", ctypes.c_int32), ("iterations", ctypes.c_int32),
                ("stop_reason", ctypes.c_int32)]


class KltParams(ctypes.Structure):
    _fields_ = [("max_iters", ctypes.c_int), ("eps", ctypes.c_float), ("min_eig", ctypes.c_float)]


# name -> (restype, argtypes); every symbol include/vo_hip.h declares
SIGNATURES = {
    "vo_last_error": (ctypes.c_char_p, []),
    "vo_version": (ctypes.c_int, []),
    "vo_ctx_create": (ctypes.c_int, [ctypes.c_int, c_vp, ctypes.POINTER(c_vp)]),
    "vo_ctx_create_own_stream": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(c_vp)]),
    "vo_ctx_destroy": (ctypes.c_int, [c_vp]),
    "vo_sync": (ctypes.c_int, [c_vp]),
    "vo_dev_alloc": (ctypes.c_int, [c_vp, ctypes.c_size_t, ctypes.POINTER(c_vp)]),
    "vo_dev_free": (ctypes.c_int, [c_vp, c_vp]),
    "vo_dev_upload": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t]),
    "vo_dev_download": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t]),
    "vo_timer_start": (ctypes.c_int, [c_vp]),
    "vo_timer_stop_ms": (ctypes.c_int, [c_vp, c_f32p]),
    "vo_pyr_create": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(c_vp)]),
    "vo_pyr_destroy": (ctypes.c_int, [c_vp]),
    "vo_pyr_max_level": (ctypes.c_int, [c_vp]),
    "vo_pyr_level_shape": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                          ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)]),
    "vo_pyr_build": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]),
    "vo_pyr_download": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp]),
    "vo_klt_track": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.POINTER(KltParams),
                                    c_vp, c_vp, c_vp, ctypes.c_int]),
    "vo_klt_track_fb": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.POINTER(KltParams),
                                       ctypes.c_float, c_vp, c_vp, ctypes.c_int]),
    "vo_klt_track_fb_noalloc": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, ctypes.POINTER(KltParams),
                                               ctypes.c_float, c_vp, c_vp]),
    "vo_score_fundamental": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_double,
                                            c_vp, c_vp, ctypes.c_int]),
    "vo_score_pnp": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_f64p, c_vp, ctypes.c_int, ctypes.c_double,
                                    c_vp, c_vp, ctypes.c_int]),
    "vo_ba_residual": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_f64p,
                                      c_vp, c_vp, ctypes.c_int]),
    "vo_ba_normal_eq": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_f64p,
                                       ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int]),
    "vo_harris_response": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                          c_vp, ctypes.c_int]),
    "vo_nms_topk": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                   ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int]),
    "vo_detect_corners": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                         ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                         ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int]),
    "vo_p3p_solve": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_f64p, c_vp, ctypes.c_int, c_vp, c_vp,
                                    ctypes.c_int]),
    "vo_seven_point_solve": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int, c_vp, c_vp,
                                            ctypes.c_int]),
    "vo_triangulate": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_f64p, c_f64p, ctypes.c_int, ctypes.c_int, c_vp,
                                      ctypes.c_int]),
    "vo_cheirality_vote": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_f64p, c_f64p, c_f64p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int, c_vp, c_vp, ctypes.c_int]),
    "vo_ba_create": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_f64p, ctypes.c_int,
                                    ctypes.c_int, ctypes.POINTER(c_vp)]),
    "vo_ba_create_filtered": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, ctypes.c_int, c_f64p, c_vp,
                                             ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_vp),
                                             ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "vo_ba_window_assemble": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(ctypes.c_int32)]),
    "vo_ba_camera_params": (ctypes.c_int, [ctypes.c_int, c_vp, c_vp]),
    "vo_ba_destroy": (None, [c_vp]),
    "vo_ba_solve": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, c_vp, ctypes.c_int,
                                   c_vp]),
    "vo_ba_solve_begin": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int]),
    "vo_ba_solve_finish": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp]),
    "vo_ba_linearize": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_f64p]),
    "vo_ba_try_step": (ctypes.c_int, [c_vp, c_vp, ctypes.c_double, c_f64p, ctypes.POINTER(ctypes.c_int)]),
    "vo_ba_accept": (ctypes.c_int, [c_vp, c_vp]),
    "vo_ba_params": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int]),
    "vo_ransac_draw_subsets": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp,
                                              c_vp, ctypes.c_int, c_vp]),
    "vo_solve_pnp_ransac": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_f64p, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), c_f64p, c_vp,
                                           ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    "vo_track_pnp": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_int64, c_vp, c_vp, ctypes.c_int, c_vp, c_f64p, ctypes.c_double, ctypes.c_double,
                                    ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), c_vp, c_vp, c_f64p, c_vp,
                                    ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    "vo_track_pnp_path": (ctypes.c_int, [c_vp]),
    "vo_track_pnp_batch": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_f64p, ctypes.c_double, ctypes.c_double,
                                          ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_f64p, c_vp, c_vp, c_vp, c_vp, ctypes.c_int]),
    "vo_track_pnp_batch_begin": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_f64p, ctypes.c_double,
                                                ctypes.c_double, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp, c_vp, ctypes.c_int]),
    "vo_profile_tracker": (ctypes.c_int, [c_vp, ctypes.c_int]),
    "vo_profile_tracker_ms": (ctypes.c_int, [c_vp, c_f32p]),
    "vo_track_pnp_batch_finish": (ctypes.c_int, [c_vp, c_vp, c_f64p, c_vp, c_vp, c_vp]),
    "vo_find_fundamental_ransac": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                  ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), c_f64p, c_vp,
                                                  ctypes.POINTER(ctypes.c_int32)]),
    "vo_refine_pose": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, c_vp, c_f64p, c_f64p, ctypes.c_int, c_f64p, c_f64p,
                                      ctypes.c_int]),
    "vo_harris_mask": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                      ctypes.c_float, ctypes.c_int, ctypes.c_double, c_vp, ctypes.c_int]),
    "vo_corner_subpix": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, c_vp, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int]),
    "vo_connected_components": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_vp, c_vp,
                                               c_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    "vo_harris_corners": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_vp,
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    "vo_sift_detect_compute": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                              ctypes.c_double, ctypes.c_double, c_vp, c_vp, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    "vo_sift_scale_space_image": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_double,
                                                 ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.POINTER(ctypes.c_int),
                                                 ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "vo_knn_match2": (ctypes.c_int, [c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_vp, c_vp,
                                     c_vp, ctypes.c_int]),
}

_lib = None


class VoError(RuntimeError):
    pass


def load():
    """Load libvo_hip.so.  Raises if the library is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VoError(f"{LIB_PATH} is missing: build it with `make -C {os.path.dirname(LIB_PATH)}` "
                      "(or __graft_entry__.build()); the HIP library is required, there is no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().vo_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise VoError(f"{what}: [{rc}] {msg}")
