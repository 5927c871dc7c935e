// Internal declarations shared by the HIP translation units of libvo_hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include "../../include/vo_hip.h"

#define VO_MAX_LEVELS 10

void vo_set_error(const char *fmt, ...);

#define VO_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            vo_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return VO_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define VO_REQUIRE(cond, ...)                                                           \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            vo_set_error(__VA_ARGS__);                                                  \
            return VO_EINVAL;                                                           \
        }                                                                               \
    } while (0)

// Growable device scratch owned by a context (staging of host-side arguments).
struct vo_devbuf {
    void *ptr = nullptr;
    size_t cap = 0;
};

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    vo_devbuf stage[8];
    void *pinned = nullptr;      // page-locked host buffer for small read-backs (a copy into pageable memory blocks the caller)
    size_t pinned_cap = 0;
};

bool vo_ctx_is_live(const vo_ctx *ctx);

// Looks up the current device's copy of the fixed scratch array of vo_klt_track_fb_noalloc (vo_klt.hip).
int vo_klt_noalloc_setup(int device);

// Page-locked host scratch of at least `bytes` (contents survive until the next call that asks for more).
int vo_pinned_reserve(vo_ctx *ctx, size_t bytes, void **out);

// Make stage[slot] at least `bytes` large; returns its device pointer in *out.
int vo_stage_reserve(vo_ctx *ctx, int slot, size_t bytes, void **out);

struct vo_pyr {
    vo_ctx *ctx = nullptr;
    int batch = 0, win = 0, max_level = 0;  // max_level: effective last level
    int h[VO_MAX_LEVELS] = {0}, w[VO_MAX_LEVELS] = {0};
    int64_t pitch[VO_MAX_LEVELS] = {0};       // bytes between rows
    int64_t img_stride[VO_MAX_LEVELS] = {0};  // bytes between images
    uint8_t *lvl[VO_MAX_LEVELS] = {nullptr};  // device pointers; lvl[0] may alias caller memory
    uint8_t *pool = nullptr;                  // levels >= 1
    uint8_t *own0 = nullptr;                  // library-owned level 0 (host-memory builds)
    size_t own0_bytes = 0;
    bool built = false;
};
