// Context, error reporting, staging buffers and stream timers of libvo_hip.
#include "vo_internal.h"
#include <mutex>
#include <new>
#include <unordered_set>

// Live contexts: objects that outlive their context (pyramids destroyed at
// interpreter shutdown in arbitrary order) must not touch a freed vo_ctx.
static std::mutex g_live_mu;
static std::unordered_set<const vo_ctx *> &live_set()
{
    static auto *s = new std::unordered_set<const vo_ctx *>();  // never destroyed: safe during process exit
    return *s;
}
bool vo_ctx_is_live(const vo_ctx *ctx)
{
    std::lock_guard<std::mutex> lk(g_live_mu);
    return live_set().count(ctx) != 0;
}

static thread_local char g_err[512] = "";

void vo_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *vo_last_error(void) { return g_err; }
extern "C" int vo_version(void) { return 100; }

__global__ void copy_pinned_wide_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, unsigned n16);

extern "C" int vo_ctx_create(int device, void *stream, vo_ctx **out)
{
    VO_REQUIRE(out != nullptr, "vo_ctx_create: out is NULL");
    int count = 0;
    VO_HIP_CHECK(hipGetDeviceCount(&count));
    VO_REQUIRE(device >= 0 && device < count, "vo_ctx_create: device %d out of range (%d visible)", device, count);
    VO_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    VO_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        vo_set_error("vo_ctx_create: device %d is %s; this library ships gfx950 code only", device, prop.gcnArchName);
        return VO_ESTATE;
    }
    {
        static std::mutex setup_mu;
        std::lock_guard<std::mutex> lk(setup_mu);
        int rc = vo_klt_noalloc_setup(device);   // once per device, with that device current
        if (rc) return rc;
        static bool loaded[64];
        if (device < 64 && !loaded[device]) {
            const char *lazy = getenv("VO_LAZY_CODE_LOADING");        // "1": leave code loading to the first launches
            if (!(lazy && lazy[0] == '1')) {
                int (*const loaders[])(void) = {vo_load_code_ba, vo_load_code_cc, vo_load_code_frame, vo_load_code_harris, vo_load_code_klt, vo_load_code_match, vo_load_code_pyramid, vo_load_code_score, vo_load_code_sift, vo_load_code_solvers, vo_load_code_subpix, vo_load_code_twoview};
                for (auto fn : loaders)
                    if ((rc = fn())) return rc;
                // the copy kernels of this file, and the runtime's first-use set-up of page-locked memory and of a stream
                hipFuncAttributes attr;
                VO_HIP_CHECK(hipFuncGetAttributes(&attr, (const void *)copy_pinned_wide_kernel));
            }
            loaded[device] = true;
        }
    }
    vo_ctx *c = new (std::nothrow) vo_ctx();
    if (!c) { vo_set_error("vo_ctx_create: out of host memory"); return VO_ENOMEM; }
    c->device = device;
    c->stream = (hipStream_t)stream;
    VO_HIP_CHECK(hipEventCreate(&c->ev0));
    VO_HIP_CHECK(hipEventCreate(&c->ev1));
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        live_set().insert(c);
    }
    *out = c;
    return VO_OK;
}

extern "C" int vo_ctx_create_own_stream(int device, vo_ctx **out)
{
    VO_REQUIRE(out != nullptr, "vo_ctx_create_own_stream: out is NULL");
    *out = nullptr;
    int count = 0;
    VO_HIP_CHECK(hipGetDeviceCount(&count));
    VO_REQUIRE(device >= 0 && device < count, "vo_ctx_create_own_stream: device %d out of range (%d visible)", device, count);
    VO_HIP_CHECK(hipSetDevice(device));
    hipStream_t stream = nullptr;
    // non-blocking: no implicit ordering with the device's default stream, which the default context works on
    VO_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    int rc = vo_ctx_create(device, stream, out);
    if (rc) { (void)hipStreamDestroy(stream); return rc; }
    (*out)->owns_stream = true;
    return VO_OK;
}

extern "C" int vo_ctx_destroy(vo_ctx *ctx)
{
    if (!ctx) return VO_OK;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (!live_set().erase(ctx)) return VO_OK;  // already destroyed
    }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->stage)
        if (b.ptr) (void)hipFree(b.ptr);
    if (ctx->ba_pool) (void)hipFree(ctx->ba_pool);
    vo_frame_batch_free(ctx);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->up_ring) (void)hipHostFree(ctx->up_ring);
    if (ctx->res_area) (void)hipHostFree(ctx->res_area);
    if (ctx->prof0) (void)hipEventDestroy(ctx->prof0);
    if (ctx->prof1) (void)hipEventDestroy(ctx->prof1);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VO_OK;
}

extern "C" int vo_sync(vo_ctx *ctx)
{
    VO_REQUIRE(ctx != nullptr, "vo_sync: ctx is NULL");
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return VO_OK;
}

extern "C" int vo_timer_start(vo_ctx *ctx)
{
    VO_REQUIRE(ctx != nullptr, "vo_timer_start: ctx is NULL");
    VO_HIP_CHECK(hipEventRecord(ctx->ev0, ctx->stream));
    return VO_OK;
}

extern "C" int vo_timer_stop_ms(vo_ctx *ctx, float *ms)
{
    VO_REQUIRE(ctx != nullptr && ms != nullptr, "vo_timer_stop_ms: NULL argument");
    VO_HIP_CHECK(hipEventRecord(ctx->ev1, ctx->stream));
    VO_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    VO_HIP_CHECK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VO_OK;
}

// ---- plain device buffers for callers that bring no tensor library --------------------------------------
extern "C" int vo_dev_alloc(vo_ctx *ctx, size_t bytes, void **out)
{
    VO_REQUIRE(ctx && out, "vo_dev_alloc: NULL argument");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    *out = nullptr;
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    if (e != hipSuccess) { vo_set_error("vo_dev_alloc: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return VO_ENOMEM; }
    return VO_OK;
}

extern "C" int vo_dev_free(vo_ctx *ctx, void *ptr)
{
    VO_REQUIRE(ctx != nullptr, "vo_dev_free: ctx is NULL");
    if (!ptr) return VO_OK;
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));       // queued work of this context may still use the buffer
    VO_HIP_CHECK(hipFree(ptr));
    return VO_OK;
}

extern "C" int vo_dev_upload(vo_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes)
{
    VO_REQUIRE(ctx && (bytes == 0 || (dev_dst && host_src)), "vo_dev_upload: NULL argument");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    return vo_upload(ctx, dev_dst, host_src, bytes);
}

extern "C" int vo_dev_download(vo_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    VO_REQUIRE(ctx && (bytes == 0 || (host_dst && dev_src)), "vo_dev_download: NULL argument");
    VO_HIP_CHECK(hipSetDevice(ctx->device));
    VO_RESULT(ctx, host_dst, dev_src, bytes);
    VO_RESULTS_FINISH(ctx);
    return VO_OK;
}

extern "C" int vo_profile_tracker(vo_ctx *ctx, int enable)
{
    VO_REQUIRE(ctx != nullptr, "vo_profile_tracker: ctx is NULL");
    if (enable && !ctx->prof0) {
        VO_HIP_CHECK(hipSetDevice(ctx->device));
        VO_HIP_CHECK(hipEventCreate(&ctx->prof0));
        VO_HIP_CHECK(hipEventCreate(&ctx->prof1));
    }
    ctx->prof_tracker = enable != 0;
    return VO_OK;
}

extern "C" int vo_profile_tracker_ms(vo_ctx *ctx, float *ms)
{
    VO_REQUIRE(ctx && ms, "vo_profile_tracker_ms: NULL argument");
    if (!ctx->prof_recorded) { vo_set_error("vo_profile_tracker_ms: no tracker launch was timed on this context"); return VO_ESTATE; }
    VO_HIP_CHECK(hipEventSynchronize(ctx->prof1));
    VO_HIP_CHECK(hipEventElapsedTime(ms, ctx->prof0, ctx->prof1));
    return VO_OK;
}

int vo_stage_reserve(vo_ctx *ctx, int slot, size_t bytes, void **out)
{
    vo_devbuf &b = ctx->stage[slot];
    if (bytes == 0) bytes = 16;
    if (b.cap < bytes) {
        if (b.ptr) {
            // a previous asynchronous call may still read the old buffer
            VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            VO_HIP_CHECK(hipFree(b.ptr));
            b.ptr = nullptr;
            b.cap = 0;
        }
        // a frame loop's requests grow slowly (a bundle-adjustment window gains a few thousand observations per
        // keyframe): double, and never start below 16 MB, so that a run regrows a handful of times in all -- each
        // regrowth is a stream synchronisation, a hipFree and a hipMalloc, about a millisecond
        size_t cap = 2 * bytes;
        if (cap < ((size_t)16 << 20)) cap = (size_t)16 << 20;
        hipError_t e = hipMalloc(&b.ptr, cap);
        if (e != hipSuccess) {
            vo_set_error("device allocation of %zu bytes failed: %s", cap, hipGetErrorString(e));
            return VO_ENOMEM;
        }
        b.cap = cap;
    }
    *out = b.ptr;
    return VO_OK;
}

int vo_pinned_reserve(vo_ctx *ctx, size_t bytes, void **out)
{
    if (bytes < 4096) bytes = 4096;
    if (ctx->pinned_cap < bytes) {
        if (ctx->pinned) {
            VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));     // an enqueued copy may still target the old buffer
            VO_HIP_CHECK(hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_cap = 0;
        }
        size_t cap = 2 * bytes;                                  // grows as the stage buffers do
        if (cap < ((size_t)4 << 20)) cap = (size_t)4 << 20;
        hipError_t e = hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault);
        if (e != hipSuccess) {
            vo_set_error("page-locked host allocation of %zu bytes failed: %s", cap, hipGetErrorString(e));
            return VO_ENOMEM;
        }
        ctx->pinned_cap = cap;
    }
    *out = ctx->pinned;
    return VO_OK;
}

struct CopySegs { uint32_t *dst[3]; const uint32_t *src[3]; unsigned words[3]; };
__global__ __launch_bounds__(256) void copy_pinned_kernel(CopySegs s)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < s.words[0]) s.dst[0][i] = s.src[0][i];
    else if (i - s.words[0] < s.words[1]) s.dst[1][i - s.words[0]] = s.src[1][i - s.words[0]];
    else if (i - s.words[0] - s.words[1] < s.words[2]) s.dst[2][i - s.words[0] - s.words[1]] = s.src[2][i - s.words[0] - s.words[1]];
}

int vo_copy_pinned_async(vo_ctx *ctx, const vo_copy_seg *segs, int count)
{
    VO_REQUIRE(count >= 1 && count <= 3, "vo_copy_pinned_async: %d segments", count);
    CopySegs s = {};
    size_t total = 0;
    for (int k = 0; k < count; ++k) {
        VO_REQUIRE(segs[k].bytes % 4 == 0 && segs[k].bytes < (1ull << 31), "vo_copy_pinned_async: segment of %zu bytes", segs[k].bytes);
        s.dst[k] = (uint32_t *)segs[k].dst;
        s.src[k] = (const uint32_t *)segs[k].src;
        s.words[k] = (unsigned)(segs[k].bytes / 4);
        total += s.words[k];
    }
    if (total == 0) return VO_OK;
    hipLaunchKernelGGL(copy_pinned_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, s);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

__global__ __launch_bounds__(256) void copy_pinned_wide_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, unsigned n16)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

int vo_copy_pinned_wide_async(vo_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    VO_REQUIRE(((uintptr_t)dst & 15) == 0 && ((uintptr_t)src & 15) == 0 && bytes < (1ull << 34), "vo_copy_pinned_wide_async: bad block");
    const unsigned n16 = (unsigned)((bytes + 15) / 16);
    if (n16 == 0) return VO_OK;
    hipLaunchKernelGGL(copy_pinned_wide_kernel, dim3((n16 + 255) / 256), dim3(256), 0, ctx->stream, (uint4 *)dst, (const uint4 *)src, n16);
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

// bulk in 16-byte pieces, the last bytes one by one (dst 16-byte aligned; src is, the ring hands out 256-byte aligned regions)
__global__ __launch_bounds__(256) void upload_kernel(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src,
                                                     unsigned n16, unsigned tail)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
    else if (i - n16 < tail) dst[(size_t)n16 * 16 + (i - n16)] = src[(size_t)n16 * 16 + (i - n16)];
}
__global__ __launch_bounds__(256) void upload_bytes_kernel(unsigned char *__restrict__ dst, const unsigned char *__restrict__ src, unsigned n)
{
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// The ring is kept small on purpose: its lines then stay in the CPU's cache from one upload to the next.  memcpy into
// a region that has fallen out of the cache first reads every line from DRAM; with a 16 MiB ring two frame-sized
// uploads cost 0.2 ms more than with 2 MiB (PoseEstimation.KLT: 0.33 against 0.13 ms).
#define VO_UPLOAD_RING_BYTES ((size_t)2 << 20)
#define VO_UPLOAD_MAX_BYTES ((size_t)1 << 20)            // one piece of the ring up to here
#define VO_UPLOAD_CHUNKED_MAX_BYTES ((size_t)16 << 20)   // several pieces up to here, the runtime's copy beyond

// A 256-byte aligned region of the ring; waits for the stream before going round.
static int upload_region(vo_ctx *ctx, size_t bytes, char **out)
{
    if (!ctx->up_ring) {
        hipError_t e = hipHostMalloc((void **)&ctx->up_ring, VO_UPLOAD_RING_BYTES, hipHostMallocDefault);
        if (e != hipSuccess) { vo_set_error("page-locked allocation of the upload ring failed: %s", hipGetErrorString(e)); return VO_ENOMEM; }
        ctx->up_cap = VO_UPLOAD_RING_BYTES;
        ctx->up_off = 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (ctx->up_off + need > ctx->up_cap) {
        VO_HIP_CHECK(hipStreamSynchronize(ctx->stream));        // every upload out of the ring has run
        ctx->up_off = 0;
    }
    *out = ctx->up_ring + ctx->up_off;
    ctx->up_off += need;
    return VO_OK;
}

static int upload_launch(vo_ctx *ctx, void *dst, const char *ring, size_t bytes)
{
    if (((uintptr_t)dst & 15) == 0) {
        const unsigned n16 = (unsigned)(bytes / 16), tail = (unsigned)(bytes % 16), threads = n16 + tail;
        hipLaunchKernelGGL(upload_kernel, dim3((threads + 255) / 256), dim3(256), 0, ctx->stream, (unsigned char *)dst,
                           (const unsigned char *)ring, n16, tail);
    } else {
        hipLaunchKernelGGL(upload_bytes_kernel, dim3((unsigned)((bytes + 255) / 256)), dim3(256), 0, ctx->stream, (unsigned char *)dst,
                           (const unsigned char *)ring, (unsigned)bytes);
    }
    VO_HIP_CHECK(hipGetLastError());
    return VO_OK;
}

int vo_upload(vo_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return VO_OK;
    if (bytes > VO_UPLOAD_CHUNKED_MAX_BYTES) {
        VO_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return VO_OK;
    }
    // Up to a few ring sizes go through the ring in pieces (the stream is waited for when the ring goes round, by when
    // the earlier pieces' copies have long run): hipMemcpyAsync from pageable memory is staged by the runtime, which
    // took 1.2 ms for the 3.3 MB of a 140 000-observation window and 6.7 ms the first time it was used.
    const size_t piece = VO_UPLOAD_RING_BYTES / 4;
    for (size_t at = 0; at < bytes;) {
        const size_t now = bytes > VO_UPLOAD_MAX_BYTES ? (bytes - at < piece ? bytes - at : piece) : bytes;
        char *ring;
        int rc = upload_region(ctx, now, &ring);
        if (rc) return rc;
        memcpy(ring, (const char *)src + at, now);
        if ((rc = upload_launch(ctx, (char *)dst + at, ring, now))) return rc;
        at += now;
    }
    return VO_OK;
}

int vo_upload_2d(vo_ctx *ctx, void *dst, const void *src, size_t src_stride, size_t width, size_t height)
{
    if (src_stride == width) return vo_upload(ctx, dst, src, width * height);
    const size_t bytes = width * height;
    if (bytes == 0) return VO_OK;
    if (bytes > VO_UPLOAD_MAX_BYTES) {
        VO_HIP_CHECK(hipMemcpy2DAsync(dst, width, src, src_stride, width, height, hipMemcpyHostToDevice, ctx->stream));
        return VO_OK;
    }
    char *ring;
    int rc = upload_region(ctx, bytes, &ring);
    if (rc) return rc;
    for (size_t y = 0; y < height; ++y) memcpy(ring + y * width, (const char *)src + y * src_stride, width);
    return upload_launch(ctx, dst, ring, bytes);
}

int vo_download_async(vo_ctx *ctx, void *pinned_dst, const void *dev_src, size_t bytes)
{
    if (bytes == 0) return VO_OK;
    if ((((uintptr_t)pinned_dst | (uintptr_t)dev_src) & 15) != 0 || bytes > ((size_t)4 << 20)) {
        VO_HIP_CHECK(hipMemcpyAsync(pinned_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return VO_OK;
    }
    return vo_copy_pinned_wide_async(ctx, pinned_dst, dev_src, bytes);
}

#define VO_RESULT_AREA_BYTES ((size_t)2 << 20)
#define VO_RESULT_MAX_BYTES ((size_t)1 << 20)

int vo_result(vo_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    if (bytes == 0) return VO_OK;
    if (!ctx->res_area) {
        hipError_t e = hipHostMalloc((void **)&ctx->res_area, VO_RESULT_AREA_BYTES, hipHostMallocDefault);
        if (e != hipSuccess) { vo_set_error("page-locked allocation of the result area failed: %s", hipGetErrorString(e)); return VO_ENOMEM; }
        ctx->res_off = 0;
        ctx->res_count = 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (bytes > VO_RESULT_MAX_BYTES || ctx->res_count == 16 || ctx->res_off + need > VO_RESULT_AREA_BYTES) {
        VO_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        return VO_OK;
    }
    char *region = ctx->res_area + ctx->res_off;
    int rc = vo_download_async(ctx, region, dev_src, bytes);
    if (rc) return rc;
    ctx->res_pending[ctx->res_count].dst = host_dst;
    ctx->res_pending[ctx->res_count].src = region;
    ctx->res_pending[ctx->res_count].bytes = bytes;
    ++ctx->res_count;
    ctx->res_off += need;
    return VO_OK;
}

int vo_results_finish(vo_ctx *ctx)
{
    hipError_t e = hipStreamSynchronize(ctx->stream);
    const int count = ctx->res_count;
    ctx->res_count = 0;
    ctx->res_off = 0;
    if (e != hipSuccess) { vo_set_error("hipStreamSynchronize failed: %s", hipGetErrorString(e)); return VO_EHIP; }
    for (int k = 0; k < count; ++k) memcpy(ctx->res_pending[k].dst, ctx->res_pending[k].src, ctx->res_pending[k].bytes);
    return VO_OK;
}

void vo_results_abandon(vo_ctx *ctx)
{
    ctx->res_count = 0;
    ctx->res_off = 0;
}
