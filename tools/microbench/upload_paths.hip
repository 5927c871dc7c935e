// One frame to the device followed by a kernel that reads it and a small read-back, as a drop-in call does:
// (a) pageable hipMemcpyAsync, (b) memcpy into one page-locked buffer + copy kernel, (c) the same through a 16 MiB ring,
// (d) through a 2 MiB ring, (e) page-locked buffer + hipMemcpyAsync.  Microseconds per call.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void copy16(uint4 *d, const uint4 *s, unsigned n) { unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = s[i]; }
__global__ void consume(const unsigned char *d, unsigned n, unsigned *out) { unsigned i = blockIdx.x * 256 + threadIdx.x; if (i < n && d[i] == 255) atomicAdd(out, 1u); }
int main()
{
    const size_t n = 1241 * 376, n16 = (n + 15) / 16;
    unsigned char *src = (unsigned char *)malloc(n), *dev, *one, *ring;
    unsigned *dout, *hout;
    memset(src, 7, n);
    CK(hipMalloc((void **)&dev, n + 256));
    CK(hipMalloc((void **)&dout, 64));
    CK(hipHostMalloc((void **)&hout, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&one, n + 256, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&ring, 16 << 20, hipHostMallocDefault));
    hipStream_t s = nullptr;
    const int R = 300;
    for (int rep = 0; rep < 3; ++rep) {
        double t[6];
        t[0] = now();
        for (int i = 0; i < R; ++i) {
            CK(hipMemcpyAsync(dev, src, n, hipMemcpyHostToDevice, s));
            consume<<<(n + 255) / 256, 256, 0, s>>>(dev, n, dout);
            CK(hipMemcpyAsync(hout, dout, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        }
        t[1] = now();
        for (int i = 0; i < R; ++i) {
            memcpy(one, src, n);
            copy16<<<(n16 + 255) / 256, 256, 0, s>>>((uint4 *)dev, (const uint4 *)one, n16);
            consume<<<(n + 255) / 256, 256, 0, s>>>(dev, n, dout);
            CK(hipMemcpyAsync(hout, dout, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        }
        t[2] = now();
        size_t off = 0;
        for (int i = 0; i < R; ++i) {
            if (off + n + 256 > (16u << 20)) off = 0;
            memcpy(ring + off, src, n);
            copy16<<<(n16 + 255) / 256, 256, 0, s>>>((uint4 *)dev, (const uint4 *)(ring + off), n16);
            off += (n + 255) & ~(size_t)255;
            consume<<<(n + 255) / 256, 256, 0, s>>>(dev, n, dout);
            CK(hipMemcpyAsync(hout, dout, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        }
        t[3] = now();
        off = 0;
        for (int i = 0; i < R; ++i) {
            if (off + n + 256 > (2u << 20)) off = 0;
            memcpy(ring + off, src, n);
            copy16<<<(n16 + 255) / 256, 256, 0, s>>>((uint4 *)dev, (const uint4 *)(ring + off), n16);
            off += (n + 255) & ~(size_t)255;
            consume<<<(n + 255) / 256, 256, 0, s>>>(dev, n, dout);
            CK(hipMemcpyAsync(hout, dout, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        }
        t[4] = now();
        for (int i = 0; i < R; ++i) {
            memcpy(one, src, n);
            CK(hipMemcpyAsync(dev, one, n, hipMemcpyHostToDevice, s));
            consume<<<(n + 255) / 256, 256, 0, s>>>(dev, n, dout);
            CK(hipMemcpyAsync(hout, dout, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        }
        t[5] = now();
        printf("pageable %.1f | one pinned + kernel %.1f | 16 MiB ring + kernel %.1f | 2 MiB ring + kernel %.1f | one pinned + hipMemcpyAsync %.1f us\n",
               (t[1] - t[0]) / R, (t[2] - t[1]) / R, (t[3] - t[2]) / R, (t[4] - t[3]) / R, (t[5] - t[4]) / R);
    }
    return 0;
}
