// Host-to-device upload of one frame: pageable hipMemcpyAsync + wait against memcpy into page-locked memory
// (coherent / non-coherent) + asynchronous copy; prints microseconds per variant.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 1241 * 376;
    unsigned char *src = (unsigned char *)malloc(n), *dev, *pinc, *pinn;
    memset(src, 7, n);
    hipMalloc((void **)&dev, n);
    hipHostMalloc((void **)&pinc, n, hipHostMallocDefault);
    hipHostMalloc((void **)&pinn, n, hipHostMallocNonCoherent);
    hipStream_t s = nullptr;
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        for (int i = 0; i < 200; ++i) { hipMemcpyAsync(dev, src, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
        double t1 = now();
        for (int i = 0; i < 200; ++i) { memcpy(pinc, src, n); }
        double t2 = now();
        for (int i = 0; i < 200; ++i) { memcpy(pinn, src, n); }
        double t3 = now();
        for (int i = 0; i < 200; ++i) { memcpy(pinc, src, n); hipMemcpyAsync(dev, pinc, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
        double t4 = now();
        for (int i = 0; i < 200; ++i) { memcpy(pinn, src, n); hipMemcpyAsync(dev, pinn, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); }
        double t5 = now();
        for (int i = 0; i < 200; ++i) { hipMemcpyAsync(dev, src, n, hipMemcpyHostToDevice, s); }
        double t6 = now();
        hipStreamSynchronize(s);
        printf("pageable+sync %.1f | memcpy->coherent %.1f | memcpy->noncoherent %.1f | coherent+async+sync %.1f | noncoherent+async+sync %.1f | pageable async call only %.1f us\n",
               (t1 - t0) / 200, (t2 - t1) / 200, (t3 - t2) / 200, (t4 - t3) / 200, (t5 - t4) / 200, (t6 - t5) / 200);
    }
    return 0;
}
