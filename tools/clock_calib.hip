// Calibrates __builtin_readcyclecounter() (s_memtime) against HIP events: ticks per microsecond.
// hipcc --offload-arch=gfx950 -O2 tools/clock_calib.hip -o /tmp/clock_calib && /tmp/clock_calib
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t - t0;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (unsigned long long n : {1000000ull, 10000000ull, 100000000ull}) {
        spin<<<1, 64>>>(1000, d);
        hipDeviceSynchronize();
        hipEventRecord(a);
        spin<<<1, 64>>>(n, d);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("%llu ticks in %.3f ms -> %.1f ticks/us\n", n, ms, n / (ms * 1e3));
    }
    return 0;
}
