// clock_rate.hip -- what wall_clock64() ticks at on this device (attribute and measured against HIP events)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    out[0] = wall_clock64() - t0;
}
int main()
{
    int khz = 0;
    (void)hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    printf("hipDeviceAttributeWallClockRate = %d kHz\n", khz);
    unsigned long long *d; (void)hipMalloc(&d, 8);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (unsigned long long ticks : {100000ull, 1000000ull}) {
        k_spin<<<1, 64>>>(ticks, d); (void)hipDeviceSynchronize();
        (void)hipEventRecord(a); k_spin<<<1, 64>>>(ticks, d); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%llu ticks took %.3f ms -> %.1f MHz\n", ticks, ms, ticks / ms / 1000.0);
    }
    return 0;
}
