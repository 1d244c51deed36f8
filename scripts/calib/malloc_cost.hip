// hipMalloc / hipFree cost by size, first and repeated (build: hipcc --offload-arch=gfx950 -O2 malloc_cost.hip -o malloc_cost)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double ms(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
__global__ void touch(char *p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x * 4096) p[i * 1] = 1; }
int main()
{
    hipFree(0);
    for (int rep = 0; rep < 2; ++rep)
        for (size_t gb : {1, 4, 16, 32}) {
            void *p = nullptr;
            auto t0 = std::chrono::steady_clock::now();
            hipError_t e = hipMalloc(&p, gb << 30);
            double a = ms(t0);
            t0 = std::chrono::steady_clock::now();
            hipMemsetAsync(p, 0, 64 << 20, 0); hipDeviceSynchronize();
            double b = ms(t0);
            t0 = std::chrono::steady_clock::now();
            hipFree(p);
            double c = ms(t0);
            printf("rep %d: %2zu GB: hipMalloc %.1f ms (%s), first 64 MB memset %.2f ms, hipFree %.1f ms\n", rep, gb, a, hipGetErrorString(e), b, c);
        }
    // many blocks kept, as the pool does
    auto t0 = std::chrono::steady_clock::now();
    void *q[24];
    for (int i = 0; i < 24; ++i) hipMalloc(&q[i], 1ull << 30);
    printf("24 x 1 GB kept: %.1f ms\n", ms(t0));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 24; ++i) hipFree(q[i]);
    printf("24 x hipFree: %.1f ms\n", ms(t0));
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 24; ++i) hipMalloc(&q[i], 1ull << 30);
    printf("24 x 1 GB again: %.1f ms\n", ms(t0));
    return 0;
}
