// atomic_rate.hip -- calibration: scattered 32-bit atomics over a table, returning vs non-returning, against plain gathers and
// scatters of the same shape.  Prices the k-truss / k-core peel's decrements.
//   hipcc --offload-arch=gfx950 -O3 atomic_rate.hip -o atomic_rate && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint64_t i)
{
    uint64_t x = i * 0x9E3779B97F4A7C15ull;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return (uint32_t)x;
}

template <int MODE>   // 0 returning sub, 1 non-returning sub, 2 gather, 3 scatter store, 4 gather + returning sub on hit of (value & 1), 5 returning sub, 2 independent per lane
__global__ __launch_bounds__(256) void k_ops(int32_t *t, uint32_t n, uint64_t ops, unsigned long long *sink)
{
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ops; i += stride) {
        const uint32_t idx = mix(i) % n;
        if (MODE == 0) acc += (uint32_t)atomicSub(&t[idx], 1);
        else if (MODE == 1) atomicSub(&t[idx], 1);
        else if (MODE == 2) acc += (uint32_t)t[idx];
        else if (MODE == 3) t[idx] = (int32_t)i;
        else if (MODE == 4) { const int32_t v = t[idx]; if (v & 1) acc += (uint32_t)atomicSub(&t[idx], 2); else acc += 1; }
        else if (MODE == 5) {
            const uint32_t idx2 = mix(i + ops) % n;
            const uint32_t a = (uint32_t)atomicSub(&t[idx], 1), b = (uint32_t)atomicSub(&t[idx2], 1);
            acc += a + b;
        }
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1ull);
}

int main()
{
    const uint64_t ops = 177ull << 20;
    unsigned long long *sink = nullptr;
    CK(hipMalloc(&sink, 8));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const char *names[6] = {"returning atomicSub", "non-returning atomicSub", "plain 4-byte gather", "plain 4-byte scatter store", "gather, then returning sub on half", "two returning subs per lane per trip"};
    for (uint32_t n : {100u << 20, 25u << 20, 1u << 20}) {
        int32_t *t = nullptr;
        CK(hipMalloc(&t, (size_t)n * 4));
        for (int grid : {256 * 8, 256 * 4}) {                // 8 and 4 workgroups of 256 threads per CU
            const int block = 256;
            for (int mode = 0; mode < 6; ++mode) {
                CK(hipMemset(t, 0x55, (size_t)n * 4));
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(a));
                    const uint64_t o = mode == 5 ? ops / 2 : ops;
                    switch (mode) {
                    case 0: hipLaunchKernelGGL(k_ops<0>, grid, block, 0, 0, t, n, o, sink); break;
                    case 1: hipLaunchKernelGGL(k_ops<1>, grid, block, 0, 0, t, n, o, sink); break;
                    case 2: hipLaunchKernelGGL(k_ops<2>, grid, block, 0, 0, t, n, o, sink); break;
                    case 3: hipLaunchKernelGGL(k_ops<3>, grid, block, 0, 0, t, n, o, sink); break;
                    case 4: hipLaunchKernelGGL(k_ops<4>, grid, block, 0, 0, t, n, o, sink); break;
                    default: hipLaunchKernelGGL(k_ops<5>, grid, block, 0, 0, t, n, o, sink); break;
                    }
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf("table %4u M words, grid %4d x %4d: %-40s %8.3f ms for %llu M ops = %7.1f G ops/s\n", n >> 20, grid, block, names[mode], best,
                       (unsigned long long)(ops >> 20), (double)ops / best / 1e6);
            }
        }
        CK(hipFree(t));
    }
    return 0;
}
