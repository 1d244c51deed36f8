// roundtrip.hip -- calibration: what a small device-to-host read costs between two dependent kernels.
//   (a) hipMemcpyAsync into pinned memory + hipStreamSynchronize (the library's d2h helper);
//   (b) a one-wavefront kernel that stores the words into MAPPED pinned memory, fences and raises a sequence word the host spins on.
// The GPU-side gap is what matters: kernel A ... [read its result on the host] ... kernel B; measured with HIP events around A..B.
//   hipcc --offload-arch=gfx950 -O3 roundtrip.hip -o roundtrip && ./roundtrip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_work(uint32_t *out, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v; }
__global__ void k_post(const uint32_t *src, int words, volatile uint32_t *host_dst, volatile uint32_t *host_seq, uint32_t seq)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) host_dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { *host_seq = seq; __threadfence_system(); }
}

int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    uint32_t *d = nullptr; CK(hipMalloc(&d, 4096));
    uint32_t *h_pin = nullptr; CK(hipHostMalloc(&h_pin, 4096, hipHostMallocDefault));
    uint32_t *h_map = nullptr; CK(hipHostMalloc(&h_map, 8192, hipHostMallocMapped));
    uint32_t *d_map = nullptr; CK(hipHostGetDevicePointer((void **)&d_map, h_map, 0));
    volatile uint32_t *h_seq = h_map + 1024;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 200;
    for (int words : {4, 32, 1024}) {
        for (int mode = 0; mode < 2; ++mode) {
            // warm
            for (int w = 0; w < 5; ++w) { hipLaunchKernelGGL(k_work, 64, 256, 0, s, d, 1u); CK(hipStreamSynchronize(s)); }
            *h_seq = 0;
            float gpu_ms = 0.f;
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(a, s));
            for (int r = 1; r <= reps; ++r) {
                hipLaunchKernelGGL(k_work, 64, 256, 0, s, d, (uint32_t)r);
                if (mode == 0) {
                    CK(hipMemcpyAsync(h_pin, d, (size_t)words * 4, hipMemcpyDeviceToHost, s));
                    CK(hipStreamSynchronize(s));
                    if (h_pin[0] != (uint32_t)r) { printf("bad value\n"); return 1; }
                } else {
                    hipLaunchKernelGGL(k_post, 1, 64, 0, s, d, words, (volatile uint32_t *)d_map, (volatile uint32_t *)(d_map + 1024), (uint32_t)r);
                    long spins = 0;
                    while (*h_seq != (uint32_t)r) { if (++spins > 200000000L) { printf("timeout\n"); return 1; } }
                    if (h_map[0] != (uint32_t)r) { printf("bad value (mapped)\n"); return 1; }
                }
            }
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&gpu_ms, a, b));
            const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("%4d words, %-46s %7.2f us per round trip (GPU timeline), %7.2f us wall\n", words,
                   mode == 0 ? "memcpyAsync to pinned + hipStreamSynchronize:" : "post kernel to mapped memory + host spin:", gpu_ms * 1000.0 / reps, wall / reps);
        }
    }
    return 0;
}
