// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE on gfx950 for the access widths the komb kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/calib/fetch_calib.hip -o gpurun_out/fetch_calib
// Run:   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib -- gpurun_out/fetch_calib
// Each kernel reads a KNOWN number of bytes from a 1 GiB buffer (far beyond the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <class T>
__global__ void k_stream(const T *__restrict__ p, size_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = p[i];
        const unsigned *w = reinterpret_cast<const unsigned *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += w[k];
    }
    if (acc == 0x123456789ull) *out = acc;
}
// random 4-byte gathers: n loads at hashed positions of a table of `tbl` words
__global__ void k_gather4(const unsigned *__restrict__ p, size_t tbl, size_t n, unsigned long long *out)
{
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = (i + 1) * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        acc += p[h % tbl];
    }
    if (acc == 0x123456789ull) *out = acc;
}
int main()
{
    const size_t bytes = 1ull << 30;
    void *buf; unsigned long long *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 8);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    k_stream<unsigned><<<4096, 256>>>((const unsigned *)buf, bytes / 4, out);          // 1 GiB as 4 B/lane
    k_stream<uint2><<<4096, 256>>>((const uint2 *)buf, bytes / 8, out);                // 1 GiB as 8 B/lane
    k_stream<uint4><<<4096, 256>>>((const uint4 *)buf, bytes / 16, out);               // 1 GiB as 16 B/lane
    k_gather4<<<4096, 256>>>((const unsigned *)buf, bytes / 4, 64u << 20, out);        // 64 Mi gathers: 256 MiB of words, 4 GiB of 64-B lines
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
