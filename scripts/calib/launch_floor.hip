// launch_floor.hip -- what one dependent step of a device-driven loop costs on gfx950, piece by piece.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/calib/launch_floor.hip -o gpurun_out/launch_floor
// Every variant is launched N times back to back on one stream (pre-queued, like the peel's blind batches)
// and timed with one pair of events: microseconds per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}

// every workgroup reads a control word, leaves if told so (the "done" path of the peel)
__global__ void k_read_ctrl(const int *ctrl) { if (ctrl[0]) return; }

// control word + two-level arrival ticket; the last workgroup bumps a counter (the peel's step skeleton)
__global__ void k_ticket(int *ctrl, unsigned *grp, unsigned *top)
{
    __shared__ int s;
    if (threadIdx.x == 0) s = ctrl[0];
    __syncthreads();
    if (s) return;
    if (threadIdx.x != 0) return;
    const unsigned g = blockIdx.x / 32, ng = (gridDim.x + 31) / 32;
    const unsigned gs = (g == ng - 1) ? gridDim.x - g * 32 : 32u;
    if (atomicAdd(&grp[g], 1u) == gs - 1) {
        atomicExch(&grp[g], 0u);
        if (ng == 1 || atomicAdd(top, 1u) == ng - 1) { atomicExch(top, 0u); ctrl[1] += 1; }
    }
}

// skeleton + a chain of `hops` dependent global loads per wave (cold after the kernel boundary?)
__global__ void k_chain(int *ctrl, unsigned *grp, unsigned *top, const unsigned *next, int hops, unsigned *sink)
{
    __shared__ int s;
    if (threadIdx.x == 0) s = ctrl[0];
    __syncthreads();
    if (s) return;
    unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    for (int h = 0; h < hops; ++h) p = next[p];
    if (p == 0xFFFFFFFFu) *sink = p;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const unsigned g = blockIdx.x / 32, ng = (gridDim.x + 31) / 32;
    const unsigned gs = (g == ng - 1) ? gridDim.x - g * 32 : 32u;
    if (atomicAdd(&grp[g], 1u) == gs - 1) {
        atomicExch(&grp[g], 0u);
        if (ng == 1 || atomicAdd(top, 1u) == ng - 1) { atomicExch(top, 0u); ctrl[1] += 1; }
    }
}

// same chain, but with returning atomics instead of loads (what a decrement costs)
__global__ void k_chain_atomic(int *ctrl, unsigned *tbl, int hops, unsigned *sink)
{
    __shared__ int s;
    if (threadIdx.x == 0) s = ctrl[0];
    __syncthreads();
    if (s) return;
    unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    for (int h = 0; h < hops; ++h) p = atomicAdd(&tbl[p], 0u);
    if (p == 0xFFFFFFFFu) *sink = p;
}

int main()
{
    const int N = 2000;
    int *ctrl; unsigned *grp, *top, *next, *sink;
    const size_t tbl = 64u << 20;                         // 256 MiB of uint32: chains stride through it
    CK(hipMalloc(&ctrl, 64)); CK(hipMalloc(&grp, 4096)); CK(hipMalloc(&top, 64)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&next, tbl * 4));
    CK(hipMemset(ctrl, 0, 64)); CK(hipMemset(grp, 0, 4096)); CK(hipMemset(top, 0, 64));
    {
        std::vector<unsigned> h(tbl);
        for (size_t i = 0; i < tbl; ++i) h[i] = (unsigned)((i * 2654435761ull + 12345u) % tbl) & ~15u;
        CK(hipMemcpy(next, h.data(), tbl * 4, hipMemcpyHostToDevice));
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto &&launch) {
        for (int i = 0; i < 50; ++i) launch();
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        for (int i = 0; i < N; ++i) launch();
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-52s %7.2f us/launch\n", name, ms * 1000.f / N);
    };
    timeit("empty, 1 x 64", [&] { k_empty<<<1, 64, 0, s>>>(); });
    timeit("empty, 1 x 1024", [&] { k_empty<<<1, 1024, 0, s>>>(); });
    timeit("empty, 256 x 1024", [&] { k_empty<<<256, 1024, 0, s>>>(); });
    timeit("read ctrl, 256 x 1024", [&] { k_read_ctrl<<<256, 1024, 0, s>>>(ctrl); });
    timeit("ctrl + ticket, 1 x 1024", [&] { k_ticket<<<1, 1024, 0, s>>>(ctrl, grp, top); });
    timeit("ctrl + ticket, 32 x 1024", [&] { k_ticket<<<32, 1024, 0, s>>>(ctrl, grp, top); });
    timeit("ctrl + ticket, 256 x 1024", [&] { k_ticket<<<256, 1024, 0, s>>>(ctrl, grp, top); });
    for (int hops : {1, 2, 4, 8}) {
        char nm[96];
        snprintf(nm, sizeof nm, "ctrl + %d dependent loads + ticket, 256 x 1024", hops);
        timeit(nm, [&] { k_chain<<<256, 1024, 0, s>>>(ctrl, grp, top, next, hops, sink); });
        snprintf(nm, sizeof nm, "ctrl + %d dependent loads + ticket, 8 x 1024", hops);
        timeit(nm, [&] { k_chain<<<8, 1024, 0, s>>>(ctrl, grp, top, next, hops, sink); });
    }
    for (int hops : {1, 4}) {
        char nm[96];
        snprintf(nm, sizeof nm, "ctrl + %d dependent returning atomics, 256 x 1024", hops);
        timeit(nm, [&] { k_chain_atomic<<<256, 1024, 0, s>>>(ctrl, next, hops, sink); });
    }
    // the same 24-launch batch as a hipGraph (what the peel driver could replay instead of 24 launches)
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 24; ++i) k_ticket<<<256, 1024, 0, s>>>(ctrl, grp, top);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 4; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        for (int i = 0; i < N / 24; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        printf("%-52s %7.2f us/launch\n", "ctrl + ticket, 256 x 1024, hipGraph of 24", ms * 1000.f / (N / 24 * 24));
        hipGraph_t g2; hipGraphExec_t ge2;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 24; ++i) k_empty<<<256, 1024, 0, s>>>();
        CK(hipStreamEndCapture(s, &g2));
        CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int i = 0; i < 4; ++i) (void)hipGraphLaunch(ge2, s);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(a, s);
        for (int i = 0; i < N / 24; ++i) (void)hipGraphLaunch(ge2, s);
        (void)hipEventRecord(b, s);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%-52s %7.2f us/launch\n", "empty, 256 x 1024, hipGraph of 24", ms * 1000.f / (N / 24 * 24));
    }
    int one = 1;
    CK(hipMemcpy(ctrl, &one, 4, hipMemcpyHostToDevice));
    timeit("done path (ctrl says stop), 256 x 1024", [&] { k_ticket<<<256, 1024, 0, s>>>(ctrl, grp, top); });
    return 0;
}
