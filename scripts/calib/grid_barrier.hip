// grid_barrier.hip -- what a sub-round boundary would cost INSIDE one persistent launch: R rounds of an
// arrive-and-wait barrier among G co-resident workgroups (one per CU at most), each round also doing `hops`
// dependent coherent loads like a peel step does.  Spins are bounded: a barrier that does not complete sets an
// error flag and everyone leaves, so the kernel cannot hang.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/calib/grid_barrier.hip -o /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(1024) void k_rounds(unsigned *arrive, unsigned *gen, unsigned *err, const unsigned *next, int rounds, int hops,
                                                 unsigned *sink)
{
    __shared__ unsigned s_gen;
    unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
    unsigned my_gen = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int h = 0; h < hops; ++h) p = __hip_atomic_load(&next[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 0) {
            ++my_gen;
            if (atomicAdd(arrive, 1u) == gridDim.x - 1) {            // last arrival: reset, release the others
                atomicExch(arrive, 0u);
                __hip_atomic_store(gen, my_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                int spins = 0;
                while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < my_gen) {
                    if (++spins > 2000000 || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicExch(err, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            s_gen = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_gen) return;
    }
    if (p == 0xFFFFFFFFu) *sink = p;
}

int main()
{
    unsigned *ctl, *next, *sink;
    const size_t tbl = 16u << 20;
    (void)hipMalloc(&ctl, 256); (void)hipMalloc(&sink, 64); (void)hipMalloc(&next, tbl * 4);
    {
        std::vector<unsigned> h(tbl);
        for (size_t i = 0; i < tbl; ++i) h[i] = (unsigned)((i * 2654435761ull + 12345u) % tbl) & ~15u;
        (void)hipMemcpy(next, h.data(), tbl * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int rounds = 2000;
    for (int hops : {0, 4}) {
        for (int G : {8, 32, 128, 256}) {
            (void)hipMemset(ctl, 0, 256);
            k_rounds<<<G, 1024>>>(ctl, ctl + 16, ctl + 32, next, 50, hops, sink);      // warm
            (void)hipDeviceSynchronize();
            (void)hipMemset(ctl, 0, 256);
            (void)hipEventRecord(a);
            k_rounds<<<G, 1024>>>(ctl, ctl + 16, ctl + 32, next, rounds, hops, sink);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
            unsigned e = 0; (void)hipMemcpy(&e, ctl + 32, 4, hipMemcpyDeviceToHost);
            printf("G = %3d workgroups, %d dependent loads per round: %6.2f us per round%s\n", G, hops, ms * 1000.f / rounds, e ? "  (BARRIER TIMED OUT)" : "");
        }
    }
    return 0;
}
