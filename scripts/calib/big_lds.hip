// big_lds.hip -- launch cost of a single-workgroup kernel by static LDS size
#include <hip/hip_runtime.h>
#include <cstdio>
template <int WORDS>
__global__ __launch_bounds__(1024) void k_lds(unsigned *out)
{
    __shared__ unsigned pool[WORDS];
    pool[threadIdx.x % WORDS] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = pool[(out[1] + 1) % WORDS];
}
template <int WORDS> void run(unsigned *d)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_lds<WORDS><<<1, 1024>>>(d); (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 20; ++i) k_lds<WORDS><<<1, 1024>>>(d);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
    printf("LDS %6d B: %.1f us per launch\n", WORDS * 4, ms * 1000 / 20);
}
int main()
{
    unsigned *d; (void)hipMalloc(&d, 64); (void)hipMemset(d, 0, 64);
    run<1024>(d); run<16000>(d); run<16385>(d); run<24000>(d); run<32768>(d); run<40448>(d);
    return 0;
}
