// How fast can ONE wave per SIMD issue fp64 FMAs?  K independent dependency chains per lane, W waves per SIMD.
// (round 3: decides whether intra-wave ILP can lift the N = 65 536 configuration, where only ~1-2 waves per SIMD exist)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K>
__global__ void __launch_bounds__(256) chains(double* out, double x, double y, int iters)
{
    double a[K];
#pragma unroll
    for (int j = 0; j < K; j++) a[j] = threadIdx.x * 1e-3 + j;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 64 / K; r++)
#pragma unroll
            for (int j = 0; j < K; j++) a[j] = __builtin_fma(a[j], x, y);
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < K; j++) s += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int K>
void run(double* d, int blocks_per_cu)
{
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    chains<K><<<256 * blocks_per_cu, 256>>>(d, 0.999999, 1e-7, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    chains<K><<<256 * blocks_per_cu, 256>>>(d, 0.999999, 1e-7, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 64.0 * iters;   // per wave
    printf("K=%d chains, %d wave(s)/SIMD: %.2f ns per wave-instruction per wave, %.2f ns per instruction per SIMD\n", K, blocks_per_cu,
           ms * 1e6 / insts, ms * 1e6 / insts / blocks_per_cu);
}
int main()
{
    double* d; hipMalloc(&d, sizeof(double) * 256 * 256 * 8);
    for (int w : {1, 2, 4}) { run<1>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); }
    return 0;
}
