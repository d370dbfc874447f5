// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the fp64 / integer instructions the
// RHS kernels are made of, on gfx950.  One wave per SIMD or 2/3 waves per SIMD, dependent vs independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int OP, int ILP>
__global__ void __launch_bounds__(256) k(double* out, double seed, int iters)
{
    double a[ILP];
    int ia[ILP];
#pragma unroll
    for (int j = 0; j < ILP; j++) { a[j] = seed + threadIdx.x * 1e-3 + j; ia[j] = threadIdx.x + j; }
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
#pragma unroll
            for (int j = 0; j < ILP; j++) {
                if (OP == 0) a[j] = __builtin_fma(a[j], c1, c2);
                if (OP == 1) a[j] = a[j] + c2;
                if (OP == 2) a[j] = a[j] * c1;
                if (OP == 3) a[j] = __builtin_amdgcn_rcp(a[j]);
                if (OP == 4) a[j] = ldexp(a[j], ia[j] & 1);
                if (OP == 5) a[j] = __builtin_rint(a[j] * c1);
                if (OP == 6) { ia[j] = (int)a[j]; a[j] = a[j] + (double)ia[j]; }
                if (OP == 7) a[j] = (a[j] > c1) ? a[j] : c2;             // cmp + 2 cndmask
                if (OP == 8) ia[j] = (ia[j] >> 3) + (ia[j] & 127);         // 32-bit int ops
                if (OP == 9) a[j] = fmax(a[j], c1);
                if (OP == 10) asm volatile("v_mov_b32 %0, %0" : "+v"(ia[j]));
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int j = 0; j < ILP; j++) s += a[j] + ia[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)(t1 - t0) * 0;
    if (threadIdx.x == 0) ((unsigned long long*)out)[gridDim.x * blockDim.x + blockIdx.x] = t1 - t0;
}

template <int OP, int ILP>
void run(const char* name, int blocks_per_cu)
{
    const int iters = 200, nblk = 256 * blocks_per_cu;
    double* d;
    hipMalloc(&d, sizeof(double) * (nblk * 256 + nblk));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP, ILP><<<nblk, 256>>>(d, 1.0000001, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP, ILP><<<nblk, 256>>>(d, 1.0000001, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> cyc(nblk);
    hipMemcpy(cyc.data(), (unsigned long long*)d + (size_t)nblk * 256, sizeof(unsigned long long) * nblk, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto c : cyc) avg += c;
    avg /= nblk;
    const double ninst = (double)iters * REP * ILP;  // per wave
    // waves per SIMD = blocks_per_cu (each block = 4 waves = 1 per SIMD)
    printf("%-28s ILP=%d waves/SIMD=%d : %6.2f shader-clk per inst per wave, %6.2f clk per inst per SIMD, wall %.3f ms, eff clock %.2f GHz\n", name, ILP,
           blocks_per_cu, avg / ninst, avg / ninst / blocks_per_cu, ms, avg / (ms * 1e6));
    hipFree(d);
}

int main()
{
#define ALL(OP, NAME) run<OP, 1>(NAME, 1); run<OP, 4>(NAME, 1); run<OP, 1>(NAME, 3); run<OP, 4>(NAME, 3);
    ALL(0, "v_fma_f64")
    ALL(1, "v_add_f64")
    ALL(2, "v_mul_f64")
    ALL(3, "v_rcp_f64")
    ALL(4, "v_ldexp_f64(+and)")
    ALL(5, "v_rndne_f64(+mul)")
    ALL(6, "cvt_i32_f64+cvt_f64_i32+add")
    ALL(7, "v_cmp_f64+2cndmask")
    ALL(8, "3x int32 alu")
    ALL(9, "v_max_f64")
    ALL(10, "v_mov_b32")
    return 0;
}
