// Micro-benchmark (inline asm, nothing for the compiler to fold): wall-clock cost per wave-instruction per
// SIMD of selected gfx950 VALU instructions, 3 waves per SIMD, 4 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 32
#define BODY(ASM)                                                        \
    for (int it = 0; it < iters; it++) {                                 \
        _Pragma("unroll") for (int r = 0; r < REP; r++) {                \
            asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(i0), "+v"(i1) : "v"(b), "s"(sb) : "vcc"); \
        }                                                                \
    }

template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double seed, int iters)
{
    double a0 = seed + threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed * 0.999;
    int i0 = threadIdx.x, i1 = threadIdx.x + 7;
    double sb = seed * 0.5;
    if (OP == 0) BODY("v_fma_f64 %0, %0, %6, %7\n v_fma_f64 %1, %1, %6, %7\n v_fma_f64 %2, %2, %6, %7\n v_fma_f64 %3, %3, %6, %7")
    if (OP == 1) BODY("v_cmp_gt_f64 vcc, %0, %6\n v_cmp_gt_f64 vcc, %1, %6\n v_cmp_gt_f64 vcc, %2, %6\n v_cmp_gt_f64 vcc, %3, %6")
    if (OP == 2) BODY("v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %4, vcc")
    if (OP == 3) BODY("v_cmp_gt_f64 vcc, %0, %6\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_gt_f64 vcc, %1, %6\n v_cndmask_b32 %5, %5, %4, vcc")
    if (OP == 4) BODY("v_mov_b32 %4, %5\n v_mov_b32 %5, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %4")
    if (OP == 5) BODY("v_cmp_gt_u32 vcc, %4, %5\n v_cmp_gt_u32 vcc, %5, %4\n v_cmp_gt_u32 vcc, %4, %5\n v_cmp_gt_u32 vcc, %5, %4")
    if (OP == 6) BODY("v_max_f64 %0, %0, %6\n v_max_f64 %1, %1, %6\n v_max_f64 %2, %2, %6\n v_max_f64 %3, %3, %6")
    if (OP == 7) BODY("v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %4\n v_ldexp_f64 %2, %2, %4\n v_ldexp_f64 %3, %3, %4")
    if (OP == 8) BODY("v_add_u32 %4, %4, %5\n v_and_b32 %5, %5, %4\n v_ashrrev_i32 %4, 3, %4\n v_lshlrev_b32 %5, 1, %5")
    if (OP == 9) BODY("v_cmp_class_f64 vcc, %0, %4\n v_cmp_class_f64 vcc, %1, %4\n v_cmp_class_f64 vcc, %2, %4\n v_cmp_class_f64 vcc, %3, %4")
    if (OP == 10) BODY("v_mul_f64 %0, %0, %6\n v_add_f64 %1, %1, %6\n v_mul_f64 %2, %2, %6\n v_add_f64 %3, %3, %6")
    if (OP == 11) BODY("v_fma_f64 %0, %0, %6, %7\n v_mov_b32 %4, %5\n v_fma_f64 %1, %1, %6, %7\n v_mov_b32 %5, %4")
    if (OP == 12) BODY("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3")
    if (OP == 13) BODY("v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_f64_i32 %2, %4\n v_cvt_f64_i32 %3, %5")
    if (OP == 14) BODY("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3")
    if (OP == 15) BODY("v_cmp_gt_f64 s[20:21], %0, %6\n v_cndmask_b32 %4, %4, %5, s[20:21]\n v_cmp_gt_f64 s[22:23], %1, %6\n v_cndmask_b32 %5, %5, %4, s[22:23]")
    if (OP == 16) BODY("v_mov_b64 %0, %1\n v_mov_b64 %1, %0\n v_mov_b64 %2, %3\n v_mov_b64 %3, %2")
    if (OP == 17) BODY("v_fma_f64 %0, %0, %6, 1.0\n v_fma_f64 %1, %1, %6, 1.0\n v_fma_f64 %2, %2, %6, 1.0\n v_fma_f64 %3, %3, %6, 1.0")
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1;
}

template <int OP>
void run(const char* name)
{
    const int iters = 400, nblk = 256 * 3;
    double* d;
    hipMalloc(&d, sizeof(double) * nblk * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<nblk, 256>>>(d, 1.0000001, 10);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int t = 0; t < 3; t++) {
        hipEventRecord(e0);
        k<OP><<<nblk, 256>>>(d, 1.0000001, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double per_simd = 3.0 * iters * REP * 4;  // wave-instructions per SIMD
    printf("%-44s %7.3f ns per wave-instruction per SIMD   (= %5.2f x v_fma_f64 @ 2.135)\n", name, best * 1e6 / per_simd, best * 1e6 / per_simd / 2.135);
    hipFree(d);
}

int main()
{
    run<0>("v_fma_f64");
    run<17>("v_fma_f64 (inline const)");
    run<10>("v_mul_f64 / v_add_f64");
    run<6>("v_max_f64");
    run<1>("v_cmp_gt_f64 -> vcc");
    run<9>("v_cmp_class_f64 -> vcc");
    run<5>("v_cmp_gt_u32 -> vcc");
    run<2>("v_cndmask_b32 (vcc)");
    run<3>("v_cmp_gt_f64 + v_cndmask_b32 (vcc) pairs");
    run<15>("v_cmp_gt_f64 + v_cndmask_b32 (sgpr) pairs");
    run<4>("v_mov_b32");
    run<16>("v_mov_b64");
    run<11>("v_fma_f64 + v_mov_b32 alternating");
    run<8>("int32 alu (add/and/ashr/lshl)");
    run<7>("v_ldexp_f64");
    run<12>("v_rndne_f64");
    run<13>("v_cvt_i32_f64 / v_cvt_f64_i32");
    run<14>("v_rcp_f64");
    return 0;
}
