// Lab probe: what does one level of the one-workgroup PCR solve chain (pcr_solve_all, marl_radau.h) cost, and which part of it?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I integrating-diagenetic-equations-using-python_amd/csrc -o tools/lab_bin/solve_chain_probe tools/lab_src/solve_chain_probe.hip
// Variants (one workgroup of 1024 threads, N = 200 cells, 8 levels + the final D^-1 b, REPS solves back to back in one launch):
//   0 the shipped chain           1 no global loads (factor rows = constants)      2 no barrier between levels (wrong results, timing only)
//   3 no LDS traffic (b values = constants; loads + FMAs + barrier)                4 barrier only
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "marl_radau.h"
using namespace marl;
using namespace marl::radau;

template <int VARIANT, class T>
__global__ void __launch_bounds__(PCR_FUSED_THREADS) probe(int64_t N, int nlevels, PcrSystem<T> S, T* x, int reps)
{
    __shared__ T lds[2 * PCR_FUSED_MAX];
    const int n = (int)(NF * N);
    const int k = threadIdx.x;
    for (int kk = k; kk < n; kk += PCR_FUSED_THREADS) lds[kk] = x[kk];
    __syncthreads();
    int cur = 0;
    for (int rep = 0; rep < reps; rep++) {
        for (int level = 0; level <= nlevels; level++) {
            const T* b = lds + cur * PCR_FUSED_MAX;
            T* o = lds + (cur ^ 1) * PCR_FUSED_MAX;
            if (k < n) {
                if constexpr (VARIANT == 0 || VARIANT == 2) pcr_solve_row<T>(N, k, level, nlevels, S, b, o);
                else if constexpr (VARIANT == 1) {   // LDS + FMAs + barrier, no global loads
                    const int64_t i = k / NF, s = (int64_t)1 << (level < nlevels ? level : 0);
                    T acc = b[k];
                    const T c = lift(1e-3, acc);
                    if (i - s >= 0) for (int q = 0; q < NF; q++) acc = madd(acc, c, b[(i - s) * NF + q]);
                    if (i + s < N) for (int q = 0; q < NF; q++) acc = madd(acc, c, b[(i + s) * NF + q]);
                    o[k] = acc;
                } else if constexpr (VARIANT == 3) {  // global loads + FMAs + barrier, one LDS read / write
                    const int64_t i = k / NF;
                    const int r = k % NF, lv = level < nlevels ? level : 0;
                    const T* al = S.alpha + ((int64_t)lv * N + i) * 25 + r * NF;
                    const T* ga = S.gamma + ((int64_t)lv * N + i) * 25 + r * NF;
                    T acc = b[k];
                    const T c = lift(1e-3, acc);
                    for (int q = 0; q < NF; q++) acc = madd(acc, al[q], c);
                    for (int q = 0; q < NF; q++) acc = madd(acc, ga[q], c);
                    o[k] = acc;
                }
            }
            if constexpr (VARIANT != 2) __syncthreads();
            cur ^= 1;
        }
    }
    for (int kk = k; kk < n; kk += PCR_FUSED_THREADS) x[kk] = lds[cur * PCR_FUSED_MAX + kk];
}

static double mk(double a, double) { return a; }
static cplx mk(double a, cplx) { return cplx{a, 0.5 * a}; }

template <class T>
static void run(const char* name)
{
    const int64_t N = 200;
    const int nlevels = 8, reps = 2000;
    PcrSystem<T> S{};
    std::vector<T> h((size_t)nlevels * 25 * N);
    for (size_t j = 0; j < h.size(); j++) h[j] = mk(1e-4 * (double)((double)((j * 7919) % 1000) - 500.0), T{});
    hipMalloc((void**)&S.alpha, h.size() * sizeof(T));
    hipMalloc((void**)&S.gamma, h.size() * sizeof(T));
    hipMemcpy(S.alpha, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    hipMemcpy(S.gamma, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    for (int q = 0; q < 2; q++) { hipMalloc((void**)&S.Dinv[q], 25 * N * sizeof(T)); hipMemcpy(S.Dinv[q], h.data(), 25 * N * sizeof(T), hipMemcpyHostToDevice); }
    T* x;
    hipMalloc((void**)&x, NF * N * sizeof(T));
    hipMemset(x, 0, NF * N * sizeof(T));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto kern, int variant) {
        float best = 1e30f;
        for (int t = 0; t < 3; t++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(1), dim3(PCR_FUSED_THREADS), 0, 0, N, nlevels, S, x, reps);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("%s variant %d: %.3f us per solve, %.3f us per level\n", name, variant, best * 1e3 / reps, best * 1e3 / reps / (nlevels + 1));
    };
    time(probe<0, T>, 0); time(probe<1, T>, 1); time(probe<2, T>, 2); time(probe<3, T>, 3); time(probe<4, T>, 4);
}

int main()
{
    run<double>("real   ");
    run<cplx>("complex");
    return 0;
}
