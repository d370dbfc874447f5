// Kernel lab: times ONE instantiation of the fused RK4 kernel on a synthetic N = 2^20 state, without
// Python / torch in the loop (seconds per experiment).  Used with -DMARL_ABLATE_* to price the pieces of
// the point evaluation (tools/README in DESIGN.md).   hipcc --offload-arch=gfx950 -O3 -std=c++17 ...
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#define MARL_LAB 1
#include "../integrating-diagenetic-equations-using-python_amd/csrc/marl_api.hip"

#ifndef LAB_BLK
#define LAB_BLK 256
#endif
#ifndef LAB_CPT
#define LAB_CPT 1
#endif
#ifndef LAB_NSTEPS
#define LAB_NSTEPS 1
#endif

int main(int argc, char** argv)
{
    const int64_t N = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    marl_params p{};
    // default scenario (tests/golden/params_default.json)
    p.CA0 = 0.6; p.CC0 = 0.3; p.cCa0 = p.cCO30 = 0.4991345125083419; p.Phi0 = 0.8;
    p.sedimentationrate = 0.1; p.Xstar = 1319; p.Tstar = 13190; p.k1 = p.k2 = 1; p.k3 = p.k4 = 0.1;
    p.m1 = p.m2 = 2.48; p.n1 = p.n2 = 2.8; p.b = 0.0005; p.beta = 0.1; p.rhos = p.rhos0 = 2.8630000000000004; p.rhow = 1.023;
    p.KA = 6.45654229034655e-07; p.KC = 4.2657951880159254e-07; p.muA = 100.09; p.D0Ca = 131.9; p.PhiNR = 0.8; p.PhiInfty = 0.01;
    p.PhiIni = 0.8; p.DCa = 131.9; p.DCO3 = 272.6; p.length = 500.0 / 1319; p.shallow_limit = 50.0 / 1319; p.deep_limit = 150.0 / 1319;
    p.FV_switch = 1;
    marl_ctx* ctx = nullptr;
    if (marl_ctx_create(&p, 1, N, 0, &ctx)) { printf("create failed: %s\n", marl_last_error(nullptr)); return 1; }
    if (argc > 3 && atoi(argv[3])) marl_set_option(ctx, "no_reuse", 1);   // every evaluation takes its full path (the sweeps' regime)
    std::vector<double> y(5 * N);
    const double L = p.length, ini[5] = {0.6, 0.3, p.cCa0, p.cCO30, 0.8};
    for (int f = 0; f < 5; f++)
        for (int64_t i = 0; i < N; i++) y[f * N + i] = ini[f] * (1 + 0.01 * sin(2 * M_PI * 8 * ((i + 0.5) * L / N) / L));
    double *d0, *d1;
    hipMalloc(&d0, sizeof(double) * 5 * N);
    hipMalloc(&d1, sizeof(double) * 5 * N);
    hipMemcpy(d0, y.data(), sizeof(double) * 5 * N, hipMemcpyHostToDevice);
    const double dx = L / N, dt = 0.25 * dx * dx;
    constexpr int V = LAB_BLK * LAB_CPT - 8 * LAB_NSTEPS;
    const dim3 grid((unsigned)((N + V - 1) / V));
    auto launch = [&](double* a, double* b) {
        hipLaunchKernelGGL((rk4_fused_kernel<LAB_BLK, LAB_CPT, LAYOUT_FIELD_MAJOR, LAB_NSTEPS>), grid, dim3(LAB_BLK), 0, 0, a, b, ctx->dconsts, ctx->slab, dt);
    };
    for (int i = 0; i < 10; i++) { launch(d0, d1); std::swap(d0, d1); }
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < steps / LAB_NSTEPS; i++) { launch(d0, d1); std::swap(d0, d1); }
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipMemcpy(y.data(), d0, sizeof(double) * 5 * N, hipMemcpyDeviceToHost);
    double chk = 0;
    for (double v : y) chk += v;
#ifdef MARL_LAB_CLOCK
    {
        std::vector<unsigned long long> ck(3 * 16384);
        hipError_t ce = hipMemcpyFromSymbol(ck.data(), HIP_SYMBOL(marl::marl_lab_clock), sizeof(unsigned long long) * ck.size(), 0, hipMemcpyDeviceToHost);
        if (ce != hipSuccess) printf("symbol copy failed: %s\n", hipGetErrorString(ce));
        const int nb = (int)std::min<int64_t>(grid.x, 16384);
        std::vector<double> clk, life;
        unsigned long long rmin = ~0ull, rmax = 0;
        for (int b = 0; b < nb; b++) {
            clk.push_back((double)ck[3 * b] / (double)ck[3 * b + 1] * 0.1);
            life.push_back(ck[3 * b + 1] * 0.01);
            rmin = std::min(rmin, ck[3 * b + 2]); rmax = std::max(rmax, ck[3 * b + 2] + ck[3 * b + 1]);
        }
        std::sort(clk.begin(), clk.end()); std::sort(life.begin(), life.end());
        printf("in-kernel clock median %.3f GHz (min %.3f max %.3f); block lifetime median %.2f us (min %.2f max %.2f); kernel span %.2f us\n",
               clk[nb / 2], clk[0], clk[nb - 1], life[nb / 2], life[0], life[nb - 1], (rmax - rmin) * 0.01);
    }
#endif
#ifdef MARL_LAB_PHASE_CLOCK
    {
        unsigned long long ph[8];
        hipMemcpyFromSymbol(ph, HIP_SYMBOL(marl::marl_lab_phase), sizeof ph, 0, hipMemcpyDeviceToHost);
        const double ev = (double)ph[5];
        const char* name[5] = {"edge writes (5 ds_write)", "own-cell phase (point_local)", "wait at the exchange barrier", "neighbour reads + stencil phase (point_rates)",
                               "between evaluations (RK combination, loads / stores)"};
        double tot = 0;
        for (int k = 0; k < 5; k++) tot += ph[k] / ev;
        printf("shader-clock cycles per RHS evaluation of wave 0 of every workgroup (%.0f evaluations; six s_memtime marks per evaluation included):\n", ev);
        for (int k = 0; k < 5; k++) printf("  %-58s %8.1f cycles  %5.1f %%\n", name[k], ph[k] / ev, 100.0 * ph[k] / ev / tot);
        printf("  %-58s %8.1f cycles\n", "total per evaluation", tot);
    }
#endif
    const int done = (steps / LAB_NSTEPS) * LAB_NSTEPS;
    printf("BLK=%d CPT=%d NSTEPS=%d N=%lld: %.3f us/step, %.3e gp-steps/s (%.1f%% of 1e11), checksum %.15g\n", LAB_BLK, LAB_CPT, LAB_NSTEPS,
           (long long)N, best * 1e3 / done, (double)N * done / (best * 1e-3), (double)N * done / (best * 1e-3) / 1e9, chk);
    return 0;
}
