// marl_radau_wg.h - a sweep of implicit Radau integrations with ONE WORKGROUP PER INSTANCE: the whole integration of an instance
// (scipy's step logic, finite-difference Jacobian, block cyclic reduction, Newton iterations, error estimate, monitors) runs inside
// one persistent 1024-thread workgroup, instance after instance from an atomic queue - no launch, no host round trip and no other
// workgroup anywhere on an instance's path.
//
// Why (round 3): the launch-per-action sweep of marl_radau_batch.h is bound by its slowest instance - every Newton iteration of that
// instance is one host-driven cycle of ~8 launches (150 - 200 us), 4000 cycles for the slowest of 512 scenarios: 0.7 s of the sweep's
// 0.87 s.  Instances are independent, so the natural unit of parallelism is the instance (the shape of rk45_sweep_kernel).
//
// STATUS (round 3, profiles/r03_lab_radau_wg.log).  Two ways to use it (option radau_sweep_wg):
//   1  HYBRID - the default for sweeps of small grids: the workgroup runs everything of an instance that is sequential and small (step
//      logic, single right-hand sides, Newton iterations, error estimates, accepted steps, event root finding) and hands the instance
//      back to the host cycle when the step logic asks for a Jacobian or a factorisation - the two pieces that want the whole chip, done
//      by the launch kernels over work lists as before.  One host cycle per Jacobian / factorisation instead of one per action.
//      64 / 512 / 4096 scenarios: 0.42 / 0.72 / 1.73 s (launch per action: 0.55 / 0.88 / 1.98 s); bit-identical statistics.
//   2  everything in the workgroup, no host in the loop: 0.98 / 1.15 / 2.7 s - cyclic reduction does log2 N times the work of a block
//      Thomas sweep, and on ONE compute unit its 126 group calls per factorisation are dependent L2 round trips: 0.64 ms per
//      factorisation, 24 of the 35 ms of a Scenario-A instance (Newton iterations 6.1 ms = 118 x 52 us, the rest 3.6 ms; the launch path
//      needs 9.4 ms for the same instance alone on the chip).  What this mode needs to pay: a wave-cooperative block THOMAS
//      factorisation / solve inside the workgroup (estimate: 12 ms per instance, 0.24 / 0.41 s for 512 / 4096 scenarios).  Not built.
//
// Arithmetic: every piece below restates the body of the corresponding batched launch kernel (marl_radau_batch.h), the step logic IS the
// same function (radau_control_step), and the contraction-ambiguous expressions go through the same spelled-out helpers (dot3,
// cmul_ref) in both: the hybrid path reproduces the launch path's statistics exactly and its states to 1e-12
// (tests/test_gpu_radau.py::test_radau_sweep_workgroup_paths_against_the_launch_path).  Mode 2 inlines the factorisation into the big
// kernel, where the compiler may contract its complex multiply-adds differently: compared like two correct runs.
// Grids of up to PCR_FUSED_MAX / 5 = 409 cells (the solve keeps a right-hand side in LDS).
#pragma once
#include "marl_radau_batch.h"

namespace marl {
namespace radau {

constexpr int WG_THREADS = 1024;
constexpr int WG_GROUPS = WG_THREADS / 32;   // cells factorised per pass (25 lanes of a 32-lane group per cell)

// device pointers of INSTANCE 0's work arena (marl_api.hip radau_alloc); instance b lives `zs` bytes further
struct WgWork {
    double *y, *f, *fnew, *ynew, *err, *yerr, *yold, *scale, *tmp, *Z, *W, *F, *Q, *YS;
    double *fac, *h, *yscale, *maxdiff, *scl, *hnew, *Jraw, *YP, *FN, *J, *rhs_r;
    cplx* rhs_c;
    int32_t* small;
    const int32_t* groups;   // shared by all instances
    PcrSystem<double> Sr;
    PcrSystem<cplx> Sc;
    int ng, nlevels;
    int64_t zs;
    // cyclic reduction in front of PCR (hybrid modes: the launch kernels factorise, the solves here follow; plan.k == 0: PCR alone)
    CrPlan plan;
    CrSystem<double> Cr;
    CrSystem<cplx> Cc;
};

template <class P>
__device__ __forceinline__ P* wg_at(P* p, int64_t off) { return reinterpret_cast<P*>(reinterpret_cast<char*>(p) + off); }
template <class T>
__device__ __forceinline__ PcrSystem<T> wg_at(PcrSystem<T> S, int64_t off)
{
#pragma unroll
    for (int k = 0; k < 2; k++) { S.L[k] = wg_at(S.L[k], off); S.D[k] = wg_at(S.D[k], off); S.U[k] = wg_at(S.U[k], off); S.Dinv[k] = wg_at(S.Dinv[k], off); S.b[k] = wg_at(S.b[k], off); }
    S.alpha = wg_at(S.alpha, off); S.gamma = wg_at(S.gamma, off);
    return S;
}

template <class T>
__device__ __forceinline__ CrSystem<T> wg_at(CrSystem<T> C, int64_t off)
{
    C.L = wg_at(C.L, off); C.D = wg_at(C.D, off); C.U = wg_at(C.U, off); C.Dinv = wg_at(C.Dinv, off); C.P = wg_at(C.P, off); C.Q = wg_at(C.Q, off);
    C.alpha = wg_at(C.alpha, off); C.gamma = wg_at(C.gamma, off); C.b = wg_at(C.b, off);
    return C;
}

// LDS of the workgroup: one buffer reused by the phases (factorisation stage / solve right-hand sides / reductions), the log / exp
// tables, the instance's controller and small words
union WgBuf {
    PcrStage<cplx> stage[WG_GROUPS];          // 38 KB
    cplx solve[2 * PCR_FUSED_MAX];            // 64 KB (the real system uses half)
    double red[NQ * WG_THREADS];              // 64 KB (block_reduce of the monitors; red[0 .. 1024) for the sums of squares)
};

// dst[s][.] = f(src[s][.]) for ns states of the instance's model, field-major (rhs_kernel's body)
template <bool VD>
__device__ __forceinline__ void wg_rhs(const double* __restrict__ src, double* __restrict__ dst, int ns, int64_t N, const DevConsts& C, const HotConsts& K,
                                       const Tables& T)
{
    const int64_t n = NF * N;
    for (int64_t idx = threadIdx.x; idx < ns * N; idx += WG_THREADS) {
        const int64_t s = idx / N, l = idx % N;
        const double* y = src + s * n;
        double uc[NF], um[NF], up[NF], r[NF];
        PointAux aux;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            uc[f] = y[f * N + l];
            um[f] = (l > 0) ? y[f * N + l - 1] : ghost_lower(C.bc[f], uc[f]);
        }
#pragma unroll
        for (int f = 0; f < NF; f++) up[f] = (l < N - 1) ? y[f * N + l + 1] : ghost_upper(f, uc[f], um[f]);
        PointCache<0> pc;
        bool live = false;
        rhs_point<TR_PLAIN, 0, VD>(uc, um, up, l >= C.mask_lo && l < C.mask_hi, K, &C, T, r, aux, pc, live);
#pragma unroll
        for (int f = 0; f < NF; f++) dst[s * n + f * N + l] = r[f];
    }
}

// the seven monitors of y -> g (every thread), through block_reduce (monitors_kernel + reduce_records_kernel: extrema only - exact)
__device__ __forceinline__ void wg_monitors(const double* __restrict__ y, int64_t N, const DevConsts& C, const Tables& T, double* red, double* g_out)
{
    double q[NQ];
    monitors_init(q);
    for (int64_t l = threadIdx.x; l < N; l += WG_THREADS) {
        double u[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) u[f] = y[f * N + l];
        const double Phi = u[4];
        const double F = 1.0 - fast_exp(10.0 - 10.0 * rcp_nr(Phi), T);
        const double rF = C.hot.rhorat * F;
        monitors_accumulate(q, u, C.hot.presum + rF * (Phi * Phi * Phi) * rcp_nr(1.0 - Phi), C.hot.presum - rF * Phi * Phi);
    }
    block_reduce<WG_THREADS, NQ, NQMIN>(q, red);   // (ends with a barrier)
    if (threadIdx.x == 0) {
        const double g[7] = {q[1], q[2], q[3], q[5] - 1.0, q[6] - 1.0, q[4], q[7]};   // record_to_events (marl_api.hip)
        for (int e = 0; e < 7; e++) g_out[e] = g[e];
    }
    __syncthreads();
}

// sum over the workgroup in the tree order of newton_update_batch_kernel / error_norm_batch_kernel
__device__ __forceinline__ double wg_sum(double ss, double* red)
{
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// hybrid != 0 (round 3, the default for sweeps of small grids): the workgroup runs everything of an instance that is SEQUENTIAL and
// small - the step logic, single right-hand sides, Newton iterations (stage derivatives + both cyclic-reduction solves + update + norm),
// error estimates, accepted-step bookkeeping, event root finding - and hands the instance back to the host cycle when the step logic asks
// for a Jacobian or a factorisation: those are the two pieces that want the whole chip (finite-difference columns of 15 - 21 perturbed
// states; 2 x 9 cyclic-reduction levels over all cells), and the launch kernels of marl_radau.h do them over work lists as before.  An
// instance's host cycles drop from one per ACTION (~4000 for the slowest of 512 scenarios) to one per Jacobian / factorisation.
// counts / lists: the work lists of marl_radau_batch.h (L_JAC, L_LU, L_RUNNING); g_dense / t_events: event root finding (may be NULL).
// HYBRID is a template parameter (round 4): the instantiation the sweeps use by default (2) then carries neither the in-workgroup
// factorisation (a 5 x 5 Gauss-Jordan per lane group: the register-hungriest piece) nor its share of the kernel's spills.
template <bool VD, int HYBRID = 0>
__global__ void __launch_bounds__(WG_THREADS) radau_wg_kernel(RadauCtl* __restrict__ ctls, int64_t B, int64_t N, WgWork w, const DevConsts* __restrict__ consts,
                                                              double fd_threshold, P33 P, double E0, double E1, double E2, unsigned* __restrict__ next_instance,
                                                              int32_t* __restrict__ counts = nullptr, int32_t* __restrict__ lists = nullptr,
                                                              double* __restrict__ t_events = nullptr, const int32_t* __restrict__ run_list = nullptr,
                                                              int64_t todo = -1)
{
    constexpr int hybrid = HYBRID;
    __shared__ WgBuf buf;
    __shared__ double tabs[TABLE_DOUBLES];
    __shared__ RadauCtl sc;
    __shared__ double g_now[7], g_dense[7];
    __shared__ unsigned s_b;
    __shared__ int s_nonfinite;
    const Tables T = load_tables(tabs, WG_THREADS);
    const int64_t n = NF * N;
    const int tid = threadIdx.x;

    while (true) {
        __syncthreads();
        if (tid == 0) s_b = atomicAdd(next_instance, 1u);
        __syncthreads();
        if ((int64_t)s_b >= (todo < 0 ? B : todo)) break;
        const int64_t b = run_list ? (int64_t)run_list[s_b] : (int64_t)s_b;   // (this pass's instances: all of them, or the ones handed back last time)
        const int64_t off = b * w.zs;
        const DevConsts& C = consts[b];
        const HotConsts K = load_hot(&C);
        if (tid == 0) { sc = ctls[b]; s_nonfinite = 0; }
        __syncthreads();
        if (sc.pc == PC_DONE) continue;
        double* y = wg_at(w.y, off);
        wg_monitors(y, N, C, T, buf.red, g_now);   // monitors of y(t0) (the launch path's cycle 0)

#ifdef MARL_WG_CLOCK   // kernel-lab build: where an instance's time goes (100 MHz ticks per kind of work, printed for instance 0)
        unsigned long long ck[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck0 = __builtin_amdgcn_s_memrealtime();
        unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WG_TICK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); ck[k] += t_ - ck0; cnt[k]++; ck0 = t_; } while (0)
#else
#define WG_TICK(k) do { } while (0)
#endif
        while (true) {
            if (tid == 0) {
                double g[7], gd[7];
                for (int e = 0; e < 7; e++) { g[e] = g_now[e]; gd[e] = g_dense[e]; }
                radau_control_step(sc, g, n, gd, t_events ? t_events + b * 7 * sc.max_events : nullptr);
            }
            __syncthreads();
            WG_TICK(0);
            const int32_t action = sc.action;
            if (sc.pc == PC_DONE && action == 0) break;
            // back to the host cycle: the launch kernels do this over the work lists (hybrid == 2: factorisations only - the Jacobian stays here)
            if (hybrid && ((action & A_LU) || (hybrid == 1 && (action & A_JAC)))) {
                if (tid == 0) {
                    atomicAdd(&counts[L_RUNNING], 1);
                    const int which = (action & A_JAC) ? L_JAC : L_LU;
                    lists[(int64_t)which * B + atomicAdd(&counts[which], 1)] = (int32_t)b;
                }
                break;
            }
            // ---- event root finding: the accepted step's dense output at the abscissa Brent asks for, and that state's monitors
            if (action & A_DENSE) {
                const double *Q = wg_at(w.Q, off), *yold = wg_at(w.yold, off);
                double* out = wg_at(w.tmp, off);
                const double p1 = sc.dense_x, p2 = p1 * p1, p3 = p2 * p1;
                for (int64_t i = tid; i < n; i += WG_THREADS) out[i] = dot3(Q[3 * i], p1, Q[3 * i + 1], p2, Q[3 * i + 2], p3) + yold[i];
                __syncthreads();
                wg_monitors(out, N, C, T, buf.red, g_dense);
            }

            // ---- a single state -> its derivative: y -> f (start), y + err -> tmp (second error estimate), y_new -> f_new (accepted step)
            if (action & (A_RHS_Y | A_ERR2 | A_ACCEPT)) {
                const double* src = (action & A_RHS_Y) ? y : ((action & A_ERR2) ? wg_at(w.yerr, off) : wg_at(w.ynew, off));
                double* dst = (action & A_RHS_Y) ? wg_at(w.f, off) : ((action & A_ERR2) ? wg_at(w.tmp, off) : wg_at(w.fnew, off));
                wg_rhs<VD>(src, dst, 1, N, C, K, T);
                __syncthreads();
                WG_TICK(1);
            }
            // ---- accepted step: Q = Z^T P, y_old <- y, y <- y_new, f <- f_new (accept_kernel), then the monitors of the new y
            if (action & A_ACCEPT) {
                const double* Z = wg_at(w.Z, off);
                double *Q = wg_at(w.Q, off), *yold = wg_at(w.yold, off), *f = wg_at(w.f, off);
                const double *ynew = wg_at(w.ynew, off), *fnew = wg_at(w.fnew, off);
                for (int64_t i = tid; i < n; i += WG_THREADS) {
                    const double z0 = Z[i], z1 = Z[n + i], z2 = Z[2 * n + i];
#pragma unroll
                    for (int m = 0; m < 3; m++) Q[3 * i + m] = dot3(z0, P.p[0][m], z1, P.p[1][m], z2, P.p[2][m]);
                    yold[i] = y[i];
                    y[i] = ynew[i];
                    f[i] = fnew[i];
                }
                __syncthreads();
                wg_monitors(y, N, C, T, buf.red, g_now);
                WG_TICK(2);
            }
            // ---- finite-difference Jacobian at (y, f): num_jac with its second trial step (fd_prepare / fd_columns / fd_finish kernels)
            if (HYBRID != 1 && (action & A_JAC)) {
                const double* f0 = wg_at(w.f, off);
                double *fac = wg_at(w.fac, off), *h = wg_at(w.h, off), *yscale = wg_at(w.yscale, off), *YP = wg_at(w.YP, off), *FN = wg_at(w.FN, off);
                double *Jraw = wg_at(w.Jraw, off), *maxdiff = wg_at(w.maxdiff, off), *scl = wg_at(w.scl, off), *hnew = wg_at(w.hnew, off), *J = wg_at(w.J, off);
                int32_t* small = wg_at(w.small, off);
                const int first = !sc.have_factor;
                const int ng = w.ng;
                for (int64_t j = tid; j < n; j += WG_THREADS) {
                    double fc = first ? sqrt(EPS) : fac[j];
                    const double yj = y[j];
                    const double f_sign = (f0[j] >= 0) ? 1.0 : -1.0;
                    const double ay = fabs(yj);
                    const double ys = f_sign * (fd_threshold > ay ? fd_threshold : ay);
                    double hj = (yj + fc * ys) - yj;
                    while (hj == 0) { fc *= 10; hj = (yj + fc * ys) - yj; }
                    fac[j] = fc; h[j] = hj; yscale[j] = ys;
                    const int gj = w.groups[j];
                    for (int g = 0; g < ng; g++) YP[(int64_t)g * n + j] = (g == gj) ? yj + hj : yj;
                }
                __syncthreads();
                wg_rhs<VD>(YP, FN, ng, N, C, K, T);
                __syncthreads();
                for (int64_t j = tid; j < n; j += WG_THREADS) {
                    const int fp = (int)(j / N);
                    const int64_t ip = j % N;
                    const int gj = w.groups[j];
                    const double* fg = FN + (int64_t)gj * n;
                    double dcol[15];
                    int64_t arg;
                    const double md = fd_column(f0, fg, fp, ip, N, dcol, arg);
                    const double a = fabs(f0[arg]), bb = fabs(fg[arg]);
                    const double scv = a > bb ? a : bb;
#pragma unroll
                    for (int e = 0; e < 15; e++) Jraw[j * 15 + e] = dcol[e];
                    maxdiff[j] = md; scl[j] = scv;
                    const bool sm = md < pow(EPS, 0.875) * scv;
                    small[j] = sm ? 1 : 0;
                    const double yj = y[j];
                    const double hn = sm ? (yj + (10 * fac[j]) * yscale[j]) - yj : 0.0;
                    hnew[j] = hn;
                    for (int g = 0; g < ng; g++) YP[(int64_t)g * n + j] = (g == gj) ? yj + hn : yj;
                }
                __syncthreads();
                wg_rhs<VD>(YP, FN, ng, N, C, K, T);
                __syncthreads();
                for (int64_t j = tid; j < n; j += WG_THREADS) {
                    const int fp = (int)(j / N);
                    const int64_t ip = j % N;
                    double dcol[15];
#pragma unroll
                    for (int e = 0; e < 15; e++) dcol[e] = Jraw[j * 15 + e];
                    double md = maxdiff[j], scv = scl[j], fc = fac[j], hj = h[j];
                    if (small[j]) {
                        const double* fg = FN + (int64_t)w.groups[j] * n;
                        double dnew[15];
                        int64_t arg;
                        const double md_new = fd_column(f0, fg, fp, ip, N, dnew, arg);
                        const double a = fabs(f0[arg]), bb = fabs(fg[arg]);
                        const double sc_new = a > bb ? a : bb;
                        if (md * sc_new < md_new * scv) {
                            fc = 10 * fc; hj = hnew[j]; scv = sc_new; md = md_new;
#pragma unroll
                            for (int e = 0; e < 15; e++) dcol[e] = dnew[e];
                        }
                    }
#pragma unroll
                    for (int f = 0; f < NF; f++)
#pragma unroll
                        for (int di = 0; di < 3; di++) {
                            const int64_t i = ip + di - 1;
                            if (i >= 0 && i < N) J[((i * 3 + (2 - di)) * NF + f) * NF + fp] = dcol[f * 3 + di] / hj;
                        }
                    if (md < pow(EPS, 0.75) * scv) fc *= 10;
                    if (md > pow(EPS, 0.25) * scv) fc *= 0.1;
                    if (fc < 1e3 * EPS) fc = 1e3 * EPS;
                    fac[j] = fc; h[j] = hj;
                }
                __syncthreads();
                WG_TICK(3);
            }
            // ---- factorise mu_r I - J and mu_c I - J: block cyclic reduction, every level here (pcr_factor_kernel per level and system)
            if (HYBRID == 0 && (action & A_LU)) {
                const double* J = wg_at(w.J, off);
                const PcrSystem<double> Sr = wg_at(w.Sr, off);
                const PcrSystem<cplx> Sc = wg_at(w.Sc, off);
                const double mu_r = sc.mu_r;
                const cplx mu_c = {sc.mu_c_re, sc.mu_c_im};
                const int g = tid >> 5, e = tid & 31;
                const int64_t passes = (N + WG_GROUPS - 1) / WG_GROUPS;
                for (int level = -1; level < w.nlevels; level++) {
                    for (int64_t p = 0; p < passes; p++)
                        pcr_factor_group<double>(J, N, level, mu_r, Sr, *reinterpret_cast<PcrStage<double>*>(&buf.stage[g]), 1.0, p * WG_GROUPS + g, e);
                    for (int64_t p = 0; p < passes; p++) pcr_factor_group<cplx>(J, N, level, mu_c, Sc, buf.stage[g], 1.0, p * WG_GROUPS + g, e);
                    __syncthreads();   // a level reads what the level before wrote (global memory, this workgroup only)
                }
                WG_TICK(4);
            }
            // ---- one Newton iteration of the collocation system (newton_begin / rhs / newton_rhs / solve / newton_update kernels)
            if (action & A_NEWTON) {
                double *scale = wg_at(w.scale, off), *Z = wg_at(w.Z, off), *W = wg_at(w.W, off), *YS = wg_at(w.YS, off), *F = wg_at(w.F, off);
                double* rhs_r = wg_at(w.rhs_r, off);
                cplx* rhs_c = wg_at(w.rhs_c, off);
                if (sc.newton_begin) {
                    const double *Q = wg_at(w.Q, off), *yold = wg_at(w.yold, off);
                    for (int64_t i = tid; i < n; i += WG_THREADS) {
                        const double yi = y[i];
                        double z[3] = {0.0, 0.0, 0.0};
                        if (sc.have_sol) {
                            const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2], yo = yold[i];
#pragma unroll
                            for (int s = 0; s < 3; s++) {
                                const double p1 = sc.x3[s], p2 = p1 * p1, p3 = p2 * p1;
                                z[s] = (dot3(q0, p1, q1, p2, q2, p3) + yo) - yi;
                            }
                        }
                        scale[i] = sc.atol + fabs(yi) * sc.rtol;
                        Z[i] = z[0]; Z[n + i] = z[1]; Z[2 * n + i] = z[2];
                        W[i] = dot3(TI00, z[0], TI01, z[1], TI02, z[2]);
                        W[n + i] = dot3(TI10, z[0], TI11, z[1], TI12, z[2]);
                        W[2 * n + i] = dot3(TI20, z[0], TI21, z[1], TI22, z[2]);
                        YS[i] = yi + z[0]; YS[n + i] = yi + z[1]; YS[2 * n + i] = yi + z[2];
                    }
                    __syncthreads();
                }
                wg_rhs<VD>(YS, F, 3, N, C, K, T);
                __syncthreads();
                {
                    const double mu_r = sc.mu_r;
                    const cplx mu_c = {sc.mu_c_re, sc.mu_c_im};
                    int bad = 0;
                    for (int64_t kk = tid; kk < n; kk += WG_THREADS) {
                        const int64_t i = to_field_major(kk, N);
                        const double f0 = F[i], f1 = F[n + i], f2 = F[2 * n + i];
                        bad |= !(isfinite(f0) && isfinite(f1) && isfinite(f2));
                        rhs_r[kk] = __builtin_fma(-mu_r, W[i], dot3(f0, TI00, f1, TI01, f2, TI02));
                        const cplx wv = {W[n + i], W[2 * n + i]};
                        const cplx fc = {dot3(f0, TI10, f1, TI11, f2, TI12), dot3(f0, TI20, f1, TI21, f2, TI22)};
                        rhs_c[kk] = fc - cmul_ref(mu_c, wv);
                    }
                    if (bad) s_nonfinite = 1;
                }
                __syncthreads();
                crpcr_solve_all<double>(w.plan, N, w.nlevels, wg_at(w.Cr, off), wg_at(w.Sr, off), rhs_r, rhs_r, reinterpret_cast<double*>(buf.solve));
                __syncthreads();
                crpcr_solve_all<cplx>(w.plan, N, w.nlevels, wg_at(w.Cc, off), wg_at(w.Sc, off), rhs_c, rhs_c, buf.solve);
                __syncthreads();
                double ss = 0;
                for (int64_t kk = tid; kk < n; kk += WG_THREADS) {
                    const int64_t i = to_field_major(kk, N);
                    const double d0 = rhs_r[kk], d1 = rhs_c[kk].re, d2 = rhs_c[kk].im;
                    const double s = scale[i];
                    const double e0 = d0 / s, e1 = d1 / s, e2 = d2 / s;
                    ss += dot3(e0, e0, e1, e1, e2, e2);
                    const double w0 = W[i] + d0, w1 = W[n + i] + d1, w2 = W[2 * n + i] + d2;
                    W[i] = w0; W[n + i] = w1; W[2 * n + i] = w2;
                    const double z0 = dot3(T00, w0, T01, w1, T02, w2), z1 = dot3(T10, w0, T11, w1, T12, w2), z2 = dot3(T20, w0, T21, w1, T22, w2);
                    Z[i] = z0; Z[n + i] = z1; Z[2 * n + i] = z2;
                    const double yi = y[i];
                    YS[i] = yi + z0; YS[n + i] = yi + z1; YS[2 * n + i] = yi + z2;
                }
                const double total = wg_sum(ss, buf.red);
                if (tid == 0) { sc.sumsq = total; sc.nonfinite = s_nonfinite; s_nonfinite = 0; }
                WG_TICK(5);
            }
            // ---- error estimate: real system with f (or f(y + err) in tmp) + Z^T E / h (error_rhs / solve / error_norm kernels)
            if (action & (A_ERR | A_ERR2)) {
                const double* fvec = (action & A_ERR2) ? wg_at(w.tmp, off) : wg_at(w.f, off);
                const double* Z = wg_at(w.Z, off);
                double *rhs_r = wg_at(w.rhs_r, off), *ynew = wg_at(w.ynew, off), *err = wg_at(w.err, off), *yerr = wg_at(w.yerr, off);
                const double hstep = sc.h;
                for (int64_t kk = tid; kk < n; kk += WG_THREADS) {
                    const int64_t i = to_field_major(kk, N);
                    const double ZE = dot3(Z[i], E0, Z[n + i], E1, Z[2 * n + i], E2) / hstep;
                    rhs_r[kk] = fvec[i] + ZE;
                    ynew[i] = y[i] + Z[2 * n + i];
                }
                __syncthreads();
                crpcr_solve_all<double>(w.plan, N, w.nlevels, wg_at(w.Cr, off), wg_at(w.Sr, off), rhs_r, rhs_r, reinterpret_cast<double*>(buf.solve));
                __syncthreads();
                double ss = 0;
                for (int64_t kk = tid; kk < n; kk += WG_THREADS) {
                    const int64_t i = to_field_major(kk, N);
                    const double e = rhs_r[kk];
                    const double a = fabs(y[i]), bb = fabs(ynew[i]);
                    const double s = sc.atol + ((a > bb || a != a) ? a : bb) * sc.rtol;
                    const double q = e / s;
                    ss = __builtin_fma(q, q, ss);
                    err[i] = e;
                    yerr[i] = y[i] + e;
                }
                const double total = wg_sum(ss, buf.red);
                if (tid == 0) sc.sumsq = total;
                WG_TICK(6);
            }
            __syncthreads();
        }
#ifdef MARL_WG_CLOCK
        if (tid == 0 && b == 0)
            printf("WGCLOCK instance 0 [x10 ns / count]: control %llu/%llu rhs1 %llu/%llu accept+monitors %llu/%llu jac %llu/%llu lu %llu/%llu newton %llu/%llu err %llu/%llu\n",
                   ck[0], cnt[0], ck[1], cnt[1], ck[2], cnt[2], ck[3], cnt[3], ck[4], cnt[4], ck[5], cnt[5], ck[6], cnt[6]);
#endif
        if (tid == 0) ctls[b] = sc;
    }
}

// ---- single runs on small grids: a Newton iteration in TWO launches (option radau_fused_solve = 3) ---------------------------
// The host-driven iteration of solve_collocation_system was four launches - stage derivatives (rhs_kernel x 3), right-hand
// sides (newton_rhs_kernel), both solves side by side in two workgroups (pcr_solve_fused_kernel), update + norm
// (newton_update_kernel) - and a wait.  Here the two solve workgroups assemble their own right-hand side straight into LDS
// (newton_solve2_kernel), and the update workgroup goes on to evaluate the stage derivatives of the NEXT iteration
// (newton_update_rhs_kernel: the norm is published first, so the host's convergence tests overlap that evaluation; when the
// iteration was the last one the three evaluations are wasted - 600 cells).  Same per-element operations and reduction tree as the
// kernels they replace: bit-identical (tests/test_gpu_radau.py).  Keeping real and complex chain in two workgroups matters: one
// compute unit's memory path bounds a chain (solve_chain_probe), which is why everything-in-one-workgroup (radau_fused_solve = 2) lost.
__global__ void __launch_bounds__(PCR_FUSED_THREADS) newton_solve2_kernel(const double* __restrict__ F, const double* __restrict__ W, int64_t N, double M_real, cplx M_c,
                                                                          int nlevels, PcrSystem<double> Sr, PcrSystem<cplx> Sc, double* __restrict__ rhs_r,
                                                                          cplx* __restrict__ rhs_c, int32_t* __restrict__ flags, CrPlan pl, CrSystem<double> Cr,
                                                                          CrSystem<cplx> Cc)
{
    __shared__ cplx lds[2 * PCR_FUSED_MAX];   // (the real system uses half of the bytes)
    const int64_t n = NF * N;
    if (blockIdx.y == 0) {
        double* l = reinterpret_cast<double*>(lds);
        for (int64_t kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // newton_rhs_kernel, real part
            const int64_t i = to_field_major(kk, N);
            const double f0 = F[i], f1 = F[n + i], f2 = F[2 * n + i];
            if (!(isfinite(f0) && isfinite(f1) && isfinite(f2))) *flags = 1;
            l[kk] = ((f0 * TI00 + f1 * TI01) + f2 * TI02) - M_real * W[i];
        }
        __syncthreads();
        if (pl.k == 0) pcr_solve_all<double>(N, nlevels, Sr, l, rhs_r, l);   // (staged onto itself)
        else crpcr_solve_all<double>(pl, N, nlevels, Cr, Sr, nullptr, rhs_r, l, true);
    } else {
        for (int64_t kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // newton_rhs_kernel, complex part
            const int64_t i = to_field_major(kk, N);
            const double f0 = F[i], f1 = F[n + i], f2 = F[2 * n + i];
            const cplx w = {W[n + i], W[2 * n + i]};
            const cplx fc = {(f0 * TI10 + f1 * TI11) + f2 * TI12, (f0 * TI20 + f1 * TI21) + f2 * TI22};
            lds[kk] = fc - M_c * w;
        }
        __syncthreads();
        if (pl.k == 0) pcr_solve_all<cplx>(N, nlevels, Sc, lds, rhs_c, lds);
        else crpcr_solve_all<cplx>(pl, N, nlevels, Cc, Sc, nullptr, rhs_c, lds, true);
    }
}

template <bool VD>
__global__ void __launch_bounds__(WG_THREADS) newton_update_rhs_kernel(const double* __restrict__ y, const double* __restrict__ rhs_r, const cplx* __restrict__ rhs_c,
                                                                       const double* __restrict__ scale, int64_t N, double* __restrict__ W, double* __restrict__ Z,
                                                                       double* __restrict__ YS, double* __restrict__ F, const DevConsts* __restrict__ consts,
                                                                       double* __restrict__ out)
{
    __shared__ double red[WG_THREADS];
    __shared__ double tabs[TABLE_DOUBLES];
    const Tables T = load_tables(tabs, WG_THREADS);
    const DevConsts& C = consts[0];
    const HotConsts K = load_hot(&C);
    const int64_t n = NF * N;
    double ss = 0;
    for (int64_t kk = threadIdx.x; kk < n; kk += WG_THREADS) {   // newton_update_kernel, one workgroup
        const int64_t i = to_field_major(kk, N);
        const double d0 = rhs_r[kk], d1 = rhs_c[kk].re, d2 = rhs_c[kk].im;
        const double s = scale[i];
        const double e0 = d0 / s, e1 = d1 / s, e2 = d2 / s;
        ss += (e0 * e0 + e1 * e1) + e2 * e2;
        const double w0 = W[i] + d0, w1 = W[n + i] + d1, w2 = W[2 * n + i] + d2;
        W[i] = w0; W[n + i] = w1; W[2 * n + i] = w2;
        const double z0 = (T00 * w0 + T01 * w1) + T02 * w2, z1 = (T10 * w0 + T11 * w1) + T12 * w2, z2 = (T20 * w0 + T21 * w1) + T22 * w2;
        Z[i] = z0; Z[n + i] = z1; Z[2 * n + i] = z2;
        const double yi = y[i];
        YS[i] = yi + z0; YS[n + i] = yi + z1; YS[2 * n + i] = yi + z2;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];   // the host's convergence tests start now ...
    wg_rhs<VD>(YS, F, 3, N, C, K, T);        // ... while the stage derivatives of the next iteration are evaluated (the barriers above made YS visible)
}

// The error estimate of a step (radau.py:466-478) in ONE launch for small grids: right-hand side fvec + Z^T E / h and y_new
// (error_rhs_kernel) straight into LDS, the real system's solve, then err, y + err and the scaled norm (error_norm_kernel with one
// workgroup) - three launches before.  Same operations, same reduction tree.
__global__ void __launch_bounds__(PCR_FUSED_THREADS) error_fused_kernel(const double* __restrict__ fvec, const double* __restrict__ Z, const double* __restrict__ y,
                                                                        int64_t N, double E0, double E1, double E2, double h, int nlevels, PcrSystem<double> Sr,
                                                                        double rtol, double atol, double* __restrict__ ynew, double* __restrict__ err,
                                                                        double* __restrict__ yerr, double* __restrict__ out, CrPlan pl, CrSystem<double> Cr)
{
    __shared__ double lds[3 * PCR_FUSED_MAX];
    __shared__ double red[PCR_FUSED_THREADS];
    const int64_t n = NF * N;
    for (int64_t kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // error_rhs_kernel
        const int64_t i = to_field_major(kk, N);
        const double ZE = ((Z[i] * E0 + Z[n + i] * E1) + Z[2 * n + i] * E2) / h;
        lds[kk] = fvec[i] + ZE;
        ynew[i] = y[i] + Z[2 * n + i];
    }
    __syncthreads();
    double* x = lds + 2 * PCR_FUSED_MAX;
    if (pl.k == 0) pcr_solve_all<double>(N, nlevels, Sr, lds, x, lds);   // (staged onto itself)
    else crpcr_solve_all<double>(pl, N, nlevels, Cr, Sr, nullptr, x, lds, true);
    __syncthreads();
    double ss = 0;
    for (int64_t kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // error_norm_kernel, one workgroup
        const int64_t i = to_field_major(kk, N);
        const double e = x[kk];
        const double a = fabs(y[i]), b = fabs(ynew[i]);
        const double s = atol + ((a > b || a != a) ? a : b) * rtol;
        const double q = e / s;
        ss += q * q;
        err[i] = e;
        yerr[i] = y[i] + e;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// What follows an accepted step, in ONE launch for small grids: f(y_new) (rhs_kernel), the dense-output coefficients Q = Z^T P
// (dense_q_kernel) and the seven monitors of y_new (monitors_kernel + reduce_records_kernel: extrema, exact) - four launches before.
// words: the zero-copy result block (marl_api.hip): g -> words[16 .. 23), then words[0] (the word the host waits for).
template <bool VD>
__global__ void __launch_bounds__(WG_THREADS) accept_fused_kernel(const double* __restrict__ ynew, double* __restrict__ fnew, const double* __restrict__ Z, int64_t N, P33 P,
                                                                  double* __restrict__ Q, const DevConsts* __restrict__ consts, double* __restrict__ words)
{
    __shared__ double red[NQ * WG_THREADS];
    __shared__ double tabs[TABLE_DOUBLES];
    __shared__ double g_new[7];
    const Tables T = load_tables(tabs, WG_THREADS);
    const DevConsts& C = consts[0];
    const HotConsts K = load_hot(&C);
    const int64_t n = NF * N;
    wg_rhs<VD>(ynew, fnew, 1, N, C, K, T);
    for (int64_t i = threadIdx.x; i < n; i += WG_THREADS) {   // dense_q_kernel
        const double z0 = Z[i], z1 = Z[n + i], z2 = Z[2 * n + i];
#pragma unroll
        for (int m = 0; m < 3; m++) Q[3 * i + m] = (z0 * P.p[0][m] + z1 * P.p[1][m]) + z2 * P.p[2][m];
    }
    wg_monitors(ynew, N, C, T, red, g_new);
    if (threadIdx.x == 0) {
        for (int e = 0; e < 7; e++) words[16 + e] = g_new[e];
        __threadfence_system();
        words[0] = 1.0;
    }
}

}  // namespace radau
}  // namespace marl