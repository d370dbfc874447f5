// marl_bdf_wg.h - solve_bdf_system (scipy/integrate/_ivp/bdf.py:36-68) of a small grid as ONE launch of one workgroup.
//
// On the reference's own grid (N = 200) a BDF step is sequential and tiny: the host-driven loop of bdf_run (marl_api.hip) spends a step
// on ~9 launches and ~4 waits for a result word - per Newton iteration one RHS launch, one launch of the fused linear algebra
// (bdf::newton_fused_kernel) and one wait, so that the host can apply the convergence tests.  Here the workgroup keeps going: predictor
// (or the restart after a Jacobian refresh), then up to NEWTON_MAXITER times { f(y) - wg_rhs, rhs_kernel's body; c f - psi - d; every
// cyclic-reduction level; y += dy, d += dy; the norm; the convergence tests on one lane }, and for the converged state the local error
// sum and the seven monitors (the host needs both for the accepted step).  One launch and one wait per solve.  Every vector operation
// is the one the launch kernels do, in their thread mapping and reduction order (bit-identical states); the scalar tests are bdf_run's,
// evaluated on the device (same IEEE operations; `pow` is the device library's).
#pragma once
#include "marl_bdf.h"
#include "marl_radau_wg.h"

namespace marl {
namespace bdf {

constexpr int NEWTON_MAXITER = 4;
// words of the result block (zero-copy host memory, doubles): the host waits for word 0, which is written last
enum : int { SW_DONE = 0, SW_FLAGS = 1, SW_ERR = 2, SW_ITERS = 4, SW_CONVERGED = 5, SW_G = 16 };

union SolveBuf {
    double solve[3 * radau::PCR_FUSED_MAX];      // ping-pong right-hand sides + the solution (48 KB)
    double red[NQ * radau::WG_THREADS];          // reductions (monitors: 64 KB)
};

// mode: 0 = predictor first (bdf.py:358-361 and y = y_predict, d = 0), 1 = restart (y = y_predict, d = 0), 2 = continue from y, d as they are
template <bool VD>
__global__ void __launch_bounds__(radau::WG_THREADS) solve_wg_kernel(int mode, const double* __restrict__ D, int order, Vec6 gamma, double alpha_order,
                                                                      double* __restrict__ ypred, double* __restrict__ psi, double* __restrict__ scale,
                                                                      double* __restrict__ ynew, double* __restrict__ d, double* __restrict__ f, int64_t N, double c,
                                                                      int nlevels, radau::PcrSystem<double> Sr, const DevConsts* __restrict__ consts,
                                                                      double newton_tol, double err_coef, double rtol, double atol, double* __restrict__ words,
                                                                      radau::CrPlan pl = radau::CrPlan{}, radau::CrSystem<double> Cr = radau::CrSystem<double>{})
{
    using namespace radau;
    __shared__ SolveBuf buf;
    __shared__ double red[WG_THREADS];
    __shared__ double tabs[TABLE_DOUBLES];
    __shared__ double g_new[7];
    __shared__ int s_verdict;
    const Tables T = load_tables(tabs, WG_THREADS);
    const DevConsts& C = consts[0];
    const HotConsts K = load_hot(&C);
    const int n = (int)(NF * N);
    const int tid = threadIdx.x;
    double* lds = buf.solve;

    if (mode == 0) {   // predict_kernel
        for (int i = tid; i < n; i += WG_THREADS) {
            double a = D[i], q = 0;
#pragma unroll
            for (int j = 1; j <= MAX_ORDER; j++)
                if (j <= order) {
                    const double dj = D[(int64_t)j * n + i];
                    a += dj;
                    q += dj * gamma.v[j];
                }
            ypred[i] = a;
            ynew[i] = a;
            d[i] = 0;
            scale[i] = atol + rtol * fabs(a);
            psi[i] = q / alpha_order;
        }
    } else if (mode == 1) {   // newton_restart_kernel
        for (int i = tid; i < n; i += WG_THREADS) {
            ynew[i] = ypred[i];
            d[i] = 0;
        }
    }
    __syncthreads();

    double dy_norm_old = -1, dy_ss = 0;
    int iters = 0, converged = 0, any_bad = 0;
    for (int k = 0; k < NEWTON_MAXITER; k++) {
        wg_rhs<VD>(ynew, f, 1, N, C, K, T);
        __syncthreads();
        iters = k + 1;
        // ---- newton_fused_kernel ----
        int bad = 0;
        for (int kk = tid; kk < n; kk += WG_THREADS) {
            const int64_t i = to_field_major(kk, N);
            const double fi = f[i];
            bad |= !isfinite(fi);
            lds[kk] = (c * fi - psi[i]) - d[i];
        }
        any_bad = __syncthreads_or(bad);
        if (any_bad) break;   // (bdf.py:44-45; uniform)
        double* x = lds + 2 * PCR_FUSED_MAX;
        if (pl.k == 0) {
            int cur = 0;
            for (int level = 0; level < nlevels; level++) {
                const double* b = lds + cur * PCR_FUSED_MAX;
                double* o = lds + (cur ^ 1) * PCR_FUSED_MAX;
                for (int kk = tid; kk < n; kk += WG_THREADS) pcr_solve_row<double>(N, kk, level, nlevels, Sr, b, o);
                __syncthreads();
                cur ^= 1;
            }
            for (int kk = tid; kk < n; kk += WG_THREADS) pcr_solve_row<double>(N, kk, nlevels, nlevels, Sr, lds + cur * PCR_FUSED_MAX, x);
        } else {
            crpcr_solve_all<double>(pl, N, nlevels, Cr, Sr, nullptr, x, lds, true);   // (the right-hand side is in lds already)
        }
        __syncthreads();
        double ss = 0;
        for (int kk = tid; kk < n; kk += WG_THREADS) {   // (newton_update_kernel with one workgroup)
            const int64_t i = to_field_major(kk, N);
            const double dy = x[kk];
            const double e = dy / scale[i];
            ss += e * e;
            ynew[i] += dy;
            d[i] += dy;
        }
        red[tid] = ss;
        __syncthreads();
        for (int s2 = 512; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] += red[tid + s2];
            __syncthreads();
        }
        dy_ss = red[0];
        // ---- the tests of solve_bdf_system (bdf.py:51-66), as bdf_run applies them to the norm it reads back ----
        if (tid == 0) {
            const double dy_norm = sqrt(dy_ss) / sqrt((double)n);
            double rate = -1;
            if (dy_norm_old >= 0) rate = dy_norm / dy_norm_old;
            int v = 0;
            if (rate >= 0 && (rate >= 1 || pow_small_int(rate, NEWTON_MAXITER - k) / (1 - rate) * dy_norm > newton_tol)) v = 2;
            else if (dy_norm == 0 || (rate >= 0 && rate / (1 - rate) * dy_norm < newton_tol)) v = 1;
            s_verdict = v;
        }
        __syncthreads();
        const int verdict = s_verdict;
        dy_norm_old = sqrt(dy_ss) / sqrt((double)n);
        if (verdict == 1) converged = 1;
        if (verdict != 0) break;
    }
    double err_ss = 0;
    if (converged) {
        // the local error norm of the converged state (bdf.py:398-400; scaled_norm_kernel with one workgroup)
        double es = 0;
        for (int64_t i = tid; i < n; i += WG_THREADS) {
            const double e = err_coef * d[i] / (atol + rtol * fabs(ynew[i]));
            es += e * e;
        }
        __syncthreads();
        red[tid] = es;
        __syncthreads();
        for (int s2 = 512; s2 > 0; s2 >>= 1) {
            if (tid < s2) red[tid] += red[tid + s2];
            __syncthreads();
        }
        err_ss = red[0];
        __syncthreads();
        wg_monitors(ynew, N, C, T, buf.red, g_new);   // what the accepted step's event tests need (monitors_kernel + reduce_records_kernel)
    }
    if (tid == 0) {
        int32_t* flags = reinterpret_cast<int32_t*>(words + SW_FLAGS);
        if (any_bad) *flags = 1;
        words[SW_ERR] = err_ss;
        words[SW_ITERS] = (double)iters;
        words[SW_CONVERGED] = (double)converged;
        if (converged)
            for (int e = 0; e < 7; e++) words[SW_G + e] = g_new[e];
        __threadfence_system();   // everything above is visible to the host before the word it waits for
        words[SW_DONE] = dy_ss;
    }
}

}  // namespace bdf
}  // namespace marl
