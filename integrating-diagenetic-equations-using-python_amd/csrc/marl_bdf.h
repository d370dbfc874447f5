// marl_bdf.h - vector kernels of the implicit BDF path (scipy/integrate/_ivp/bdf.py; the other method the reference's Solver names for
// its jac_sparsity, marlpde/parameters.py:205-219).  The finite-difference Jacobian, the block-tridiagonal factorisation / solves of
// I - c J (cyclic reduction, marl_radau.h) and the RHS are shared with the Radau path; the step logic runs on the host (marl_api.hip,
// bdf_run) like marl_integrate_radau's.  State vectors are field-major (the reference's layout), linear systems cell-major.
#pragma once
#include "marl_radau.h"

namespace marl {
namespace bdf {

constexpr int MAX_ORDER = 5;
struct Mat6 { double m[MAX_ORDER + 1][MAX_ORDER + 1]; };
struct Vec6 { double v[MAX_ORDER + 1]; };

// change_D (bdf.py:28-33): D[:order + 1] = (R U)^T D[:order + 1];  D: [MAX_ORDER + 3][n]
__global__ void __launch_bounds__(256) change_D_kernel(double* __restrict__ D, int64_t n, int order, Mat6 RU)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double d[MAX_ORDER + 1];
#pragma unroll
    for (int j = 0; j <= MAX_ORDER; j++) d[j] = j <= order ? D[(int64_t)j * n + i] : 0.0;
#pragma unroll
    for (int k = 0; k <= MAX_ORDER; k++) {
        if (k > order) break;
        double a = 0;
#pragma unroll
        for (int j = 0; j <= MAX_ORDER; j++)
            if (j <= order) a += RU.m[j][k] * d[j];
        D[(int64_t)k * n + i] = a;
    }
}

// y_predict = sum D[:order + 1]; scale = atol + rtol |y_predict|; psi = D[1:order + 1]^T gamma[1:order + 1] / alpha[order]  (bdf.py:358-361)
// and the start of solve_bdf_system: y = y_predict, d = 0
__global__ void __launch_bounds__(256) predict_kernel(const double* __restrict__ D, int64_t n, int order, Vec6 gamma, double alpha_order, double rtol,
                                                      double atol, double* __restrict__ ypred, double* __restrict__ scale, double* __restrict__ psi,
                                                      double* __restrict__ ynew, double* __restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
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

// restart of the Newton iteration after a Jacobian refresh: y = y_predict, d = 0
__global__ void __launch_bounds__(256) newton_restart_kernel(const double* __restrict__ ypred, int64_t n, double* __restrict__ ynew, double* __restrict__ d)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    ynew[i] = ypred[i];
    d[i] = 0;
}

// right-hand side of the Newton system, cell-major:  c f - psi - d  (bdf.py:48); flags[0] |= a non-finite f
__global__ void __launch_bounds__(256) newton_rhs_kernel(const double* __restrict__ f, const double* __restrict__ psi, const double* __restrict__ d, int64_t N,
                                                         double c, double* __restrict__ rhs, int32_t* __restrict__ flags)
{
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= NF * N) return;
    const int64_t i = radau::to_field_major(kk, N);
    const double fi = f[i];
    if (!isfinite(fi)) *flags = 1;
    rhs[kk] = (c * fi - psi[i]) - d[i];
}

// dy = the solved system (cell-major); out[block] = sum (dy / scale)^2;  y += dy, d += dy  (bdf.py:49, 58-59; when the iteration is
// abandoned after this norm, y and d are not used any more)
__global__ void __launch_bounds__(1024) newton_update_kernel(const double* __restrict__ dy_cm, const double* __restrict__ scale, int64_t N,
                                                             double* __restrict__ ynew, double* __restrict__ d, double* __restrict__ out)
{
    __shared__ double red[1024];
    const int64_t n = NF * N;
    double ss = 0;
    for (int64_t kk = (int64_t)blockIdx.x * 1024 + threadIdx.x; kk < n; kk += (int64_t)gridDim.x * 1024) {
        const int64_t i = radau::to_field_major(kk, N);
        const double dy = dy_cm[kk];
        const double e = dy / scale[i];
        ss += e * e;
        ynew[i] += dy;
        d[i] += dy;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// One Newton iteration's linear algebra in ONE launch for small systems (5 N <= PCR_FUSED_MAX; the real system only, so one workgroup
// holds it all): right-hand side (newton_rhs_kernel) staged straight into LDS, every cyclic-reduction level (pcr_solve_all), then the
// update and its norm (newton_update_kernel) - the same arithmetic in the same order, three launches less per iteration.
__global__ void __launch_bounds__(radau::PCR_FUSED_THREADS) newton_fused_kernel(const double* __restrict__ f, const double* __restrict__ psi, int64_t N, double c,
                                                                                  int nlevels, radau::PcrSystem<double> Sr, const double* __restrict__ scale,
                                                                                  double* __restrict__ ynew, double* __restrict__ d, int32_t* __restrict__ flags,
                                                                                  double* __restrict__ out, double err_coef = 0.0, double rtol = 0.0, double atol = 0.0,
                                                                                  double* __restrict__ err_out = nullptr, radau::CrPlan pl = radau::CrPlan{},
                                                                                  radau::CrSystem<double> Cr = radau::CrSystem<double>{})
{
    using namespace radau;
    __shared__ double lds[2 * PCR_FUSED_MAX + PCR_FUSED_MAX];   // ping-pong right-hand sides + the solution
    __shared__ double red[PCR_FUSED_THREADS];
    const int n = (int)(NF * N);
    int bad = 0;
    for (int kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {
        const int64_t i = to_field_major(kk, N);
        const double fi = f[i];
        bad |= !isfinite(fi);
        lds[kk] = (c * fi - psi[i]) - d[i];
    }
    // The non-finite flag and out[0] may both live in host-coherent memory that the host POLLS (marl_api.hip radau_read): it waits
    // for out[0] and then trusts the flag.  A workgroup barrier does not order stores of different waves to system memory, so the
    // flag is folded through the barrier and written by the one thread that later writes out[0], with a system-scope fence between.
    const int any_bad = __syncthreads_or(bad);
    double* x = lds + 2 * PCR_FUSED_MAX;
    if (pl.k == 0) {
        int cur = 0;
        for (int level = 0; level < nlevels; level++) {
            const double* b = lds + cur * PCR_FUSED_MAX;
            double* o = lds + (cur ^ 1) * PCR_FUSED_MAX;
            for (int k = threadIdx.x; k < n; k += PCR_FUSED_THREADS) pcr_solve_row<double>(N, k, level, nlevels, Sr, b, o);
            __syncthreads();
            cur ^= 1;
        }
        for (int k = threadIdx.x; k < n; k += PCR_FUSED_THREADS) pcr_solve_row<double>(N, k, nlevels, nlevels, Sr, lds + cur * PCR_FUSED_MAX, x);
    } else {
        crpcr_solve_all<double>(pl, N, nlevels, Cr, Sr, nullptr, x, lds, true);   // (the right-hand side is in lds already)
    }
    __syncthreads();
    double ss = 0;
    for (int kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // (newton_update_kernel with one workgroup)
        const int64_t i = to_field_major(kk, N);
        const double dy = x[kk];
        const double e = dy / scale[i];
        ss += e * e;
        ynew[i] += dy;
        d[i] += dy;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    const double dy_ss = red[0];
    // err_out: the local error norm this iteration's state WOULD have if the iteration turns out to be the converged one
    // (bdf.py:398-400: error = error_const[order] d, scale = atol + rtol |y_new|) - the same terms, thread mapping and reduction tree
    // as scaled_norm_kernel with one workgroup (bit-identical), so that the host needs neither a launch nor a wait for it
    if (err_out) {
        __syncthreads();   // every thread's d / ynew updates are visible; red[] is free again
        double es = 0;
        for (int64_t i = threadIdx.x; i < n; i += PCR_FUSED_THREADS) {
            const double e = err_coef * d[i] / (atol + rtol * fabs(ynew[i]));
            es += e * e;
        }
        red[threadIdx.x] = es;
        __syncthreads();
        for (int s2 = 512; s2 > 0; s2 >>= 1) {
            if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        if (any_bad) *flags = 1;
        if (err_out) *err_out = red[0];
        __threadfence_system();   // the flag and the error sum are visible to the host before the word it waits for
        out[0] = dy_ss;
    }
}

// out[block] = sum (coef v / (atol + rtol |yref|))^2   (the error norms of bdf.py:398-400, 428-436)
__global__ void __launch_bounds__(1024) scaled_norm_kernel(const double* __restrict__ v, double coef, const double* __restrict__ yref, double rtol, double atol,
                                                           int64_t n, double* __restrict__ out)
{
    __shared__ double red[1024];
    double ss = 0;
    for (int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 1024) {
        const double e = coef * v[i] / (atol + rtol * fabs(yref[i]));
        ss += e * e;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// The two error norms of the order selection (bdf.py:428-436: error_m from D[order], error_p from D[order + 2]) in ONE launch of one
// workgroup - each with the loop and the reduction tree of scaled_norm_kernel (bit-identical), one launch and one wait less per selection.
// out2 is written before out[0] (the word the host polls), with a system-scope fence in between.
__global__ void __launch_bounds__(1024) scaled_norm2_kernel(const double* __restrict__ v1, double coef1, const double* __restrict__ v2, double coef2,
                                                            const double* __restrict__ yref, double rtol, double atol, int64_t n, double* __restrict__ out,
                                                            double* __restrict__ out2)
{
    __shared__ double red[1024];
    double r[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const double* v = k ? v2 : v1;
        const double coef = k ? coef2 : coef1;
        double ss = 0;
        for (int64_t i = threadIdx.x; i < n; i += 1024) {
            const double e = coef * v[i] / (atol + rtol * fabs(yref[i]));
            ss += e * e;
        }
        __syncthreads();
        red[threadIdx.x] = ss;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        r[k] = red[0];
    }
    if (threadIdx.x == 0) {
        *out2 = r[1];
        __threadfence_system();
        out[0] = r[0];
    }
}

// the accepted step's update of the differences (bdf.py:419-422); y = y_new
__global__ void __launch_bounds__(256) accept_kernel(double* __restrict__ D, const double* __restrict__ d, const double* __restrict__ ynew, int64_t n, int order,
                                                     double* __restrict__ y)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double di = d[i];
    D[(int64_t)(order + 2) * n + i] = di - D[(int64_t)(order + 1) * n + i];
    D[(int64_t)(order + 1) * n + i] = di;
    double above = di;
    for (int j = order; j >= 0; j--) {
        const double v = D[(int64_t)j * n + i] + above;
        D[(int64_t)j * n + i] = v;
        above = v;
    }
    y[i] = ynew[i];
}

// D[0] = y0, D[1] = f h  (bdf.py:251-254)
__global__ void __launch_bounds__(256) init_D_kernel(const double* __restrict__ y, const double* __restrict__ f, double h, int64_t n, double* __restrict__ D)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    D[i] = y[i];
    D[n + i] = f[i] * h;
}

// BdfDenseOutput._call_impl (bdf.py:465-478): out = D[0] + D[1:order + 1]^T p,  p = cumprod((t - t_shift) / denom)  (computed by the host)
__global__ void __launch_bounds__(256) dense_kernel(const double* __restrict__ D, int64_t n, int order, Vec6 p, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double a = 0;
#pragma unroll
    for (int j = 0; j < MAX_ORDER; j++)
        if (j < order) a += D[(int64_t)(j + 1) * n + i] * p.v[j];
    out[i] = a + D[i];
}

}  // namespace bdf
}  // namespace marl
