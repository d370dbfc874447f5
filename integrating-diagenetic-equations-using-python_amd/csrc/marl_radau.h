// marl_radau.h - device side of the implicit path: scipy's Radau IIA (order 5) as the reference runs it by default
// (marlpde/parameters.py:213 method "Radau" with the 27-diagonal jac_sparsity of :150-199; call site
// marlpde/Evolve_scenario.py:104-109; algorithm scipy/integrate/_ivp/radau.py and num_jac of common.py:268-451).
//
// The step logic (Newton convergence tests, step-size prediction, Jacobian / factorisation reuse) is scalar work and
// lives on the host (marl_api.hip, radau_run); everything that touches a vector or the Jacobian is a kernel here:
//   * finite-difference Jacobian over column groups: perturbed states for ALL groups are evaluated by ONE batched launch of
//     rhs_kernel; the per-column bookkeeping of num_jac (difference quality test, second trial step, step-factor adaptation)
//     runs one thread per column;
//   * the Jacobian is kept as N x 3 blocks of 5 x 5 (cell-major unknowns make mu I - J block tridiagonal); the real and the
//     complex system of the Radau collocation equations are factorised by block Thomas elimination with partial pivoting
//     inside the 5 x 5 diagonal blocks (Gauss-Jordan, explicit block inverses), real and complex concurrently in two
//     workgroups; the elimination is sequential in depth - one lane per system (a sweep of instances maps to lanes);
//   * Newton right-hand sides, updates, norms, dense output are element-wise kernels.
#pragma once
#include "marl_kernels.h"

namespace marl {
namespace radau {

constexpr double EPS = 2.220446049250313e-16;
// radau.py:11-44
constexpr double T00 = 0.09443876248897524, T01 = -0.14125529502095421, T02 = 0.03002919410514742;
constexpr double T10 = 0.25021312296533332, T11 = 0.20412935229379994, T12 = -0.38294211275726192;
constexpr double T20 = 1, T21 = 1, T22 = 0;
constexpr double TI00 = 4.17871859155190428, TI01 = 0.32768282076106237, TI02 = 0.52337644549944951;
constexpr double TI10 = -4.17871859155190428, TI11 = -0.32768282076106237, TI12 = 0.47662355450055044;
constexpr double TI20 = 0.50287263494578682, TI21 = -2.57192694985560522, TI22 = 0.59603920482822492;

struct cplx {
    double re, im;
};
__host__ __device__ inline cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__host__ __device__ inline cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__host__ __device__ inline cplx operator*(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__host__ __device__ inline cplx operator*(double a, cplx b) { return {a * b.re, a * b.im}; }
__device__ inline double abs1(double a) { return fabs(a); }
__device__ inline double abs1(cplx a) { return fabs(a.re) + fabs(a.im); }   // LAPACK's cabs1
__device__ inline double recip(double a) { return 1.0 / a; }
__device__ inline cplx recip(cplx a) { const double d = 1.0 / (a.re * a.re + a.im * a.im); return {a.re * d, -a.im * d}; }
// acc + a b and a b with the contraction spelled out: the solve kernels (one launch per level / all levels in one launch) must round
// alike, and left to itself the compiler contracts the complex multiply-adds differently from kernel to kernel
__device__ __forceinline__ double madd(double acc, double a, double b) { return __builtin_fma(a, b, acc); }
// (the form the compiler chose for the per-level kernel, whose results the implicit path's tests were pinned with)
__device__ __forceinline__ cplx mul1(cplx a, cplx b) { return {__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.im, b.re, a.re * b.im)}; }
__device__ __forceinline__ cplx madd(cplx acc, cplx a, cplx b)
{
    const cplx p = mul1(a, b);
    return {acc.re + p.re, acc.im + p.im};
}
__device__ __forceinline__ double mul1(double a, double b) { return a * b; }
// (a0 b0 + a1 b1) + a2 b2 and the complex product, SPELLED OUT for the batched kernels (marl_radau_batch.h) and the
// one-workgroup-per-instance integrator (marl_radau_wg.h) with the contraction the compiler chose in the batched launch kernels of
// round 2 (read off their machine code: the middle product is rounded, the first and the last are fused; re = fma(a.re, b.re,
// -(a.im b.im)), im = fma(a.re, b.im, a.im b.re)).  Written as `a*b + c*d` the choice is the compiler's and depends on what the
// expression is inlined into: the workgroup integrator got other roundings than the launch kernels from the same source text, and a
// last-bit difference in an error norm flips Newton / step-size decisions later on.  With these helpers the sweep paths (launch per
// action, hybrid, all-in-workgroup) give the same bits.  The single-run kernels below keep their source text - and with it the
// roundings their scipy-equal statistics were pinned with (newton_rhs_kernel's complex product, for one, is contracted the other way
// round than the batched kernel's: im = fma(a.im, b.re, a.re b.im)).
__device__ __forceinline__ double dot3(double a0, double b0, double a1, double b1, double a2, double b2)
{
    return __builtin_fma(a2, b2, __builtin_fma(a0, b0, a1 * b1));
}
__device__ __forceinline__ cplx cmul_ref(cplx a, cplx b) { return {__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.re, b.im, a.im * b.re)}; }
__device__ inline double lift(double a, double) { return a; }
__device__ inline cplx lift(double a, cplx) { return {a, 0.0}; }

// Batched sweeps of Radau instances (marl_radau_batch.h): every instance follows its own control flow, advanced one ACTION per
// cycle by a device-side controller; the kernels below serve the single-instance driver (ctl == NULL, scalar arguments) and the
// batched one (blockIdx.z = instance; per-instance scalars from ctl[z]; ZBatch masks out instances that need something else).
enum : int32_t { A_RHS_Y = 1, A_JAC = 2, A_LU = 4, A_NEWTON = 8, A_ERR = 16, A_ACCEPT = 32, A_ERR2 = 64, A_DENSE = 128 };
struct RadauCtl {
    // set once
    double t_bound, rtol, atol, newton_tol;
    int64_t max_attempts;
    // Radau._step_impl state (radau.py:404-537); a value < 0 in *_old / *_o encodes None
    double t, S_h_abs, S_h_abs_old, S_err_old, h_abs, h_abs_o, err_o, min_step, h, t_new;
    double rate, dW_norm_old, error_norm, safety, factor;
    double sol_t_old, sol_h;
    double g[7];
    int64_t n_events[7];
    int64_t nfev, njev, nlu, n_acc, n_rej, attempts;
    // what the data-parallel kernels of this cycle read
    double mu_r, mu_c_re, mu_c_im;   // MU_REAL / h, MU_COMPLEX / h
    double x3[3];                    // dense-output abscissae of the three collocation points (Z0 prediction)
    int32_t action;                  // A_* bits: what this instance needs in this cycle (0: nothing - finished)
    int32_t pc, status;
    int32_t current_jac, have_lu, have_sol, have_factor, rejected, k, n_iter, recompute_jac, newton_begin, err_second;
    // results the kernels leave for the controller
    double sumsq;                    // sum of squares of the last norm kernel
    int32_t nonfinite, pad;
    // event root finding inside a sweep (round 3; ivp.py:673-694 + solve_event_equation :51-76, Brent as in scipy's brentq): a resumable
    // state machine - every function evaluation is one A_DENSE action (dense output at dense_x, monitors of that state)
    int32_t locate_events, ev_pending, br_e, br_phase, br_iter, pad2;   // ev_pending: bit e = monitor e changed sign in the step just accepted
    int64_t max_events;
    double dense_x;                  // (t - sol_t_old) / sol_h of the state the A_DENSE kernels evaluate
    double br_a, br_b, br_fa, xpre, xcur, xblk, fpre, fcur, fblk, spre, scur;
};
__device__ __forceinline__ const RadauCtl* ctl_of(const ZBatch& B)
{
    return reinterpret_cast<const RadauCtl*>(reinterpret_cast<const char*>(B.act) - offsetof(RadauCtl, action) + z_inst(B) * B.act_stride);
}

// cell-major index kk = 5 i + f  <->  field-major index f N + i
__device__ __forceinline__ int64_t to_field_major(int64_t kk, int64_t N) { return (kk % NF) * N + kk / NF; }

// is (row field f) x (column field fp) in the reference's pattern?  (parameters.py:197 zeroes the CA, CC rows x Phi columns)
__device__ __forceinline__ bool in_pattern(int f, int fp) { return !(f < 2 && fp == 4); }

// ---- finite-difference Jacobian (num_jac, common.py:268-344; _sparse_num_jac :389-451) -----------------------------
// step sizes + the perturbed state of every group:  YP[g][j] = y[j] + (groups[j] == g ? h[j] : 0)
__global__ void __launch_bounds__(256) fd_prepare_kernel(const double* __restrict__ y, const double* __restrict__ f0, double* __restrict__ factor,
                                                         double threshold, int first, const int32_t* __restrict__ groups, int ng, int64_t n,
                                                         double* __restrict__ h, double* __restrict__ yscale, double* __restrict__ YP,
                                                         ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    if (B.act) first = !ctl_of(B)->have_factor;
    y = z_shift(y, B); f0 = z_shift(f0, B); factor = z_shift(factor, B); h = z_shift(h, B); yscale = z_shift(yscale, B); YP = z_shift(YP, B);
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double fac = first ? sqrt(EPS) : factor[j];
    const double yj = y[j];
    const double f_sign = (f0[j] >= 0) ? 1.0 : -1.0;
    const double ay = fabs(yj);
    const double ys = f_sign * (threshold > ay ? threshold : ay);
    double hj = (yj + fac * ys) - yj;
    while (hj == 0) { fac *= 10; hj = (yj + fac * ys) - yj; }
    factor[j] = fac; h[j] = hj; yscale[j] = ys;
    const int gj = groups[j];
    for (int g = 0; g < ng; g++) YP[(int64_t)g * n + j] = (g == gj) ? yj + hj : yj;
}

// |diff| column of one perturbed column j = (fp, ip) in scipy's csc row order; returns max |diff|, FIRST arg max (row index)
__device__ __forceinline__ double fd_column(const double* __restrict__ f0, const double* __restrict__ fg, int fp, int64_t ip, int64_t N,
                                            double (&dcol)[15], int64_t& arg)
{
    double best = 0;
    arg = -1;
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int di = 0; di < 3; di++) {
            const int64_t i = ip + di - 1;
            double d = 0;
            if (i >= 0 && i < N && in_pattern(f, fp)) {
                const int64_t r = f * N + i;
                d = fg[r] - f0[r];
                if (arg < 0 || fabs(d) > best) { best = fabs(d); arg = r; }
            }
            dcol[f * 3 + di] = d;
        }
    if (best == 0) arg = 0;   // scipy's sparse argmax reports row 0 for an all-zero column (scipy/sparse/_data.py:265-272)
    return best;
}

// first pass over the columns: differences, their quality, the second trial step of the columns whose difference drowned
__global__ void __launch_bounds__(256) fd_columns_kernel(const double* __restrict__ y, const double* __restrict__ f0, const double* __restrict__ FN,
                                                         const int32_t* __restrict__ groups, int ng, int64_t N, const double* __restrict__ factor,
                                                         const double* __restrict__ yscale, double* __restrict__ Jraw, double* __restrict__ maxdiff,
                                                         double* __restrict__ scl, int32_t* __restrict__ small, double* __restrict__ hnew,
                                                         double* __restrict__ YP2, ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    y = z_shift(y, B); f0 = z_shift(f0, B); FN = z_shift(FN, B); factor = z_shift(factor, B); yscale = z_shift(yscale, B); Jraw = z_shift(Jraw, B);
    maxdiff = z_shift(maxdiff, B); scl = z_shift(scl, B); small = z_shift(small, B); hnew = z_shift(hnew, B); YP2 = z_shift(YP2, B);
    const int64_t n = NF * N;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int fp = (int)(j / N);
    const int64_t ip = j % N;
    const int gj = groups[j];
    const double* fg = FN + (int64_t)gj * n;
    double dcol[15];
    int64_t arg;
    const double md = fd_column(f0, fg, fp, ip, N, dcol, arg);
    const double a = fabs(f0[arg]), b = fabs(fg[arg]);
    const double sc = a > b ? a : b;
#pragma unroll
    for (int e = 0; e < 15; e++) Jraw[j * 15 + e] = dcol[e];
    maxdiff[j] = md; scl[j] = sc;
    const bool sm = md < pow(EPS, 0.875) * sc;   // NUM_JAC_DIFF_REJECT
    small[j] = sm ? 1 : 0;
    const double yj = y[j];
    const double hn = sm ? (yj + (10 * factor[j]) * yscale[j]) - yj : 0.0;
    hnew[j] = hn;
    for (int g = 0; g < ng; g++) YP2[(int64_t)g * n + j] = (g == gj) ? yj + hn : yj;
}

// second pass: adopt the larger step where it resolves the column better, divide by h, adapt the factors, scatter into blocks
//   J[((i*3 + d)*5 + f)*5 + fp] = d rate(f, i) / d y(fp, i + d - 1)
__global__ void __launch_bounds__(256) fd_finish_kernel(const double* __restrict__ f0, const double* __restrict__ FN2, const int32_t* __restrict__ groups,
                                                        int64_t N, double* __restrict__ factor, double* __restrict__ h, const double* __restrict__ maxdiff,
                                                        const double* __restrict__ scl, const int32_t* __restrict__ small, const double* __restrict__ hnew,
                                                        const double* __restrict__ Jraw, double* __restrict__ J, ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    f0 = z_shift(f0, B); FN2 = z_shift(FN2, B); factor = z_shift(factor, B); h = z_shift(h, B); maxdiff = z_shift(maxdiff, B); scl = z_shift(scl, B);
    small = z_shift(small, B); hnew = z_shift(hnew, B); Jraw = z_shift(Jraw, B); J = z_shift(J, B);
    const int64_t n = NF * N;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int fp = (int)(j / N);
    const int64_t ip = j % N;
    double dcol[15];
#pragma unroll
    for (int e = 0; e < 15; e++) dcol[e] = Jraw[j * 15 + e];
    double md = maxdiff[j], sc = scl[j], fac = factor[j], hj = h[j];
    if (small[j]) {
        const double* fg = FN2 + (int64_t)groups[j] * n;
        double dnew[15];
        int64_t arg;
        const double md_new = fd_column(f0, fg, fp, ip, N, dnew, arg);
        const double a = fabs(f0[arg]), b = fabs(fg[arg]);
        const double sc_new = a > b ? a : b;
        if (md * sc_new < md_new * sc) {
            fac = 10 * fac; hj = hnew[j]; sc = sc_new; md = md_new;
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
    if (md < pow(EPS, 0.75) * sc) fac *= 10;     // NUM_JAC_DIFF_SMALL -> FACTOR_INCREASE
    if (md > pow(EPS, 0.25) * sc) fac *= 0.1;    // NUM_JAC_DIFF_BIG   -> FACTOR_DECREASE
    if (fac < 1e3 * EPS) fac = 1e3 * EPS;        // NUM_JAC_MIN_FACTOR
    factor[j] = fac; h[j] = hj;
}

#ifdef MARL_LAB_BLOCK_THOMAS   // the first implementation (one lane per system: 199 VGPRs, 832 B/lane of scratch), a cross-check for lab builds only
// ---- block-tridiagonal LU of  mu I - J  (cell-major ordering) -----------------------------------------------------------
// Block Thomas: D'_0 = D_0; Up_i = D'_i^-1 U_i; D'_{i+1} = D_{i+1} - L_{i+1} Up_i, with D_i = mu I - J[i][1], L_i = -J[i][0],
// U_i = -J[i][2].  D'_i is inverted by Gauss-Jordan with partial pivoting.  Stored: Dinv[i] (5x5), Up[i] (5x5).
template <class T>
__device__ void factor_system(const double* __restrict__ J, int64_t N, T mu, T* __restrict__ Dinv, T* __restrict__ Up)
{
    T Uprev[NF][NF];
    for (int64_t i = 0; i < N; i++) {
        T A[NF][2 * NF];
        const double* Jd = J + (i * 3 + 1) * 25;
        const double* Jl = J + (i * 3 + 0) * 25;
#pragma unroll
        for (int r = 0; r < NF; r++)
#pragma unroll
            for (int c = 0; c < NF; c++) {
                T a = lift(-Jd[r * NF + c], mu);
                if (r == c) a = a + mu;
                if (i > 0) {
#pragma unroll
                    for (int k = 0; k < NF; k++) a = a + Jl[r * NF + k] * Uprev[k][c];   // - L Up,  L = -Jl
                }
                A[r][c] = a;
                A[r][NF + c] = lift(r == c ? 1.0 : 0.0, mu);
            }
#pragma unroll
        for (int k = 0; k < NF; k++) {
            int p = k;
            double best = abs1(A[k][k]);
#pragma unroll
            for (int r = k + 1; r < NF; r++) {
                const double a = abs1(A[r][k]);
                if (a > best) { best = a; p = r; }
            }
#pragma unroll
            for (int r = k + 1; r < NF; r++) {
                const bool s = (p == r);
#pragma unroll
                for (int c = 0; c < 2 * NF; c++) {
                    const T t = A[k][c];
                    A[k][c] = s ? A[r][c] : t;
                    A[r][c] = s ? t : A[r][c];
                }
            }
            const T inv = recip(A[k][k]);
#pragma unroll
            for (int c = 0; c < 2 * NF; c++) A[k][c] = A[k][c] * inv;
#pragma unroll
            for (int r = 0; r < NF; r++) {
                if (r == k) continue;
                const T m = A[r][k];
#pragma unroll
                for (int c = 0; c < 2 * NF; c++) A[r][c] = A[r][c] - m * A[k][c];
            }
        }
#pragma unroll
        for (int r = 0; r < NF; r++)
#pragma unroll
            for (int c = 0; c < NF; c++) Dinv[i * 25 + r * NF + c] = A[r][NF + c];
        if (i < N - 1) {
            const double* Ju = J + (i * 3 + 2) * 25;
#pragma unroll
            for (int r = 0; r < NF; r++)
#pragma unroll
                for (int c = 0; c < NF; c++) {
                    T a = lift(0.0, mu);
#pragma unroll
                    for (int k = 0; k < NF; k++) a = a - Ju[k * NF + c] * A[r][NF + k];   // Dinv * U,  U = -Ju
                    Uprev[r][c] = a;
                    Up[i * 25 + r * NF + c] = a;
                }
        }
    }
}

// x (cell-major, in place): forward  b'_i = Dinv_i (b_i - L_i b'_{i-1}),  backward  x_i = b'_i - Up_i x_{i+1}
template <class T>
__device__ void solve_system(const double* __restrict__ J, int64_t N, const T* __restrict__ Dinv, const T* __restrict__ Up, T* __restrict__ x)
{
    T prev[NF];
    for (int64_t i = 0; i < N; i++) {
        T t[NF];
        const double* Jl = J + (i * 3 + 0) * 25;
#pragma unroll
        for (int r = 0; r < NF; r++) {
            T a = x[i * NF + r];
            if (i > 0) {
#pragma unroll
                for (int k = 0; k < NF; k++) a = a + Jl[r * NF + k] * prev[k];
            }
            t[r] = a;
        }
#pragma unroll
        for (int r = 0; r < NF; r++) {
            T a = Dinv[i * 25 + r * NF] * t[0];
#pragma unroll
            for (int k = 1; k < NF; k++) a = a + Dinv[i * 25 + r * NF + k] * t[k];
            prev[r] = a;
        }
#pragma unroll
        for (int r = 0; r < NF; r++) x[i * NF + r] = prev[r];
    }
    for (int64_t i = N - 2; i >= 0; i--) {
        T cur[NF];
#pragma unroll
        for (int r = 0; r < NF; r++) {
            T a = x[i * NF + r];
#pragma unroll
            for (int k = 0; k < NF; k++) a = a - Up[i * 25 + r * NF + k] * prev[k];
            cur[r] = a;
        }
#pragma unroll
        for (int r = 0; r < NF; r++) { prev[r] = cur[r]; x[i * NF + r] = cur[r]; }
    }
}

// block 0: real system (mu_r), block 1: complex system (mu_c) - concurrently
__global__ void __launch_bounds__(64) factor_kernel(const double* __restrict__ J, int64_t N, double mu_r, cplx mu_c, double* __restrict__ Dinv_r,
                                                    double* __restrict__ Up_r, cplx* __restrict__ Dinv_c, cplx* __restrict__ Up_c)
{
    if (threadIdx.x != 0) return;
    if (blockIdx.x == 0) factor_system<double>(J, N, mu_r, Dinv_r, Up_r);
    else factor_system<cplx>(J, N, mu_c, Dinv_c, Up_c);
}

// which: bit 0 real system, bit 1 complex system.  Results scattered to dW (field-major): dW[0] = x_r, dW[1] = Re x_c, dW[2] = Im x_c
__global__ void __launch_bounds__(64) solve_kernel(const double* __restrict__ J, int64_t N, const double* __restrict__ Dinv_r, const double* __restrict__ Up_r,
                                                   const cplx* __restrict__ Dinv_c, const cplx* __restrict__ Up_c, double* __restrict__ rhs_r,
                                                   cplx* __restrict__ rhs_c, int which)
{
    if (threadIdx.x != 0) return;
    const int sys = (which == 3) ? (int)blockIdx.x : (which == 1 ? 0 : 1);
    if (sys == 0) solve_system<double>(J, N, Dinv_r, Up_r, rhs_r);
    else solve_system<cplx>(J, N, Dinv_c, Up_c, rhs_c);
}


// ---- the same systems by block PARALLEL CYCLIC REDUCTION (the default): parallel over depth -----------------------------------
// Level l (stride s = 2^l) combines equation i with equations i -+ s so that it couples to cells i -+ 2s only:
//   alpha_i = -L_i D_{i-s}^-1,  gamma_i = -U_i D_{i+s}^-1
//   D'_i = D_i + alpha_i U_{i-s} + gamma_i L_{i+s};   L'_i = alpha_i L_{i-s};   U'_i = gamma_i U_{i+s};   b'_i = b_i + alpha_i b_{i-s} + gamma_i b_{i+s}
// After ceil(log2 N) levels every equation stands alone: x_i = D_i^-1 b_i.  Factorisation = the alpha / gamma blocks of every level
// and the last D^-1 (one thread per cell per level; the 5 x 5 inverses by Gauss-Jordan with partial pivoting); a solve = the b
// recurrences (one thread per unknown per level).  One launch per level: a grid-wide dependency sits between levels.
// Accuracy against dense LU on this model's matrices (h = 1e-6 ... 0.1, condition up to 1e20): <= 6e-10 relative.
// storage of one system: blocks are 25 T per cell
#endif   // MARL_LAB_BLOCK_THOMAS

template <class T>
struct PcrSystem {
    T* L[2]; T* D[2]; T* U[2]; T* Dinv[2];   // ping-pong sets of the level in progress
    T* alpha; T* gamma;                      // [level][cell][25]
    T* b[2];                                 // ping-pong right-hand sides, cell-major
};

template <class T>
__device__ __forceinline__ PcrSystem<T> z_shift_system(PcrSystem<T> S, const ZBatch& B)
{
#pragma unroll
    for (int k = 0; k < 2; k++) { S.L[k] = z_shift(S.L[k], B); S.D[k] = z_shift(S.D[k], B); S.U[k] = z_shift(S.U[k], B); S.Dinv[k] = z_shift(S.Dinv[k], B); S.b[k] = z_shift(S.b[k], B); }
    S.alpha = z_shift(S.alpha, B); S.gamma = z_shift(S.gamma, B);
    return S;
}

// Factorisation with ONE LANE PER BLOCK ELEMENT: a cell's 5 x 5 blocks are spread over 25 lanes of a 32-lane group (8 cells per
// 256-thread workgroup); block products and the Gauss-Jordan inverse read their operands' rows / columns from a small LDS stage.
// (The first version gave a whole cell to one lane - ~250 live complex values, 1.7 KB of scratch, 73 us per level at N = 200, 73 % of
// the GPU time of a run; this one issues ~150 instructions per lane and level.)
constexpr int PCR_CELLS_PER_BLOCK = 8;

template <class T>
struct PcrStage {   // per cell group, in LDS
    T A[25], B[25], C[25];
};

// this lane's element (r, c) of  A * B
template <class T>
__device__ __forceinline__ T mm_elem(const T* A, const T* B, int r, int c)
{
    T acc = A[r * NF] * B[c];
#pragma unroll
    for (int k = 1; k < NF; k++) acc = acc + A[r * NF + k] * B[k * NF + c];
    return acc;
}

// A cell's 25 lanes and their LDS stage belong to ONE 32-lane group, i.e. to one wave: the exchanges through the stage need no
// workgroup barrier - a wave's LDS operations execute in program order, so a fence at wavefront scope (no instruction; it only
// stops the compiler from moving the accesses) is the whole synchronisation.  (Round 3: the workgroup barriers that stood here
// cost 0.38 us each in the 1024-thread one-workgroup-per-instance integrator - 19 per group call, 92 % of a factorisation.)
__device__ __forceinline__ void group_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Gauss-Jordan inverse with partial pivoting; d = this lane's element of D (destroyed), returns its element of D^-1.
// Every lane of the group's wave calls it; lanes with e >= 25 idle.
template <class T>
__device__ __forceinline__ T gj_inverse_elem(T d, int e, int r, int c, bool act, PcrStage<T>& st)
{
    T v = lift(r == c ? 1.0 : 0.0, d);
#pragma unroll
    for (int k = 0; k < NF; k++) {
        group_sync();
        if (act) { st.A[e] = d; st.B[e] = v; }
        group_sync();
        if (act) {
            int p = k;
            double best = abs1(st.A[k * NF + k]);
#pragma unroll
            for (int rr = k + 1; rr < NF; rr++) {
                const double a = (rr > k) ? abs1(st.A[rr * NF + k]) : -1.0;
                if (a > best) { best = a; p = rr; }
            }
            // rows k and p swapped: this lane's row index in the unswapped matrix
            const int src = (r == k) ? p : ((r == p) ? k : r);
            const T dk = st.A[p * NF + c], vk = st.B[p * NF + c];          // (swapped) pivot row, this column
            const T pv = recip(st.A[p * NF + k]);                          // 1 / pivot
            const T dk_s = dk * pv, vk_s = vk * pv;                        // scaled pivot row
            if (r == k) { d = dk_s; v = vk_s; }
            else {
                const T m = st.A[src * NF + k];                            // this row's entry in the pivot column
                d = st.A[src * NF + c] - m * dk_s;
                v = st.B[src * NF + c] - m * vk_s;
            }
        }
    }
    return v;
}

// level < 0: blocks of level 0 from J and their inverses; else one PCR level (stride s = 2^level)
// i: the cell of this 32-lane group, e: the lane's index in the group (the launch kernels: 8 cells per 256-thread workgroup; the
// one-workgroup-per-instance integrator of marl_radau_wg.h: 32 cells per pass of its 1024 threads).  Every thread of the
// group's wave must call it together.
template <class T>
__device__ void pcr_factor_group(const double* __restrict__ J, int64_t N, int level, T mu, const PcrSystem<T>& S, PcrStage<T>& st, double jscale, int64_t i,
                                 int e)
{
    const bool act = e < 25 && i < N;
    const int r = act ? e / NF : 0, c = act ? e % NF : 0;
    const int64_t ic = (i < N) ? i : N - 1;   // idle groups shadow the last cell (uniform barriers, no stores)
    T d;
    if (level < 0) {
        // the matrix is  mu I - jscale J  (Radau: mu = MU / h, jscale = 1 - the product with 1.0 is exact; BDF: I - c J)
        d = lift(-(jscale * J[(ic * 3 + 1) * 25 + (act ? e : 0)]), mu);
        if (r == c) d = d + mu;
        if (act) {
            S.D[0][i * 25 + e] = d;
            S.L[0][i * 25 + e] = lift(i > 0 ? -(jscale * J[(i * 3 + 0) * 25 + e]) : 0.0, mu);
            S.U[0][i * 25 + e] = lift(i < N - 1 ? -(jscale * J[(i * 3 + 2) * 25 + e]) : 0.0, mu);
        }
        const T inv = gj_inverse_elem<T>(d, e, r, c, act, st);
        if (act) S.Dinv[0][i * 25 + e] = inv;
        return;
    }
    // (selects, not S.x[level & 1]: a runtime index into the argument struct would put the struct into scratch)
    const bool odd = level & 1;
    const T *Lc = odd ? S.L[1] : S.L[0], *Dc = odd ? S.D[1] : S.D[0], *Uc = odd ? S.U[1] : S.U[0], *Ic = odd ? S.Dinv[1] : S.Dinv[0];
    T *Ln = odd ? S.L[0] : S.L[1], *Dn_ = odd ? S.D[0] : S.D[1], *Un = odd ? S.U[0] : S.U[1], *In = odd ? S.Dinv[0] : S.Dinv[1];
    const int64_t s = (int64_t)1 << level;
    const int ee = act ? e : 0;
    d = Dc[ic * 25 + ee];
    const T zero = lift(0.0, mu);
    // lower side: alpha = -L D_{i-s}^-1;  D += alpha U_{i-s};  L' = alpha L_{i-s}
    const bool lo = ic - s >= 0;
    T al = zero, ln = zero;
    group_sync();
    if (act && lo) { st.A[e] = Lc[i * 25 + e]; st.B[e] = Ic[(i - s) * 25 + e]; }
    group_sync();
    if (act && lo) al = lift(-1.0, mu) * mm_elem<T>(st.A, st.B, r, c);
    group_sync();
    if (act && lo) { st.A[e] = al; st.B[e] = Uc[(i - s) * 25 + e]; st.C[e] = Lc[(i - s) * 25 + e]; }
    group_sync();
    if (act && lo) { d = d + mm_elem<T>(st.A, st.B, r, c); ln = mm_elem<T>(st.A, st.C, r, c); }
    // upper side: gamma = -U D_{i+s}^-1;  D += gamma L_{i+s};  U' = gamma U_{i+s}
    const bool hi = ic + s < N;
    T ga = zero, un = zero;
    group_sync();
    if (act && hi) { st.A[e] = Uc[i * 25 + e]; st.B[e] = Ic[(i + s) * 25 + e]; }
    group_sync();
    if (act && hi) ga = lift(-1.0, mu) * mm_elem<T>(st.A, st.B, r, c);
    group_sync();
    if (act && hi) { st.A[e] = ga; st.B[e] = Lc[(i + s) * 25 + e]; st.C[e] = Uc[(i + s) * 25 + e]; }
    group_sync();
    if (act && hi) { d = d + mm_elem<T>(st.A, st.B, r, c); un = mm_elem<T>(st.A, st.C, r, c); }
    if (act) {
        S.alpha[((int64_t)level * N + i) * 25 + e] = al;
        S.gamma[((int64_t)level * N + i) * 25 + e] = ga;
        Ln[i * 25 + e] = ln;
        Un[i * 25 + e] = un;
        Dn_[i * 25 + e] = d;
    }
    const T inv = gj_inverse_elem<T>(d, e, r, c, act, st);
    if (act) In[i * 25 + e] = inv;
}

// blockIdx.y: 0 real system, 1 complex system.  level < 0: initialise from J.
__global__ void __launch_bounds__(256) pcr_factor_kernel(const double* __restrict__ J, int64_t N, int level, double mu_r, cplx mu_c, PcrSystem<double> Sr,
                                                         PcrSystem<cplx> Sc, ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr}, double jscale = 1.0)
{
    if (z_masked_out(B)) return;
    if (B.act) { const RadauCtl* c = ctl_of(B); mu_r = c->mu_r; mu_c = cplx{c->mu_c_re, c->mu_c_im}; }
    J = z_shift(J, B); Sr = z_shift_system(Sr, B); Sc = z_shift_system(Sc, B);
    __shared__ PcrStage<cplx> stage[PCR_CELLS_PER_BLOCK];   // (the real system uses the same bytes)
    const int g = threadIdx.x >> 5, e = threadIdx.x & 31;
    const int64_t i = (int64_t)blockIdx.x * PCR_CELLS_PER_BLOCK + g;
    if (blockIdx.y == 0) pcr_factor_group<double>(J, N, level, mu_r, Sr, *reinterpret_cast<PcrStage<double>*>(&stage[g]), jscale, i, e);
    else pcr_factor_group<cplx>(J, N, level, mu_c, Sc, stage[g], jscale, i, e);
}

template <class T>
__device__ __forceinline__ void pcr_solve_row(int64_t N, int64_t kk, int level, int nlevels, const PcrSystem<T>& S, const T* __restrict__ bin,
                                              T* __restrict__ bout)
{
    const int64_t i = kk / NF;
    const int r = (int)(kk % NF);
    if (level < nlevels) {
        const int64_t s = (int64_t)1 << level;
        T acc = bin[kk];
        if (i - s >= 0) {
            const T* al = S.alpha + ((int64_t)level * N + i) * 25 + r * NF;
#pragma unroll
            for (int k = 0; k < NF; k++) acc = madd(acc, al[k], bin[(i - s) * NF + k]);
        }
        if (i + s < N) {
            const T* ga = S.gamma + ((int64_t)level * N + i) * 25 + r * NF;
#pragma unroll
            for (int k = 0; k < NF; k++) acc = madd(acc, ga[k], bin[(i + s) * NF + k]);
        }
        bout[kk] = acc;
    } else {   // x_i = D_i^-1 b_i
        const T* di = ((nlevels & 1) ? S.Dinv[1] : S.Dinv[0]) + i * 25 + r * NF;
        T acc = mul1(di[0], bin[i * NF]);
#pragma unroll
        for (int k = 1; k < NF; k++) acc = madd(acc, di[k], bin[i * NF + k]);
        bout[kk] = acc;
    }
}

// one level of a solve (level == nlevels: the final D^-1 b); blockIdx.y + first: which system (0 real, 1 complex)
__global__ void __launch_bounds__(256) pcr_solve_kernel(int64_t N, int level, int nlevels, int first, PcrSystem<double> Sr, PcrSystem<cplx> Sc,
                                                        const double* __restrict__ bin_r, double* __restrict__ bout_r, const cplx* __restrict__ bin_c,
                                                        cplx* __restrict__ bout_c, ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    Sr = z_shift_system(Sr, B); Sc = z_shift_system(Sc, B);
    bin_r = z_shift(bin_r, B); bout_r = z_shift(bout_r, B); bin_c = z_shift(bin_c, B); bout_c = z_shift(bout_c, B);
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= NF * N) return;
    if (blockIdx.y + first == 0) pcr_solve_row<double>(N, kk, level, nlevels, Sr, bin_r, bout_r);
    else pcr_solve_row<cplx>(N, kk, level, nlevels, Sc, bin_c, bout_c);
}

// All levels of a solve in ONE launch for small systems (5 N <= PCR_FUSED_MAX unknowns - the N = 200 grids of the reference's own
// runs and of scenario sweeps): one workgroup per (instance, system) keeps the right-hand side in LDS and steps through the levels
// with a barrier between them - the same recurrences in the same order as the per-level kernel (bit-identical results), 1 launch
// instead of 1 + ceil(log2 N) per Newton iteration.
constexpr int PCR_FUSED_MAX = 2048;
constexpr int PCR_FUSED_THREADS = 1024;

template <class T>
__device__ __forceinline__ void pcr_solve_all(int64_t N, int nlevels, const PcrSystem<T>& S, const T* __restrict__ bin, T* __restrict__ bout, T* lds)
{
    const int n = (int)(NF * N);
    for (int k = threadIdx.x; k < n; k += PCR_FUSED_THREADS) lds[k] = bin[k];
    __syncthreads();
    int cur = 0;
    for (int level = 0; level < nlevels; level++) {
        const T* b = lds + cur * PCR_FUSED_MAX;
        T* o = lds + (cur ^ 1) * PCR_FUSED_MAX;
        for (int k = threadIdx.x; k < n; k += PCR_FUSED_THREADS) pcr_solve_row<T>(N, k, level, nlevels, S, b, o);
        __syncthreads();
        cur ^= 1;
    }
    for (int k = threadIdx.x; k < n; k += PCR_FUSED_THREADS) pcr_solve_row<T>(N, k, nlevels, nlevels, S, lds + cur * PCR_FUSED_MAX, bout);
}

// ---- block CYCLIC reduction in front of PCR (marl_radau_cr.h has the story and the factor kernels): storage and the row recurrences
// of a solve, here because the one-launch solve kernels below use them too ----------------------------------------------------
// One system (real or complex) over all levels.  Level l has n_l rows in position space; in every array its rows start at row off_l.
template <class T>
struct CrSystem {
    T *L, *D, *U;        // the level's blocks [row][25] (levels 0 .. k-1; level k is set 0 of the compact PcrSystem)
    T *Dinv, *P, *Q;     // rows eliminated at their level (even positions): D^-1, -D^-1 L, -D^-1 U
    T *alpha, *gamma;    // rows that stay (odd positions)
    T *b;                // right-hand sides, then solutions [row][5] of levels 1 .. k (level 0 is the caller's vector)
};
struct CrShape {
    int64_t n_cur, off_cur, n_next, off_next;
};

// the levels of a SMALL system (one-launch solves; k <= CR_WG_MAX_LEVELS): level l has n[l] rows from row off[l]; n[k] rows go to PCR
constexpr int CR_WG_MAX_LEVELS = 3;
struct CrPlan {
    int k;   // 0: no cyclic reduction (PCR over all rows)
    int64_t n[CR_WG_MAX_LEVELS + 1], off[CR_WG_MAX_LEVELS + 1];
};
template <class T>
__device__ __forceinline__ CrSystem<T> z_shift_cr(CrSystem<T> C, const ZBatch& B)
{
    C.L = z_shift(C.L, B); C.D = z_shift(C.D, B); C.U = z_shift(C.U, B); C.Dinv = z_shift(C.Dinv, B); C.P = z_shift(C.P, B); C.Q = z_shift(C.Q, B);
    C.alpha = z_shift(C.alpha, B); C.gamma = z_shift(C.gamma, B); C.b = z_shift(C.b, B);
    return C;
}

// one thread per unknown of level l + 1:  b'_q = b_p + alpha_p b_{p-1} + gamma_p b_{p+1},  p = 2 q + 1
template <class T>
__device__ __forceinline__ void cr_rhs_row(const CrSystem<T>& C, CrShape sh, int64_t kk, const T* __restrict__ bin, T* __restrict__ bout)
{
    const int64_t q = kk / NF;
    const int r = (int)(kk % NF);
    const int64_t p = 2 * q + 1;
    T acc = bin[p * NF + r];
    const T* al = C.alpha + (sh.off_cur + p) * 25 + r * NF;
#pragma unroll
    for (int k = 0; k < NF; k++) acc = madd(acc, al[k], bin[(p - 1) * NF + k]);
    if (p + 1 < sh.n_cur) {
        const T* ga = C.gamma + (sh.off_cur + p) * 25 + r * NF;
#pragma unroll
        for (int k = 0; k < NF; k++) acc = madd(acc, ga[k], bin[(p + 1) * NF + k]);
    }
    bout[kk] = acc;
}

// Back-substitution value of row p of level l (b: the level's right-hand sides, xn: the solution of level l + 1): odd positions take their
// value from xn, even ones  x_p = D_p^-1 b_p + P_p x_{p-1} + Q_p x_{p+1}.
template <class T>
__device__ __forceinline__ T cr_back_value(const CrSystem<T>& C, CrShape sh, int64_t p, int r, const T* __restrict__ b, const T* __restrict__ xn)
{
    const int64_t q = p >> 1;
    if (p & 1) return xn[q * NF + r];
    const T* di = C.Dinv + (sh.off_cur + p) * 25 + r * NF;
    T acc = mul1(di[0], b[p * NF]);
#pragma unroll
    for (int k = 1; k < NF; k++) acc = madd(acc, di[k], b[p * NF + k]);
    if (q >= 1) {
        const T* pr = C.P + (sh.off_cur + p) * 25 + r * NF;
#pragma unroll
        for (int k = 0; k < NF; k++) acc = madd(acc, pr[k], xn[(q - 1) * NF + k]);
    }
    if (q < sh.n_next) {
        const T* qr = C.Q + (sh.off_cur + p) * 25 + r * NF;
#pragma unroll
        for (int k = 0; k < NF; k++) acc = madd(acc, qr[k], xn[q * NF + k]);
    }
    return acc;
}


// A whole solve of a small system in ONE workgroup: right-hand sides down the k cyclic-reduction levels, PCR on the rows that are
// left, solutions back up - all vectors in LDS (level l at lds + 5 off[l]; the PCR ping-pong partner behind the last level:
// 5 (off[k] + 2 n[k]) <= 2 PCR_FUSED_MAX elements for N <= 409, k <= 3).  Why not PCR alone: the chain of levels in one workgroup is
// bound by the BYTES of factor rows that must pass through one compute unit's 64 B / clock (tools/lab_src/solve_chain_probe.hip:
// 1.1 us per level of 80 KB, 0.43 us without the loads, 0.04 us for the barrier) - PCR reads 8 levels x N rows x 400 B, two or
// three levels of cyclic reduction in front of it cut that 2.4 - 2.8 times.  pl.k == 0: pcr_solve_all.  Row arithmetic: cr_rhs_row,
// pcr_solve_row, cr_back_value - the launch kernels' (bit-identical to a level-by-level solve).  In place (bout == bin) is fine.
template <class T>
__device__ __forceinline__ void crpcr_solve_all(const CrPlan& pl, int64_t N, int nlevels, const CrSystem<T>& C, const PcrSystem<T>& S, const T* bin, T* bout, T* lds,
                                                bool staged = false)   // staged (pl.k > 0 only): the right-hand side is in lds[0 .. 5 N) already
{
    if (pl.k == 0) {
        pcr_solve_all<T>(N, nlevels, S, bin, bout, lds);
        return;
    }
    const int n = (int)(NF * N);
    if (!staged)
        for (int kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) lds[kk] = bin[kk];
    __syncthreads();
#pragma unroll
    for (int l = 0; l < CR_WG_MAX_LEVELS; l++) {
        if (l < pl.k) {
            const CrShape sh{pl.n[l], pl.off[l], pl.n[l + 1], pl.off[l + 1]};
            const int cnt = (int)(NF * sh.n_next);
            for (int kk = threadIdx.x; kk < cnt; kk += PCR_FUSED_THREADS) cr_rhs_row<T>(C, sh, kk, lds + sh.off_cur * NF, lds + sh.off_next * NF);
            __syncthreads();
        }
    }
    const int64_t M = pl.n[pl.k];
    const int m = (int)(NF * M);
    T* cur = lds + pl.off[pl.k] * NF;
    T* oth = cur + m;
    for (int level = 0; level <= nlevels; level++) {
        for (int kk = threadIdx.x; kk < m; kk += PCR_FUSED_THREADS) pcr_solve_row<T>(M, kk, level, nlevels, S, cur, oth);
        __syncthreads();
        T* sw = cur; cur = oth; oth = sw;
    }
    const T* xn = cur;   // the solution of the compact system
#pragma unroll
    for (int l = CR_WG_MAX_LEVELS - 1; l >= 0; l--) {
        if (l < pl.k) {
            const CrShape sh{pl.n[l], pl.off[l], pl.n[l + 1], pl.off[l + 1]};
            T* b = lds + sh.off_cur * NF;
            const int cnt = (int)(NF * sh.n_cur);
            T v0 = T{}, v1 = T{};   // (5 n_cur <= 2 x 1024: at most two rows' worth per thread; formed before any is stored - in place)
            const int k0 = threadIdx.x, k1 = threadIdx.x + PCR_FUSED_THREADS;
            if (k0 < cnt) v0 = cr_back_value<T>(C, sh, k0 / NF, k0 % NF, b, xn);
            if (k1 < cnt) v1 = cr_back_value<T>(C, sh, k1 / NF, k1 % NF, b, xn);
            __syncthreads();
            if (k0 < cnt) b[k0] = v0;
            if (k1 < cnt) b[k1] = v1;
            __syncthreads();
            xn = b;
        }
    }
    for (int kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) bout[kk] = lds[kk];
}

// blockIdx.y + first: which system (0 real, 1 complex); in place (bout == bin) is fine: the input is staged in LDS first
__global__ void __launch_bounds__(PCR_FUSED_THREADS) pcr_solve_fused_kernel(int64_t N, int nlevels, int first, PcrSystem<double> Sr, PcrSystem<cplx> Sc,
                                                                            const double* bin_r, double* bout_r, const cplx* bin_c, cplx* bout_c,
                                                                            ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr}, CrPlan pl = CrPlan{},
                                                                            CrSystem<double> Cr = CrSystem<double>{}, CrSystem<cplx> Cc = CrSystem<cplx>{})
{
    if (z_masked_out(B)) return;
    Sr = z_shift_system(Sr, B); Sc = z_shift_system(Sc, B);
    Cr = z_shift_cr(Cr, B); Cc = z_shift_cr(Cc, B);
    bin_r = z_shift(bin_r, B); bout_r = z_shift(bout_r, B); bin_c = z_shift(bin_c, B); bout_c = z_shift(bout_c, B);
    __shared__ cplx lds[2 * PCR_FUSED_MAX];   // (the real system uses half of the bytes)
    if (blockIdx.y + first == 0) crpcr_solve_all<double>(pl, N, nlevels, Cr, Sr, bin_r, bout_r, reinterpret_cast<double*>(lds));
    else crpcr_solve_all<cplx>(pl, N, nlevels, Cc, Sc, bin_c, bout_c, lds);
}

// ---- element-wise pieces of solve_collocation_system (radau.py:47-130) and _step_impl (:404-537) -------------------------
// scale = atol + |y| rtol;  Z = Z0, W = TI Z0, YS = y + Z
__global__ void __launch_bounds__(256) newton_begin_kernel(const double* __restrict__ y, const double* __restrict__ Z0, int64_t n, double rtol, double atol,
                                                           double* __restrict__ scale, double* __restrict__ Z, double* __restrict__ W, double* __restrict__ YS)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double yi = y[i], z0 = Z0[i], z1 = Z0[n + i], z2 = Z0[2 * n + i];
    scale[i] = atol + fabs(yi) * rtol;
    Z[i] = z0; Z[n + i] = z1; Z[2 * n + i] = z2;
    W[i] = (TI00 * z0 + TI01 * z1) + TI02 * z2;
    W[n + i] = (TI10 * z0 + TI11 * z1) + TI12 * z2;
    W[2 * n + i] = (TI20 * z0 + TI21 * z1) + TI22 * z2;
    YS[i] = yi + z0; YS[n + i] = yi + z1; YS[2 * n + i] = yi + z2;
}

// right-hand sides of the two linear systems, in the cell-major ordering of the block matrices; flags[0] |= non-finite F
__global__ void __launch_bounds__(256) newton_rhs_kernel(const double* __restrict__ F, const double* __restrict__ W, int64_t N, double M_real, cplx M_c,
                                                         double* __restrict__ rhs_r, cplx* __restrict__ rhs_c, int32_t* __restrict__ flags)
{
    const int64_t n = NF * N;
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= n) return;
    const int64_t i = to_field_major(kk, N);
    const double f0 = F[i], f1 = F[n + i], f2 = F[2 * n + i];
    if (!(isfinite(f0) && isfinite(f1) && isfinite(f2))) *flags = 1;   // (a plain store: racing stores of the same value; the word may live in host memory)
    rhs_r[kk] = ((f0 * TI00 + f1 * TI01) + f2 * TI02) - M_real * W[i];
    const cplx w = {W[n + i], W[2 * n + i]};
    const cplx fc = {(f0 * TI10 + f1 * TI11) + f2 * TI12, (f0 * TI20 + f1 * TI21) + f2 * TI22};
    rhs_c[kk] = fc - M_c * w;
}

// out[blockIdx.x] = this workgroup's share of  sum (dW / scale)^2  over the 3n entries (one workgroup: the sum itself; more:
// sum_partials_kernel adds them in index order); then W += dW, Z = T W, YS = y + Z (kept only if the host goes on: a `break` of
// the Newton loop discards W and Z anyway, radau.py:112-119)
__global__ void __launch_bounds__(1024) newton_update_kernel(const double* __restrict__ y, const double* __restrict__ rhs_r, const cplx* __restrict__ rhs_c,
                                                             const double* __restrict__ scale, int64_t N, double* __restrict__ W, double* __restrict__ Z,
                                                             double* __restrict__ YS, double* __restrict__ out)
{
    __shared__ double red[1024];
    const int64_t n = NF * N;
    double ss = 0;
    for (int64_t kk = (int64_t)blockIdx.x * 1024 + threadIdx.x; kk < n; kk += (int64_t)gridDim.x * 1024) {
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
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// One Newton iteration's linear algebra in ONE launch for small systems (5 N <= PCR_FUSED_MAX; round 3, as marl_bdf.h's newton_fused_kernel
// does for BDF): the right-hand sides of both systems (newton_rhs_kernel) staged straight into LDS, every cyclic-reduction level of the
// real and then of the complex system (pcr_solve_all), then W += dW, Z = T W, YS = y + Z and the norm (newton_update_kernel with one
// workgroup) - the same per-element operations, the same reduction tree: bit-identical to the three launches it replaces
// (tests/test_gpu_radau.py::test_radau_fused_newton_launch_is_bit_identical), two launches less per iteration.  The non-finite flag is
// written by the thread that writes the polled word, with a system-scope fence in between.
__global__ void __launch_bounds__(PCR_FUSED_THREADS) newton_fused_kernel(const double* __restrict__ y, const double* __restrict__ F, int64_t N, double M_real, cplx M_c,
                                                                         int nlevels, PcrSystem<double> Sr, PcrSystem<cplx> Sc, const double* __restrict__ scale,
                                                                         double* __restrict__ W, double* __restrict__ Z, double* __restrict__ YS,
                                                                         double* __restrict__ rhs_r, cplx* __restrict__ rhs_c, int32_t* __restrict__ flags,
                                                                         double* __restrict__ out, CrPlan pl = CrPlan{}, CrSystem<double> Cr = CrSystem<double>{},
                                                                         CrSystem<cplx> Cc = CrSystem<cplx>{})
{
    __shared__ cplx lds[2 * PCR_FUSED_MAX];   // (the real system uses half of the bytes)
    __shared__ double red[PCR_FUSED_THREADS];
    const int64_t n = NF * N;
    int bad = 0;
    for (int64_t kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // newton_rhs_kernel
        const int64_t i = to_field_major(kk, N);
        const double f0 = F[i], f1 = F[n + i], f2 = F[2 * n + i];
        bad |= !(isfinite(f0) && isfinite(f1) && isfinite(f2));
        rhs_r[kk] = ((f0 * TI00 + f1 * TI01) + f2 * TI02) - M_real * W[i];
        const cplx w = {W[n + i], W[2 * n + i]};
        const cplx fc = {(f0 * TI10 + f1 * TI11) + f2 * TI12, (f0 * TI20 + f1 * TI21) + f2 * TI22};
        rhs_c[kk] = fc - M_c * w;
    }
    const int any_bad = __syncthreads_or(bad);
    crpcr_solve_all<double>(pl, N, nlevels, Cr, Sr, rhs_r, rhs_r, reinterpret_cast<double*>(lds));
    __syncthreads();
    crpcr_solve_all<cplx>(pl, N, nlevels, Cc, Sc, rhs_c, rhs_c, lds);
    __syncthreads();
    double ss = 0;
    for (int64_t kk = threadIdx.x; kk < n; kk += PCR_FUSED_THREADS) {   // newton_update_kernel, one workgroup
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
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (any_bad) *flags = 1;
        __threadfence_system();
        out[0] = red[0];
    }
}

// out[0] = partial[0] + partial[1] + ... (index order: reproducible)
__global__ void sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out)
{
    double s = 0;
    for (int i = 0; i < n; i++) s += partial[i];
    out[0] = s;
}

// right-hand side of the error estimate: fvec + Z^T E / h  (radau.py:468-470, :476), cell-major; also y_new = y + Z[2]
__global__ void __launch_bounds__(256) error_rhs_kernel(const double* __restrict__ fvec, const double* __restrict__ Z, const double* __restrict__ y, int64_t N,
                                                        double E0, double E1, double E2, double h, double* __restrict__ rhs_r, double* __restrict__ ynew)
{
    const int64_t n = NF * N;
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= n) return;
    const int64_t i = to_field_major(kk, N);
    const double ZE = ((Z[i] * E0 + Z[n + i] * E1) + Z[2 * n + i] * E2) / h;
    rhs_r[kk] = fvec[i] + ZE;
    ynew[i] = y[i] + Z[2 * n + i];
}

// err (field-major) = solution; scale = atol + max(|y|, |y_new|) rtol; out[blockIdx.x] = partial sum (err / scale)^2; yerr = y + err
__global__ void __launch_bounds__(1024) error_norm_kernel(const double* __restrict__ rhs_r, const double* __restrict__ y, const double* __restrict__ ynew,
                                                          int64_t N, double rtol, double atol, double* __restrict__ err, double* __restrict__ yerr,
                                                          double* __restrict__ out)
{
    __shared__ double red[1024];
    const int64_t n = NF * N;
    double ss = 0;
    for (int64_t kk = (int64_t)blockIdx.x * 1024 + threadIdx.x; kk < n; kk += (int64_t)gridDim.x * 1024) {
        const int64_t i = to_field_major(kk, N);
        const double e = rhs_r[kk];
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
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// dense output coefficients Q = Z^T P (radau.py:539-541), stored [i][3]
struct P33 { double p[3][3]; };
__global__ void __launch_bounds__(256) dense_q_kernel(const double* __restrict__ Z, int64_t n, P33 P, double* __restrict__ Q)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double z0 = Z[i], z1 = Z[n + i], z2 = Z[2 * n + i];
#pragma unroll
    for (int m = 0; m < 3; m++) Q[3 * i + m] = (z0 * P.p[0][m] + z1 * P.p[1][m]) + z2 * P.p[2][m];
}

// RadauDenseOutput._call_impl (radau.py:557-572) at up to three times x[s] = (t_s - t_old) / h_old:
//   out[s][i] = (Q[i] . [x, x^2, x^3] + y_old[i]) - sub[i]      (sub = NULL: nothing subtracted)
struct X3 { double x[3]; };
__global__ void __launch_bounds__(256) dense_eval_kernel(const double* __restrict__ Q, const double* __restrict__ yold, const double* __restrict__ sub,
                                                         int64_t n, X3 X, int ns, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2], yo = yold[i];
    const double sb = sub ? sub[i] : 0.0;
    for (int s = 0; s < ns; s++) {
        const double p1 = X.x[s], p2 = p1 * p1, p3 = p2 * p1;
        double v = ((q0 * p1 + q1 * p2) + q2 * p3) + yo;
        if (sub) v = v - sb;
        out[(int64_t)s * n + i] = v;
    }
}

}  // namespace radau
}  // namespace marl
