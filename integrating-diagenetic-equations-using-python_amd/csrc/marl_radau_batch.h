// marl_radau_batch.h - a SWEEP of implicit Radau integrations: many instances (own parameters, own step-size history, own Newton
// convergence) advanced together.  SURVEY 8(f) rank 3 asks for "batched block-tridiagonal solves on GPU"; at the reference's own
// grid (N = 200) one instance is launch-latency-bound (marl_integrate_radau: ~12 launches per Newton iteration), a sweep is what
// fills the chip: every launch below works on ALL instances that currently need that kind of work.
//
// Each instance runs scipy's Radau step logic (scipy/integrate/_ivp/radau.py:404-537, restated for one instance in marl_api.hip
// radau_run) as a resumable state machine ON THE DEVICE (radau_control_kernel, one lane per instance): it runs the scalar logic up to
// the next piece of data-parallel work, publishes that as an ACTION (A_* bits in RadauCtl::action) with its scalars (h, mu, dense-output
// abscissae), and resumes in the next cycle with the kernels' results (a sum of squares, a non-finite flag, the monitors record).
// The host repeats one fixed cycle of launches - control, then every kind of work, each masked to the instances that asked for it
// (ZBatch) - and reads one counter of still-running instances every few cycles.  Instances desynchronise freely.
#pragma once
#include "marl_radau.h"

namespace marl {
namespace radau {

enum : int32_t { PC_INIT = 0, PC_GOT_F0, PC_STEP, PC_ATTEMPT, PC_NEWTON_START, PC_NEWTON_LU_DONE, PC_NEWTON_ITER, PC_ERR_DONE, PC_ACCEPTED, PC_ACCEPTED_JAC_DONE, PC_DONE,
                 PC_EVENTS, PC_BRENT };
constexpr int NEWTON_MAXITER = 6;

__device__ __forceinline__ double predict_factor(double h_abs, double h_abs_old, double error_norm, double error_norm_old)   // radau.py:133-173
{
    double multiplier;
    if (error_norm_old < 0 || h_abs_old < 0 || error_norm == 0) multiplier = 1;
    else multiplier = h_abs / h_abs_old * pow(error_norm_old / error_norm, 0.25);
    return (multiplier < 1 ? multiplier : 1) * pow(error_norm, -0.25);
}

// One lane per instance: resume the step logic, run it to the next action.  rec: [B][8] monitors records of the instances' y.
// Work lists: the controller appends each instance to the list of every kernel group that serves its action; counts[L_*] are read by
// the host, which sizes the launches by them (masked-out workgroups are not free: at 512 instances a cycle that launched every kernel
// over every instance spent 2.4 ms dispatching ~320 000 workgroups that returned at once).
enum : int { L_RHS1 = 0, L_ACCEPT, L_JAC, L_LU, L_NEWTON, L_ERR, L_RUNNING, L_DENSE, L_COUNT };

// One instance's step logic from where it stopped to its next action (c.action, c.pc).  g_now: the seven monitors of the instance's y.
// One pass of Brent's method (scipy.optimize.brentq as solve_event_equation calls it: xtol = rtol = 4 eps, at most 100 iterations)
// from the top of its loop to the next function evaluation.  Returns true when the root is final (c.xcur); false: evaluate at c.xcur.
__device__ __forceinline__ bool brent_advance(RadauCtl& c)
{
    const double xtol = 4 * EPS, rtol = xtol;
    if (c.fpre != 0 && c.fcur != 0 && ((c.fpre < 0) != (c.fcur < 0))) { c.xblk = c.xpre; c.fblk = c.fpre; c.spre = c.scur = c.xcur - c.xpre; }
    if (fabs(c.fblk) < fabs(c.fcur)) { c.xpre = c.xcur; c.xcur = c.xblk; c.xblk = c.xpre; c.fpre = c.fcur; c.fcur = c.fblk; c.fblk = c.fpre; }
    const double delta = (xtol + rtol * fabs(c.xcur)) / 2, sbis = (c.xblk - c.xcur) / 2;
    if (c.fcur == 0 || fabs(sbis) < delta) return true;
    if (fabs(c.spre) > delta && fabs(c.fcur) < fabs(c.fpre)) {
        double stry;
        if (c.xpre == c.xblk) stry = -c.fcur * (c.xcur - c.xpre) / (c.fcur - c.fpre);
        else {
            const double dpre = (c.fpre - c.fcur) / (c.xpre - c.xcur), dblk = (c.fblk - c.fcur) / (c.xblk - c.xcur);
            stry = -c.fcur * (c.fblk * dblk - c.fpre * dpre) / (dblk * dpre * (c.fblk - c.fpre));
        }
        const double lim = fmin(fabs(c.spre), 3 * fabs(sbis) - delta);
        if (2 * fabs(stry) < lim) { c.spre = c.scur; c.scur = stry; }
        else { c.spre = sbis; c.scur = sbis; }
    } else { c.spre = sbis; c.scur = sbis; }
    c.xpre = c.xcur; c.fpre = c.fcur;
    if (fabs(c.scur) > delta) c.xcur += c.scur; else c.xcur += (sbis > 0 ? delta : -delta);
    return false;
}

// g_dense: the seven monitors of the dense-output state the last A_DENSE action evaluated (event root finding); t_events: this
// instance's root times [7][max_events] (NULL: sign changes are only counted).
__device__ __forceinline__ void radau_control_step(RadauCtl& c, const double (&g_now)[7], int64_t n, const double* g_dense = nullptr, double* t_events = nullptr)
{
    const double S6 = sqrt(6.0);
    const double C3[3] = {(4 - S6) / 10, (4 + S6) / 10, 1};
    const double MU_REAL = 3 + pow(3.0, 2.0 / 3) - pow(3.0, 1.0 / 3);
    const double MU_C_RE = 3 + 0.5 * (pow(3.0, 1.0 / 3) - pow(3.0, 2.0 / 3)), MU_C_IM = -0.5 * (pow(3.0, 5.0 / 6) + pow(3.0, 7.0 / 6));
    constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10;
    c.action = 0;
    int pc = c.pc;
    // a small interpreter: `continue` = go on with the new pc in this cycle, `break` out of the loop = yield
    for (int guard = 0; guard < 64; guard++) {
        if (pc == PC_INIT) {                     // Radau.__init__: f = fun(t0, y0)
            c.nfev = 1;
            c.action = A_RHS_Y; pc = PC_GOT_F0;
            break;
        }
        if (pc == PC_GOT_F0) {                   // jac_wrapped(t0, y0, f); g = events(t0, y0)
            for (int e = 0; e < 7; e++) c.g[e] = g_now[e];
            c.njev = 1;
            c.current_jac = 1;
            c.action = A_JAC; pc = PC_STEP;
            break;
        }
        if (pc == PC_STEP) {                     // top of _step_impl
            c.have_factor = 1;
            if (c.t == c.t_bound) { c.status = 0; pc = PC_DONE; break; }
            c.min_step = 10 * fabs(nextafter(c.t, __builtin_inf()) - c.t);
            c.h_abs = c.S_h_abs; c.h_abs_o = c.S_h_abs_old; c.err_o = c.S_err_old;
            if (c.S_h_abs < c.min_step) { c.h_abs = c.min_step; c.h_abs_o = -1; c.err_o = -1; }
            c.rejected = 0;
            pc = PC_ATTEMPT;
            continue;
        }
        if (pc == PC_ATTEMPT) {                  // top of `while not step_accepted`
            if (c.h_abs < c.min_step) { c.status = -1; pc = PC_DONE; break; }
            if (c.max_attempts > 0 && c.attempts >= c.max_attempts) { c.status = 2; pc = PC_DONE; break; }
            c.attempts++;
            c.h = c.h_abs;
            c.t_new = c.t + c.h;
            if (c.t_new - c.t_bound > 0) c.t_new = c.t_bound;
            c.h = c.t_new - c.t;
            c.h_abs = fabs(c.h);
            if (c.have_sol)
                for (int s = 0; s < 3; s++) c.x3[s] = ((c.t + c.h * C3[s]) - c.sol_t_old) / c.sol_h;
            pc = PC_NEWTON_START;
            continue;
        }
        if (pc == PC_NEWTON_START) {             // top of `while not converged`
            if (!c.have_lu) {
                c.mu_r = MU_REAL / c.h; c.mu_c_re = MU_C_RE / c.h; c.mu_c_im = MU_C_IM / c.h;
                c.nlu += 2;
                c.have_lu = 1;
                c.action = A_LU; pc = PC_NEWTON_LU_DONE;
                break;
            }
            pc = PC_NEWTON_LU_DONE;
            continue;
        }
        if (pc == PC_NEWTON_LU_DONE) {           // solve_collocation_system: first iteration
            c.mu_r = MU_REAL / c.h; c.mu_c_re = MU_C_RE / c.h; c.mu_c_im = MU_C_IM / c.h;   // M_real, M_complex
            c.k = 0; c.dW_norm_old = -1; c.rate = -1;
            c.newton_begin = 1;
            c.nfev += 3;
            c.action = A_NEWTON; pc = PC_NEWTON_ITER;
            break;
        }
        if (pc == PC_NEWTON_ITER) {              // the result of iteration c.k
            bool failed = false, converged = false;
            if (c.nonfinite) failed = true;
            else {
                const double dW_norm = sqrt(c.sumsq) / sqrt((double)(3 * n));
                if (c.dW_norm_old >= 0) c.rate = dW_norm / c.dW_norm_old;
                if (c.rate >= 0 && (c.rate >= 1 || pow_small_int(c.rate, NEWTON_MAXITER - c.k) / (1 - c.rate) * dW_norm > c.newton_tol)) failed = true;
                else if (dW_norm == 0 || (c.rate >= 0 && c.rate / (1 - c.rate) * dW_norm < c.newton_tol)) converged = true;
                else c.dW_norm_old = dW_norm;
            }
            c.nonfinite = 0;
            if (converged) {
                c.n_iter = c.k + 1;
                c.err_second = 0;
                c.action = A_ERR; pc = PC_ERR_DONE;
                break;
            }
            if (!failed && c.k + 1 < NEWTON_MAXITER) {
                c.k++;
                c.newton_begin = 0;
                c.nfev += 3;
                c.action = A_NEWTON; pc = PC_NEWTON_ITER;
                break;
            }
            // not converged (python: k + 1 with k the last loop value)
            c.n_iter = c.k + 1;
            if (c.current_jac) {                 // the step failed: halve
                c.h_abs *= 0.5;
                c.have_lu = 0;
                c.n_rej++;
                pc = PC_ATTEMPT;
                continue;
            }
            c.njev++;                            // J = self.jac(t, y, f); retry with a fresh Jacobian
            c.current_jac = 1;
            c.have_lu = 0;
            c.action = A_JAC; pc = PC_NEWTON_START;
            break;
        }
        if (pc == PC_ERR_DONE) {
            c.error_norm = sqrt(c.sumsq) / sqrt((double)n);
            c.safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + c.n_iter);
            if (c.rejected && c.error_norm > 1 && !c.err_second) {   // error = solve_lu(LU_real, fun(t, y + error) + ZE)
                c.err_second = 1;
                c.nfev++;
                c.action = A_ERR2; pc = PC_ERR_DONE;
                break;
            }
            c.err_second = 0;
            if (c.error_norm > 1) {
                const double sf = c.safety * predict_factor(c.h_abs, c.h_abs_o, c.error_norm, c.err_o);
                c.h_abs *= (sf > MIN_FACTOR) ? sf : MIN_FACTOR;
                c.have_lu = 0;
                c.rejected = 1;
                c.n_rej++;
                pc = PC_ATTEMPT;
                continue;
            }
            c.recompute_jac = (c.n_iter > 2 && c.rate > 1e-3) ? 1 : 0;
            double factor = predict_factor(c.h_abs, c.h_abs_o, c.error_norm, c.err_o);
            { const double sf = c.safety * factor; factor = (sf < MAX_FACTOR) ? sf : MAX_FACTOR; }
            if (!c.recompute_jac && factor < 1.2) factor = 1;
            else c.have_lu = 0;
            c.factor = factor;
            c.nfev++;                            // f_new = self.fun(t_new, y_new)
            c.action = A_ACCEPT; pc = PC_ACCEPTED;   // kernels: f_new, Q = Z^T P, y_old <- y, y <- y_new, f <- f_new, monitors(y)
            break;
        }
        if (pc == PC_ACCEPTED) {
            if (c.recompute_jac) {
                c.njev++;
                c.current_jac = 1;
                c.action = A_JAC; pc = PC_ACCEPTED_JAC_DONE;   // at (y_new, f_new), which are (y, f) now
                break;
            }
            c.current_jac = 0;
            pc = PC_ACCEPTED_JAC_DONE;
            continue;
        }
        if (pc == PC_ACCEPTED_JAC_DONE) {
            c.S_h_abs_old = c.S_h_abs;           // radau.py:512: the size proposed for this step, not the one used
            c.S_err_old = c.error_norm;
            c.S_h_abs = c.h_abs * c.factor;
            c.n_acc++;
            c.sol_t_old = c.t;
            c.t = c.t_new;
            c.sol_h = c.t - c.sol_t_old;
            c.have_sol = 1;
            c.ev_pending = 0;
            for (int e = 0; e < 7; e++) {        // non-terminal events, both directions (ivp.py:131-156)
                const bool up = c.g[e] <= 0 && g_now[e] >= 0, down = c.g[e] >= 0 && g_now[e] <= 0;
                if (up || down) {
                    if (c.locate_events && t_events && c.n_events[e] < c.max_events) c.ev_pending |= 1 << e;   // located below; the count moves with the root
                    else c.n_events[e]++;
                }
                c.g[e] = g_now[e];
            }
            pc = PC_EVENTS;
            continue;
        }
        if (pc == PC_EVENTS) {                   // the next monitor whose root in (sol_t_old, t] is still to be located (ivp.py:686-694)
            if (c.ev_pending == 0) {
                if (c.t - c.t_bound >= 0) { c.status = 0; pc = PC_DONE; break; }
                pc = PC_STEP;
                continue;
            }
            int e = 0;
            while (!((c.ev_pending >> e) & 1)) e++;
            c.br_e = e; c.br_phase = 0; c.br_iter = 0;
            c.br_a = c.sol_t_old; c.br_b = c.t;
            c.dense_x = 0.0;                     // f(a): the dense output at t_old
            c.action = A_DENSE; pc = PC_BRENT;
            break;
        }
        if (pc == PC_BRENT) {                    // the value of monitor br_e at the abscissa asked for
            const double val = g_dense[c.br_e];
            bool done = false;
            double root = 0;
            if (c.br_phase == 0) {
                c.br_fa = val;
                c.br_phase = 1;
                c.dense_x = 1.0;                 // f(b): the dense output at t
                c.action = A_DENSE;
                break;
            }
            if (c.br_phase == 1) {
                const double fa = c.br_fa, fb = val;
                if (fa == 0) { done = true; root = c.br_a; }
                else if (fb == 0) { done = true; root = c.br_b; }
                else {
                    c.xpre = c.br_a; c.xcur = c.br_b; c.fpre = fa; c.fcur = fb; c.xblk = 0; c.fblk = 0; c.spre = 0; c.scur = 0;
                    c.br_phase = 2;
                }
            } else {
                c.fcur = val;
                c.br_iter++;
                if (c.br_iter >= 100) { done = true; root = c.xcur; }
            }
            if (!done) {
                if (brent_advance(c)) { done = true; root = c.xcur; }
                else {
                    c.dense_x = (c.xcur - c.sol_t_old) / c.sol_h;
                    c.action = A_DENSE;
                    break;
                }
            }
            t_events[(int64_t)c.br_e * c.max_events + c.n_events[c.br_e]] = root;
            c.n_events[c.br_e]++;
            c.ev_pending &= ~(1 << c.br_e);
            pc = PC_EVENTS;
            continue;
        }
        break;
    }
    c.pc = pc;
}

__global__ void __launch_bounds__(64) radau_control_kernel(RadauCtl* __restrict__ ctls, const double* __restrict__ rec, int64_t B, int64_t n,
                                                           int32_t* __restrict__ counts, int32_t* __restrict__ lists,
                                                           const double* __restrict__ rec_dense = nullptr, double* __restrict__ t_events = nullptr)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    RadauCtl c = ctls[b];
    if (c.pc == PC_DONE) return;
    const double* r = rec + b * 8;
    const double g_now[7] = {r[1], r[2], r[3], r[5] - 1.0, r[6] - 1.0, r[4], r[7]};   // record_to_events (marl_api.hip)
    double g_dense[7] = {0, 0, 0, 0, 0, 0, 0};
    if (rec_dense) {
        const double* d = rec_dense + b * 8;
        const double gd[7] = {d[1], d[2], d[3], d[5] - 1.0, d[6] - 1.0, d[4], d[7]};
        for (int e = 0; e < 7; e++) g_dense[e] = gd[e];
    }
    radau_control_step(c, g_now, n, g_dense, t_events ? t_events + b * 7 * c.max_events : nullptr);
    const int pc = c.pc;
    ctls[b] = c;
    if (pc != PC_DONE) atomicAdd(&counts[L_RUNNING], 1);
    auto push = [&](int which) { lists[(int64_t)which * B + atomicAdd(&counts[which], 1)] = (int32_t)b; };
    if (c.action & (A_RHS_Y | A_ERR2 | A_ACCEPT)) push(L_RHS1);
    if (c.action & A_ACCEPT) push(L_ACCEPT);
    if (c.action & A_JAC) push(L_JAC);
    if (c.action & A_LU) push(L_LU);
    if (c.action & A_NEWTON) push(L_NEWTON);
    if (c.action & (A_ERR | A_ERR2)) push(L_ERR);
    if (c.action & A_DENSE) push(L_DENSE);
}

// A_DENSE: the dense output of the step just accepted at x = ctl.dense_x (RadauDenseOutput, radau.py:557-572; dense_eval_kernel) -> out
__global__ void __launch_bounds__(256) dense_eval_batch_kernel(const double* __restrict__ Q, const double* __restrict__ yold, int64_t n, double* __restrict__ out,
                                                               ZBatch B)
{
    if (z_masked_out(B)) return;
    const RadauCtl* c = ctl_of(B);
    Q = z_shift(Q, B); yold = z_shift(yold, B); out = z_shift(out, B);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2], yo = yold[i];
    const double p1 = c->dense_x, p2 = p1 * p1, p3 = p2 * p1;
    out[i] = dot3(q0, p1, q1, p2, q2, p3) + yo;
}

// ---- element-wise kernels of the batch (blockIdx.z = instance; scalars from the instance's controller) ---------------------

// a state -> its derivative, picked by the action: A_RHS_Y y -> f;  A_ERR2 y + err -> tmp;  A_ACCEPT y_new -> f_new.
template <int LAYOUT, bool VD>
__global__ void __launch_bounds__(256) rhs_pick_kernel(const double* y, double* f, const double* yerr, double* tmp, const double* ynew, double* fnew,
                                                       const DevConsts* __restrict__ consts, Slab S, ZBatch B)
{
    if (z_masked_out(B)) return;
    const int32_t a = ctl_of(B)->action;
    const double* src = z_shift((a & A_RHS_Y) ? y : ((a & A_ERR2) ? yerr : ynew), B);
    double* dst = z_shift((a & A_RHS_Y) ? f : ((a & A_ERR2) ? tmp : fnew), B);
    __shared__ double tabs[TABLE_DOUBLES];
    const Tables T = load_tables(tabs, 256);
    const DevConsts& C = consts[z_inst(B)];
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l < S.out_lo || l >= S.out_hi) return;
    const int64_t g = l + S.goff;
    double uc[NF], um[NF], up[NF], r[NF];
    PointAux aux;
#pragma unroll
    for (int fi = 0; fi < NF; fi++) {
        uc[fi] = src[at<LAYOUT>(fi, l, S.ld)];
        um[fi] = (g > 0) ? src[at<LAYOUT>(fi, l - 1, S.ld)] : ghost_lower(C.bc[fi], uc[fi]);
    }
#pragma unroll
    for (int fi = 0; fi < NF; fi++) up[fi] = (g < C.N - 1) ? src[at<LAYOUT>(fi, l + 1, S.ld)] : ghost_upper(fi, uc[fi], um[fi]);
    const HotConsts K = load_hot(&C);
    PointCache<0> pc;
    bool live = false;
    rhs_point<TR_PLAIN, 0, VD>(uc, um, up, g >= C.mask_lo && g < C.mask_hi, K, &C, T, r, aux, pc, live);
#pragma unroll
    for (int fi = 0; fi < NF; fi++) dst[at<LAYOUT>(fi, l, S.ld)] = r[fi];
}

// A_ACCEPT, after f_new: Q = Z^T P (radau.py:539-541);  y_old <- y;  y <- y_new;  f <- f_new
__global__ void __launch_bounds__(256) accept_kernel(const double* __restrict__ Z, double* __restrict__ Q, double* __restrict__ y, double* __restrict__ yold,
                                                     const double* __restrict__ ynew, double* __restrict__ f, const double* __restrict__ fnew, int64_t n,
                                                     P33 P, ZBatch B)
{
    if (z_masked_out(B)) return;
    Z = z_shift(Z, B); Q = z_shift(Q, B); y = z_shift(y, B); yold = z_shift(yold, B); ynew = z_shift(ynew, B); f = z_shift(f, B); fnew = z_shift(fnew, B);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double z0 = Z[i], z1 = Z[n + i], z2 = Z[2 * n + i];
#pragma unroll
    for (int m = 0; m < 3; m++) Q[3 * i + m] = dot3(z0, P.p[0][m], z1, P.p[1][m], z2, P.p[2][m]);
    yold[i] = y[i];
    y[i] = ynew[i];
    f[i] = fnew[i];
}

// A_NEWTON with newton_begin: Z0 = sol(t + h C) - y (or 0), scale, Z = Z0, W = TI Z0, YS = y + Z
__global__ void __launch_bounds__(256) newton_begin_batch_kernel(const double* __restrict__ y, const double* __restrict__ Q, const double* __restrict__ yold,
                                                                 int64_t n, double* __restrict__ scale, double* __restrict__ Z, double* __restrict__ W,
                                                                 double* __restrict__ YS, ZBatch B)
{
    if (z_masked_out(B)) return;
    const RadauCtl* c = ctl_of(B);
    if (!c->newton_begin) return;
    y = z_shift(y, B); Q = z_shift(Q, B); yold = z_shift(yold, B); scale = z_shift(scale, B); Z = z_shift(Z, B); W = z_shift(W, B); YS = z_shift(YS, B);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double yi = y[i];
    double z[3] = {0.0, 0.0, 0.0};
    if (c->have_sol) {
        const double q0 = Q[3 * i], q1 = Q[3 * i + 1], q2 = Q[3 * i + 2], yo = yold[i];
#pragma unroll
        for (int s = 0; s < 3; s++) {
            const double p1 = c->x3[s], p2 = p1 * p1, p3 = p2 * p1;
            z[s] = (dot3(q0, p1, q1, p2, q2, p3) + yo) - yi;
        }
    }
    scale[i] = c->atol + fabs(yi) * c->rtol;
    Z[i] = z[0]; Z[n + i] = z[1]; Z[2 * n + i] = z[2];
    W[i] = dot3(TI00, z[0], TI01, z[1], TI02, z[2]);
    W[n + i] = dot3(TI10, z[0], TI11, z[1], TI12, z[2]);
    W[2 * n + i] = dot3(TI20, z[0], TI21, z[1], TI22, z[2]);
    YS[i] = yi + z[0]; YS[n + i] = yi + z[1]; YS[2 * n + i] = yi + z[2];
}

// A_NEWTON: right-hand sides of the two systems (cell-major) from the stage derivatives F
__global__ void __launch_bounds__(256) newton_rhs_batch_kernel(const double* __restrict__ F, const double* __restrict__ W, int64_t N, double* __restrict__ rhs_r,
                                                               cplx* __restrict__ rhs_c, RadauCtl* __restrict__ ctls, ZBatch B)
{
    if (z_masked_out(B)) return;
    const RadauCtl* c = ctl_of(B);
    F = z_shift(F, B); W = z_shift(W, B); rhs_r = z_shift(rhs_r, B); rhs_c = z_shift(rhs_c, B);
    const int64_t n = NF * N;
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= n) return;
    const int64_t i = to_field_major(kk, N);
    const double f0 = F[i], f1 = F[n + i], f2 = F[2 * n + i];
    if (!(isfinite(f0) && isfinite(f1) && isfinite(f2))) atomicOr(&ctls[z_inst(B)].nonfinite, 1);
    rhs_r[kk] = __builtin_fma(-c->mu_r, W[i], dot3(f0, TI00, f1, TI01, f2, TI02));
    const cplx w = {W[n + i], W[2 * n + i]};
    const cplx fc = {dot3(f0, TI10, f1, TI11, f2, TI12), dot3(f0, TI20, f1, TI21, f2, TI22)};
    rhs_c[kk] = fc - cmul_ref(cplx{c->mu_c_re, c->mu_c_im}, w);
}

// A_NEWTON: one workgroup per instance: sum (dW / scale)^2 -> ctl.sumsq; W += dW, Z = T W, YS = y + Z
__global__ void __launch_bounds__(1024) newton_update_batch_kernel(const double* __restrict__ y, const double* __restrict__ rhs_r, const cplx* __restrict__ rhs_c,
                                                                   const double* __restrict__ scale, int64_t N, double* __restrict__ W, double* __restrict__ Z,
                                                                   double* __restrict__ YS, RadauCtl* __restrict__ ctls, ZBatch B)
{
    if (z_masked_out(B)) return;
    y = z_shift(y, B); rhs_r = z_shift(rhs_r, B); rhs_c = z_shift(rhs_c, B); scale = z_shift(scale, B); W = z_shift(W, B); Z = z_shift(Z, B); YS = z_shift(YS, B);
    __shared__ double red[1024];
    const int64_t n = NF * N;
    double ss = 0;
    for (int64_t kk = threadIdx.x; kk < n; kk += 1024) {
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
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ctls[z_inst(B)].sumsq = red[0];
}

// A_ERR / A_ERR2: right-hand side of the error estimate (f, or f(y + err) in tmp) + Z^T E / h, cell-major; y_new = y + Z[2]
__global__ void __launch_bounds__(256) error_rhs_batch_kernel(const double* __restrict__ f, const double* __restrict__ tmp, const double* __restrict__ Z,
                                                              const double* __restrict__ y, int64_t N, double E0, double E1, double E2,
                                                              double* __restrict__ rhs_r, double* __restrict__ ynew, ZBatch B)
{
    if (z_masked_out(B)) return;
    const RadauCtl* c = ctl_of(B);
    const double* fvec = z_shift((c->action & A_ERR2) ? tmp : f, B);
    Z = z_shift(Z, B); y = z_shift(y, B); rhs_r = z_shift(rhs_r, B); ynew = z_shift(ynew, B);
    const int64_t n = NF * N;
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= n) return;
    const int64_t i = to_field_major(kk, N);
    const double ZE = dot3(Z[i], E0, Z[n + i], E1, Z[2 * n + i], E2) / c->h;
    rhs_r[kk] = fvec[i] + ZE;
    ynew[i] = y[i] + Z[2 * n + i];
}

// A_ERR / A_ERR2: one workgroup per instance: err, y + err, sum (err / scale)^2 -> ctl.sumsq
__global__ void __launch_bounds__(1024) error_norm_batch_kernel(const double* __restrict__ rhs_r, const double* __restrict__ y, const double* __restrict__ ynew,
                                                                int64_t N, double* __restrict__ err, double* __restrict__ yerr, RadauCtl* __restrict__ ctls,
                                                                ZBatch B)
{
    if (z_masked_out(B)) return;
    const RadauCtl* c = ctl_of(B);
    rhs_r = z_shift(rhs_r, B); y = z_shift(y, B); ynew = z_shift(ynew, B); err = z_shift(err, B); yerr = z_shift(yerr, B);
    __shared__ double red[1024];
    const int64_t n = NF * N;
    double ss = 0;
    for (int64_t kk = threadIdx.x; kk < n; kk += 1024) {
        const int64_t i = to_field_major(kk, N);
        const double e = rhs_r[kk];
        const double a = fabs(y[i]), b = fabs(ynew[i]);
        const double s = c->atol + ((a > b || a != a) ? a : b) * c->rtol;
        const double q = e / s;
        ss = __builtin_fma(q, q, ss);
        err[i] = e;
        yerr[i] = y[i] + e;
    }
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ctls[z_inst(B)].sumsq = red[0];
}

// The work-list lengths of a cycle for the host: into coherent host memory (the host polls `seq`; no copy, no stream synchronisation)
// ... and zeroes them for the next cycle's controllers (one launch less per cycle than a memset in front of the control kernel)
__global__ void publish_counts_kernel(int32_t* __restrict__ counts, int32_t* __restrict__ host_words, int32_t seq)
{
    for (int i = 0; i < L_COUNT; i++) { host_words[i] = counts[i]; counts[i] = 0; }
    __threadfence_system();
    *reinterpret_cast<volatile int32_t*>(host_words + L_COUNT) = seq;
}

}  // namespace radau
}  // namespace marl
