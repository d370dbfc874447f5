// marl_radau_cr.h - block CYCLIC REDUCTION in front of the parallel cyclic reduction (marl_radau.h), for the large grids of single
// implicit runs (Radau and BDF; scipy's `lu` / `solve_lu` of radau.py:404-537 and bdf.py:36-68 on the block-tridiagonal mu I - J).
//
// Parallel cyclic reduction keeps EVERY equation through every level: ceil(log2 N) levels x N rows of block products, inverses and -
// what bounds it on a large grid - of alpha / gamma blocks that each Newton iteration reads again (N = 64 000: 16 levels x 64 000 rows x
// 1.2 KB = 1.2 GB per iteration; pcr_solve_kernel ran at 51 % of the HBM peak and was 52 % of the run, profiles/r03_implicit_kernels.json).
// Cyclic reduction keeps only every other equation: level l has n_l = floor(n_{l-1} / 2) rows (the odd POSITIONS of the level before),
// the same recurrences as a PCR level for the rows that stay,
//     alpha = -L_p D_{p-1}^-1,  gamma = -U_p D_{p+1}^-1,   D' = D_p + alpha U_{p-1} + gamma L_{p+1},  L' = alpha L_{p-1},  U' = gamma U_{p+1},
//     b' = b_p + alpha b_{p-1} + gamma b_{p+1},
// and the rows that leave (even positions) are recovered on the way back:  x_p = D_p^-1 b_p - (D_p^-1 L_p) x_{p-1} - (D_p^-1 U_p) x_{p+1}.
// Rows, products, inverses and factor bytes sum to 2 N instead of N log2 N.  After k levels the n_k rows that are left form a small
// block-tridiagonal system of their own, which the PCR kernels factorise and solve unchanged (k is chosen so that it fits the
// one-launch solve: 5 n_k <= PCR_FUSED_MAX).  The equations that stay see exactly PCR's arithmetic (same neighbours, same order);
// the eliminated ones differ from PCR's in rounding only.
//
// Single runs only (no ZBatch): sweeps integrate N = 200 grids, where a factorisation is launch-bound, not byte-bound.
#pragma once
#include "marl_radau.h"

namespace marl {
namespace radau {

constexpr int CR_MAX_LEVELS = 24;
constexpr int CR_ROWS_PER_BLOCK = 64;   // back-substitution: 64 rows x 5 unknowns = 320 threads, so that no row straddles two workgroups

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

// level 0 from the Jacobian: group g builds the blocks of cells 2g (eliminated at level 0: inverse and the two back-substitution blocks)
// and 2g + 1.  The matrix is mu I - jscale J as in pcr_factor_group.
template <class T>
__device__ void cr_init_group(const double* __restrict__ J, int64_t N, T mu, double jscale, const CrSystem<T>& C, PcrStage<T>& st, int64_t g, int e)
{
    const int64_t i0 = 2 * g;
    const bool act = e < 25 && i0 < N;
    const int r = act ? e / NF : 0, c = act ? e % NF : 0;
    const int ee = act ? e : 0;
    const int64_t ic = (i0 < N) ? i0 : 0;
    T d = lift(-(jscale * J[(ic * 3 + 1) * 25 + ee]), mu);
    if (r == c) d = d + mu;
    const T l0 = lift(ic > 0 ? -(jscale * J[(ic * 3 + 0) * 25 + ee]) : 0.0, mu);
    const T u0 = lift(ic < N - 1 ? -(jscale * J[(ic * 3 + 2) * 25 + ee]) : 0.0, mu);
    if (act) {
        C.D[i0 * 25 + e] = d;
        C.L[i0 * 25 + e] = l0;
        C.U[i0 * 25 + e] = u0;
        const int64_t i1 = i0 + 1;
        if (i1 < N) {
            T d1 = lift(-(jscale * J[(i1 * 3 + 1) * 25 + e]), mu);
            if (r == c) d1 = d1 + mu;
            C.D[i1 * 25 + e] = d1;
            C.L[i1 * 25 + e] = lift(-(jscale * J[(i1 * 3 + 0) * 25 + e]), mu);
            C.U[i1 * 25 + e] = lift(i1 < N - 1 ? -(jscale * J[(i1 * 3 + 2) * 25 + e]) : 0.0, mu);
        }
    }
    const T inv = gj_inverse_elem<T>(d, e, r, c, act, st);
    group_sync();
    if (act) { st.A[e] = inv; st.B[e] = l0; st.C[e] = u0; }
    group_sync();
    if (act) {
        C.Dinv[i0 * 25 + e] = inv;
        C.P[i0 * 25 + e] = lift(-1.0, mu) * mm_elem<T>(st.A, st.B, r, c);
        C.Q[i0 * 25 + e] = lift(-1.0, mu) * mm_elem<T>(st.A, st.C, r, c);
    }
}

// blockIdx.y: 0 real system, 1 complex system
__global__ void __launch_bounds__(256) cr_init_kernel(const double* __restrict__ J, int64_t N, double mu_r, cplx mu_c, double jscale, CrSystem<double> Cr,
                                                      CrSystem<cplx> Cc)
{
    __shared__ PcrStage<cplx> stage[PCR_CELLS_PER_BLOCK];   // (the real system uses the same bytes)
    const int g = threadIdx.x >> 5, e = threadIdx.x & 31;
    const int64_t i = (int64_t)blockIdx.x * PCR_CELLS_PER_BLOCK + g;
    if (blockIdx.y == 0) cr_init_group<double>(J, N, mu_r, jscale, Cr, *reinterpret_cast<PcrStage<double>*>(&stage[g]), i, e);
    else cr_init_group<cplx>(J, N, mu_c, jscale, Cc, stage[g], i, e);
}

// One reduction level: group g forms row q of level l + 1 from rows p - 1, p, p + 1 of level l (p = 2 q + 1) and, where row q is
// eliminated at level l + 1 (q even) or the level is the last one (the compact system needs every inverse), inverts its diagonal block.
// The first half of the groups takes the even q, the second half the odd ones, so that all groups of a wave do the same work.
// Ln / Dn / Un / In: where the new row's blocks and inverse go, indexed by q (level l + 1 of C, or set 0 of the compact system).
template <class T>
__device__ void cr_reduce_group(T mu, const CrSystem<T>& C, CrShape sh, bool last, T* __restrict__ Ln, T* __restrict__ Dn, T* __restrict__ Un,
                                T* __restrict__ In, PcrStage<T>& st, int64_t g, int e)
{
    const int64_t n_even = (sh.n_next + 1) / 2;
    const bool valid = g < sh.n_next;
    const int64_t q = !valid ? 0 : (last ? g : (g < n_even ? 2 * g : 2 * (g - n_even) + 1));
    const bool act = e < 25 && valid;
    const int r = act ? e / NF : 0, c = act ? e % NF : 0;
    const int ee = act ? e : 0;
    const int64_t p = 2 * q + 1;                 // (< n_cur: q < floor(n_cur / 2); with no valid row the loads below read row 1, which exists)
    const T* Lc = C.L + sh.off_cur * 25;
    const T* Dc = C.D + sh.off_cur * 25;
    const T* Uc = C.U + sh.off_cur * 25;
    const T* Ic = C.Dinv + sh.off_cur * 25;
    T d = Dc[p * 25 + ee];
    const T zero = lift(0.0, mu);
    // lower side (row p - 1 always exists)
    T al = zero, ln = zero;
    group_sync();
    if (act) { st.A[e] = Lc[p * 25 + e]; st.B[e] = Ic[(p - 1) * 25 + e]; }
    group_sync();
    if (act) al = lift(-1.0, mu) * mm_elem<T>(st.A, st.B, r, c);
    group_sync();
    if (act) { st.A[e] = al; st.B[e] = Uc[(p - 1) * 25 + e]; st.C[e] = Lc[(p - 1) * 25 + e]; }
    group_sync();
    if (act) { d = d + mm_elem<T>(st.A, st.B, r, c); ln = mm_elem<T>(st.A, st.C, r, c); }
    // upper side
    const bool hi = p + 1 < sh.n_cur;
    T ga = zero, un = zero;
    group_sync();
    if (act && hi) { st.A[e] = Uc[p * 25 + e]; st.B[e] = Ic[(p + 1) * 25 + e]; }
    group_sync();
    if (act && hi) ga = lift(-1.0, mu) * mm_elem<T>(st.A, st.B, r, c);
    group_sync();
    if (act && hi) { st.A[e] = ga; st.B[e] = Lc[(p + 1) * 25 + e]; st.C[e] = Uc[(p + 1) * 25 + e]; }
    group_sync();
    if (act && hi) { d = d + mm_elem<T>(st.A, st.B, r, c); un = mm_elem<T>(st.A, st.C, r, c); }
    if (act) {
        C.alpha[(sh.off_cur + p) * 25 + e] = al;
        C.gamma[(sh.off_cur + p) * 25 + e] = ga;
        Ln[q * 25 + e] = ln;
        Un[q * 25 + e] = un;
        Dn[q * 25 + e] = d;
    }
    const bool invert = act && (last || !(q & 1));
    if (__builtin_amdgcn_ballot_w64(invert) == 0) return;   // (wave-uniform: a wave of odd rows of an inner level is done)
    const T inv = gj_inverse_elem<T>(d, e, r, c, invert, st);
    if (invert) In[q * 25 + e] = inv;
    if (last) return;
    group_sync();
    if (invert) { st.A[e] = inv; st.B[e] = ln; st.C[e] = un; }
    group_sync();
    if (invert) {
        C.P[(sh.off_next + q) * 25 + e] = lift(-1.0, mu) * mm_elem<T>(st.A, st.B, r, c);
        C.Q[(sh.off_next + q) * 25 + e] = lift(-1.0, mu) * mm_elem<T>(st.A, st.C, r, c);
    }
}

__global__ void __launch_bounds__(256) cr_reduce_kernel(double mu_r, cplx mu_c, CrSystem<double> Cr, CrSystem<cplx> Cc, CrShape sh, int last,
                                                        PcrSystem<double> Sr, PcrSystem<cplx> Sc)
{
    __shared__ PcrStage<cplx> stage[PCR_CELLS_PER_BLOCK];
    const int g = threadIdx.x >> 5, e = threadIdx.x & 31;
    const int64_t i = (int64_t)blockIdx.x * PCR_CELLS_PER_BLOCK + g;
    if (blockIdx.y == 0) {
        double *Ln = last ? Sr.L[0] : Cr.L + sh.off_next * 25, *Dn = last ? Sr.D[0] : Cr.D + sh.off_next * 25;
        double *Un = last ? Sr.U[0] : Cr.U + sh.off_next * 25, *In = last ? Sr.Dinv[0] : Cr.Dinv + sh.off_next * 25;
        cr_reduce_group<double>(mu_r, Cr, sh, last != 0, Ln, Dn, Un, In, *reinterpret_cast<PcrStage<double>*>(&stage[g]), i, e);
    } else {
        cplx *Ln = last ? Sc.L[0] : Cc.L + sh.off_next * 25, *Dn = last ? Sc.D[0] : Cc.D + sh.off_next * 25;
        cplx *Un = last ? Sc.U[0] : Cc.U + sh.off_next * 25, *In = last ? Sc.Dinv[0] : Cc.Dinv + sh.off_next * 25;
        cr_reduce_group<cplx>(mu_c, Cc, sh, last != 0, Ln, Dn, Un, In, stage[g], i, e);
    }
}

// ---- a solve: right-hand sides down the levels, the compact system by PCR (marl_radau.h), solutions back up ------------------------
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

// blockIdx.y: 0 real system, 1 complex system.  b0_*: the level-0 vectors (the caller's right-hand sides); deeper levels live in C.b.
__global__ void __launch_bounds__(256) cr_rhs_kernel(CrSystem<double> Cr, CrSystem<cplx> Cc, CrShape sh, int level, const double* __restrict__ b0_r,
                                                     const cplx* __restrict__ b0_c)
{
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= NF * sh.n_next) return;
    if (blockIdx.y == 0) cr_rhs_row<double>(Cr, sh, kk, level == 0 ? b0_r : Cr.b + sh.off_cur * NF, Cr.b + sh.off_next * NF);
    else cr_rhs_row<cplx>(Cc, sh, kk, level == 0 ? b0_c : Cc.b + sh.off_cur * NF, Cc.b + sh.off_next * NF);
}

// Back-substitution of level l, IN PLACE in its right-hand side vector b (xn: the solution of level l + 1): odd positions take their
// value from xn, even ones  x_p = D_p^-1 b_p + P_p x_{p-1} + Q_p x_{p+1}.  Every thread forms its value, THEN (barrier) all store: a
// row's five threads read each other's b.
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

__global__ void __launch_bounds__(CR_ROWS_PER_BLOCK* NF) cr_back_kernel(CrSystem<double> Cr, CrSystem<cplx> Cc, CrShape sh, int level, double* b0_r, cplx* b0_c)
{
    const int64_t p = (int64_t)blockIdx.x * CR_ROWS_PER_BLOCK + threadIdx.x / NF;
    const int r = threadIdx.x % NF;
    const bool in = p < sh.n_cur;
    if (blockIdx.y == 0) {
        double* b = level == 0 ? b0_r : Cr.b + sh.off_cur * NF;
        const double v = in ? cr_back_value<double>(Cr, sh, p, r, b, Cr.b + sh.off_next * NF) : 0.0;
        __syncthreads();
        if (in) b[p * NF + r] = v;
    } else {
        cplx* b = level == 0 ? b0_c : Cc.b + sh.off_cur * NF;
        const cplx v = in ? cr_back_value<cplx>(Cc, sh, p, r, b, Cc.b + sh.off_next * NF) : cplx{0.0, 0.0};
        __syncthreads();
        if (in) b[p * NF + r] = v;
    }
}

}  // namespace radau
}  // namespace marl
