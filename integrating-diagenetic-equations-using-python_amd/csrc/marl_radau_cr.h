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

// (CrSystem / CrShape / CrPlan and the row recurrences of a solve - cr_rhs_row, cr_back_value - live in marl_radau.h, in front of the
//  one-launch solve kernels that use them.)

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
// blockIdx.z: instance of a sweep (ZBatch, as pcr_factor_kernel: per-instance mu from the instance's controller)
__global__ void __launch_bounds__(256) cr_init_kernel(const double* __restrict__ J, int64_t N, double mu_r, cplx mu_c, double jscale, CrSystem<double> Cr,
                                                      CrSystem<cplx> Cc, ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    if (B.act) { const RadauCtl* c = ctl_of(B); mu_r = c->mu_r; mu_c = cplx{c->mu_c_re, c->mu_c_im}; }
    J = z_shift(J, B); Cr = z_shift_cr(Cr, B); Cc = z_shift_cr(Cc, B);
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
                                                        PcrSystem<double> Sr, PcrSystem<cplx> Sc, ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    if (B.act) { const RadauCtl* c = ctl_of(B); mu_r = c->mu_r; mu_c = cplx{c->mu_c_re, c->mu_c_im}; }
    Cr = z_shift_cr(Cr, B); Cc = z_shift_cr(Cc, B); Sr = z_shift_system(Sr, B); Sc = z_shift_system(Sc, B);
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
// blockIdx.y: 0 real system, 1 complex system.  b0_*: the level-0 vectors (the caller's right-hand sides); deeper levels live in C.b.
__global__ void __launch_bounds__(256) cr_rhs_kernel(CrSystem<double> Cr, CrSystem<cplx> Cc, CrShape sh, int level, const double* __restrict__ b0_r,
                                                     const cplx* __restrict__ b0_c)
{
    const int64_t kk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (kk >= NF * sh.n_next) return;
    if (blockIdx.y == 0) cr_rhs_row<double>(Cr, sh, kk, level == 0 ? b0_r : Cr.b + sh.off_cur * NF, Cr.b + sh.off_next * NF);
    else cr_rhs_row<cplx>(Cc, sh, kk, level == 0 ? b0_c : Cc.b + sh.off_cur * NF, Cc.b + sh.off_next * NF);
}

// Back-substitution of level l, IN PLACE in its right-hand side vector b (cr_back_value, marl_radau.h): every thread forms its value,
// THEN (barrier) all store - a row's five threads read each other's b.
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

// ---- the launch-bound tail of a solve: several levels per launch -------------------------------------------------------------
// From the level where a launch no longer fills the machine (<= CR_TAIL_ROWS rows) every level kernel above costs its ~5.5 us of
// dispatch and drain, whatever it computes (N = 64 000: six of nine levels, 2 x 34 us of a 150 us solve).  The tail kernels walk J
// levels in one launch: a workgroup owns CR_TAIL_TOP rows of the deepest level and everything below them, keeps the vectors of the
// levels in between in LDS, and recomputes the few rows at its tile's edges that its neighbours compute as well (2^j - 1 rows at the
// j-th level from the top), so no workgroup waits for another.  Same recurrences in the same order as cr_rhs_row / cr_back_value:
// bit-identical to the level-by-level kernels.
constexpr int CR_TAIL_MAX_LEVELS = 6;
constexpr int CR_TAIL_TOP = 4;                                                                   // rows of the deepest level per workgroup
constexpr int64_t CR_TAIL_ROWS = 8192;                                                            // the tail starts at the first level this small
constexpr int CR_TAIL_TILE = (CR_TAIL_TOP + 2) << CR_TAIL_MAX_LEVELS;                             // rows of the tail's first level a workgroup may touch
constexpr int CR_TAIL_THREADS = 1024;
struct CrTail {
    int l0;                                      // levels l0 .. l0 + J - 1 are reduced / back-substituted here (J: template parameter)
    int64_t n[CR_TAIL_MAX_LEVELS + 1], off[CR_TAIL_MAX_LEVELS + 1];   // rows / first row of level l0 + j
};

// forward: b_{l0} -> b_{l0+1} ... b_{l0+J} (all written: the way back needs the right-hand sides of the rows each level eliminates)
template <class T, int J>
__device__ void cr_rhs_tail(const CrSystem<T>& C, const CrTail& t, const T* __restrict__ b0, T* lds)
{
    const int64_t nJ = t.n[J];
    const int64_t Q0 = (int64_t)blockIdx.x * CR_TAIL_TOP;
    const int64_t Q1 = (Q0 + CR_TAIL_TOP < nJ) ? Q0 + CR_TAIL_TOP : nJ;
    const bool last = Q1 == nJ;
    static_assert(J >= 1 && J <= CR_TAIL_MAX_LEVELS, "levels per tail launch");
    int64_t lo[J + 1], hi[J + 1];   // rows of level l0 + j this workgroup needs (inclusive)
    lo[J] = Q0; hi[J] = Q1 - 1;
#pragma unroll
    for (int j = J - 1; j >= 0; j--) {
        {
            lo[j] = 2 * lo[j + 1];
            const int64_t h = 2 * hi[j + 1] + 2;
            hi[j] = (last || h > t.n[j] - 1) ? t.n[j] - 1 : h;   // the last workgroup also takes the rows beyond 2^(J-j) n_J
        }
    }
    T* cur = lds;
    T* nxt = lds + CR_TAIL_TILE * NF;
    {
        const T* src = (t.l0 == 0 ? b0 : C.b + t.off[0] * NF) + lo[0] * NF;
        const int cnt = (int)(hi[0] - lo[0] + 1) * NF;
        for (int k = threadIdx.x; k < cnt; k += CR_TAIL_THREADS) cur[k] = src[k];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < J; j++) {
        const int64_t own_lo = Q0 << (J - j - 1), own_hi = last ? t.n[j + 1] : (Q1 << (J - j - 1));
        const int cnt = (int)(hi[j + 1] - lo[j + 1] + 1) * NF;
        T* out = C.b + t.off[j + 1] * NF;
        for (int k = threadIdx.x; k < cnt; k += CR_TAIL_THREADS) {
            const int64_t q = lo[j + 1] + k / NF;
            const int r = k % NF;
            const int64_t p = 2 * q + 1;
            const T* bl = cur + (p - lo[j]) * NF;          // row p of level j in the tile
            T acc = bl[r];
            const T* al = C.alpha + (t.off[j] + p) * 25 + r * NF;
#pragma unroll
            for (int kk = 0; kk < NF; kk++) acc = madd(acc, al[kk], bl[kk - NF]);
            if (p + 1 < t.n[j]) {
                const T* ga = C.gamma + (t.off[j] + p) * 25 + r * NF;
#pragma unroll
                for (int kk = 0; kk < NF; kk++) acc = madd(acc, ga[kk], bl[kk + NF]);
            }
            nxt[k] = acc;
            if (q >= own_lo && q < own_hi) out[q * NF + r] = acc;
        }
        __syncthreads();
        T* sw = cur; cur = nxt; nxt = sw;
    }
}

template <int J>
__global__ void __launch_bounds__(CR_TAIL_THREADS) cr_rhs_tail_kernel(CrSystem<double> Cr, CrSystem<cplx> Cc, CrTail t, const double* __restrict__ b0_r,
                                                                       const cplx* __restrict__ b0_c)
{
    __shared__ cplx lds[(CR_TAIL_TILE + CR_TAIL_TILE / 2 + 2) * NF];   // (the real system uses half of the bytes)
    if (blockIdx.y == 0) cr_rhs_tail<double, J>(Cr, t, b0_r, reinterpret_cast<double*>(lds));
    else cr_rhs_tail<cplx, J>(Cc, t, b0_c, lds);
}

// backward: x_{l0+J} (in C.b, left there by the PCR solve) -> x_{l0}, written over b_{l0}; the levels in between live in LDS only
template <class T, int J>
__device__ void cr_back_tail(const CrSystem<T>& C, const CrTail& t, T* b0, T* lds)
{
    const int64_t tile = (int64_t)CR_TAIL_TOP << J;
    const int64_t P0 = (int64_t)blockIdx.x * tile;
    const int64_t P1 = (P0 + tile < t.n[0]) ? P0 + tile : t.n[0];
    int64_t lo[J + 1], hi[J + 1];
    lo[0] = P0; hi[0] = P1 - 1;
#pragma unroll
    for (int j = 0; j < J; j++) {
        {
            const int64_t l = (lo[j] >> 1) - 1, h = hi[j] >> 1;       // even p needs x_{p/2 - 1}, x_{p/2}; odd p x_{(p-1)/2}
            lo[j + 1] = l < 0 ? 0 : l;
            hi[j + 1] = h > t.n[j + 1] - 1 ? t.n[j + 1] - 1 : h;
        }
    }
    T* cur = lds;                               // x of level j + 1
    T* nxt = lds + CR_TAIL_TILE * NF;           // x of level j
    {
        const T* src = C.b + (t.off[J] + lo[J]) * NF;
        const int cnt = (int)(hi[J] - lo[J] + 1) * NF;
        for (int k = threadIdx.x; k < cnt; k += CR_TAIL_THREADS) cur[k] = src[k];
    }
    __syncthreads();
#pragma unroll
    for (int j = J - 1; j >= 0; j--) {
        const int cnt = (int)(hi[j] - lo[j] + 1) * NF;
        T* bj = (t.l0 + j == 0) ? b0 : C.b + t.off[j] * NF;
        // j == 0 writes over the right-hand sides it reads (a row's five threads read each other's): every value of a pass is formed
        // before any is stored.  cnt <= 5 tile rows = at most 5 passes of the 256 threads.
        T keep[((CR_TAIL_TOP << J) * NF + CR_TAIL_THREADS - 1) / CR_TAIL_THREADS];
        int pass = 0;
        for (int k0 = 0; k0 < cnt; k0 += CR_TAIL_THREADS, pass++) {
            const int k = k0 + threadIdx.x;
            if (k >= cnt) continue;
            const int64_t p = lo[j] + k / NF;
            const int r = k % NF;
            const int64_t q = p >> 1;
            T acc;
            if (p & 1) acc = cur[(q - lo[j + 1]) * NF + r];
            else {
                const T* bp = bj + p * NF;
                const T* di = C.Dinv + (t.off[j] + p) * 25 + r * NF;
                acc = mul1(di[0], bp[0]);
#pragma unroll
                for (int kk = 1; kk < NF; kk++) acc = madd(acc, di[kk], bp[kk]);
                if (q >= 1) {
                    const T* pr = C.P + (t.off[j] + p) * 25 + r * NF;
                    const T* xm = cur + (q - 1 - lo[j + 1]) * NF;
#pragma unroll
                    for (int kk = 0; kk < NF; kk++) acc = madd(acc, pr[kk], xm[kk]);
                }
                if (q < t.n[j + 1]) {
                    const T* qr = C.Q + (t.off[j] + p) * 25 + r * NF;
                    const T* xp = cur + (q - lo[j + 1]) * NF;
#pragma unroll
                    for (int kk = 0; kk < NF; kk++) acc = madd(acc, qr[kk], xp[kk]);
                }
            }
            if (j > 0) nxt[k] = acc;
            else {
#pragma unroll
                for (int u = 0; u < (int)(sizeof(keep) / sizeof(keep[0])); u++)
                    if (u == pass) keep[u] = acc;
            }
        }
        __syncthreads();
        if (j == 0) {
            pass = 0;
            for (int k0 = 0; k0 < cnt; k0 += CR_TAIL_THREADS, pass++) {
                const int k = k0 + threadIdx.x;
                if (k >= cnt) continue;
                T v = keep[0];
#pragma unroll
                for (int u = 1; u < (int)(sizeof(keep) / sizeof(keep[0])); u++)
                    if (u == pass) v = keep[u];
                bj[lo[0] * NF + k] = v;
            }
        }
        T* sw = cur; cur = nxt; nxt = sw;
    }
}

template <int J>
__global__ void __launch_bounds__(CR_TAIL_THREADS) cr_back_tail_kernel(CrSystem<double> Cr, CrSystem<cplx> Cc, CrTail t, double* b0_r, cplx* b0_c)
{
    __shared__ cplx lds[(CR_TAIL_TILE + CR_TAIL_TILE / 2 + 2) * NF];
    if (blockIdx.y == 0) cr_back_tail<double, J>(Cr, t, b0_r, reinterpret_cast<double*>(lds));
    else cr_back_tail<cplx, J>(Cc, t, b0_c, lds);
}

}  // namespace radau
}  // namespace marl
