// marl_math.h - device-side point evaluation of the five-field L'Heureux RHS (gfx950 only).
//
// point_local() + point_rates() (together: rhs_point()) evaluate what one iteration of the reference's
// depth loop computes (marlpde/LHeureux_model.py:413-520, Appendix A of SURVEY.md) for one cell, given the
// cell's five values (own-cell phase) and those of its two neighbours (stencil phase).  The 13 separate stencil passes of the
// reference (:372-384) are folded in; virtual (ghost) cells are synthesised by the caller with
// ghost_lower()/ghost_upper() following the py-pde boundary rules the reference configures at
// LHeureux_model.py:26-30.
//
// fp64 throughout.  This path is bound by the fp64 vector ALU, not by HBM, so the
// evaluation is organised to minimise instruction count:
//   * reciprocals 1/Phi, 1/(1-Phi), 1/den are formed once (v_rcp_f64 + 2 Newton steps) and reused;
//     the reference divides 13 times per cell;
//   * of each clamp pair (min(x,1), max(x,1)) one power has base exactly 0, so ONE pow per pair
//     is evaluated; the aragonite-undersaturation power is only evaluated inside the
//     dissolution zone (its factor `mask` is 0 elsewhere, LHeureux_model.py:486-487);
//   * exp and log are table-driven (3 KB of tables in LDS, ~20-25 instructions each instead of OCML's
//     ~40 / ~96); pow(b, e) for the kinetics exponents is exp(e*log b) (error analysis in DESIGN.md);
//   * coth(Pe) - 1/Pe is 1 + 2/(e^(2Pe) - 1) - 1/Pe and only taken in the mid Peclet range; the
//     branch is wave-uniform in practice (hard part 1c);
//   * later Runge-Kutta stages expand the transcendentals around an earlier evaluation (TR_* modes below);
//   * rare paths sit behind real wave-uniform branches (`asm volatile("")` stops their speculation).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "marl_tables.h"

namespace marl {

constexpr int NF = 5;  // CA, CC, cCa, cCO3, Phi  (marlpde/Evolve_scenario.py:76-86)

// Uniform (per model instance) constants, derived on the host in marl_api.hip:derive_consts()
// from marl_params exactly as the reference's constructor does (LHeureux_model.py:36-72,130-133).
// HotConsts is what every cell evaluation reads; kernels copy it into registers (SGPRs) once.
struct HotConsts {
    double inv_dx, inv_dx2;          // 1/dx, 1/dx^2, dx = length/N (py-pde grid.discretization)
    double presum, rhorat;           // :71-72, :69-70
    double KRat, nu1, nu2;           // :51, :36, :39
    double lambda_, Da, delta;       // :64, :63, :62
    double dCa, dCO3, dPhi;          // :60, :61, :132-133 (dPhi_fixed)
    double pe_cCa, pe_cCO3, pe_Phi;  // delta_x/(2 dCa), delta_x/(2 dCO3), delta_x/(2 dPhi_fixed)  (:436,444,452)
    double m1, m2, n1, n2;           // kinetics exponents
    double hdx;                      // 0.5/dx
    double pe_smax;                  // max(pe_cCa, pe_cCO3)
    double Dal;                      // Da * lambda_
    double Da_nu1, Dal_nu2;          // Da * nu1, Da * lambda_ * nu2: the reaction prefactors folded into the saturation terms (tA, tC below)
    double rr10;                     // 10 * rhorat
    double dPhi_dx2;                 // dPhi / dx^2
    double auxcon;                   // :65-66 (the time-varying porosity diffusion coefficient, dPhi_variable)
    int32_t fv;                      // FV_switch
    int32_t generic_p0;              // some exponent <= 0, i.e. some pow(0, e) != 0: take the general combination
    int32_t var_dphi;                // marl_params.dPhi_variable (host side: selects the VD kernel instantiation)
    int32_t no_reuse;                // option "no_reuse": no centre ever validates -> every evaluation takes the full path (bench: worst case)
};

struct DevConsts {
    HotConsts hot;
    double p0_m1, p0_m2, p0_n1, p0_n2;  // pow(0, exponent): value of the clamp-pair member whose base is exactly 0
    double bc[NF];                   // Dirichlet values at x = 0  (:26-30)
    int64_t N;                       // cells of the (global) grid
    int64_t mask_lo, mask_hi;        // cells [mask_lo, mask_hi) have not_too_shallow*not_too_deep == 1 (Evolve_scenario.py:51-54)
    int64_t reserved;
};

constexpr double PECLET_MIN = 1e-2, PECLET_MAX = 1.0 / PECLET_MIN;  // LHeureux_model.py:87-88
constexpr double FV_SERIES_MAX = 0.5;   // below: power series of coth x - 1/x (fv_sigma)

// Pin a uniform value in scalar registers: the empty asm makes it opaque, so under register pressure
// the compiler spills it (v_writelane) instead of re-issuing the s_load + s_waitcnt it came from.
__device__ __forceinline__ double pin_uniform(double v)
{
    asm volatile("" : "+s"(v));
    return v;
}

__device__ __forceinline__ HotConsts load_hot(const DevConsts* __restrict__ c)
{
    HotConsts k = c->hot;
    double* f = reinterpret_cast<double*>(&k);
#pragma unroll
    for (int i = 0; i < (int)(offsetof(HotConsts, fv) / sizeof(double)); i++) f[i] = pin_uniform(f[i]);
    asm volatile("" : "+s"(k.fv), "+s"(k.generic_p0), "+s"(k.no_reuse));
    return k;
}

__device__ __forceinline__ double rcp_nr(double x)
{
    // v_rcp_f64 seed + two Newton-Raphson steps: <= 1 ulp for finite non-zero x.  x = 0 or inf gives NaN
    // (the correction computes 0 * inf); every caller's result is non-finite in the reference too there.
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r0, 1.0);
    const double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-x, r1, 1.0);
    return __builtin_fma(r1, e, r1);
}

// x^n for a small positive integer n, rounded once: the running product is kept as an unevaluated sum hi + lo (one fma recovers the
// rounding error of every multiplication), so the result is the correctly rounded power except in ~2^-47 of all cases - which is what
// libm's pow(x, n) delivers and what scipy's Newton divergence tests (rate ** (NEWTON_MAXITER - k), radau.py:108 / bdf.py:60) are
// therefore compared with on the host, in scipy and in the oracle.  OCML's pow is not correctly rounded (a knife-edge rate
// test could fall the other way on the device).
__host__ __device__ __forceinline__ double pow_small_int(double x, int n)
{
    double hi = x, lo = 0.0;
    for (int i = 1; i < n; i++) {
        const double p = hi * x;
        const double e = __builtin_fma(hi, x, -p) + lo * x;
        hi = p + e;
        lo = (p - hi) + e;
    }
    return hi;
}

// ---------------------------------------------------------------------------------------------
// Table-driven fp64 log / exp.  OCML's log costs ~96 and pow ~224 vector instructions per call; this
// path is fp64-VALU-bound, so both are replaced by the classic table + short polynomial scheme with
// the 3 KB of tables staged in LDS (one ds_read per call; neighbouring cells hit the same entry, which
// the LDS broadcasts).  Accuracy ~1 ulp (tests/test_gpu_math.py); special operands take OCML's path.
// ---------------------------------------------------------------------------------------------
struct Tables {
    const double* log_tab;  // LDS: {1/c, -log(1/c)} x 128
    const double* exp_tab;  // LDS: 2^(i/128) x 128
};
constexpr int TABLE_DOUBLES = 3 * tab::N;

// Cooperative copy of the tables into LDS; ends with a barrier.
__device__ __forceinline__ Tables load_tables(double* lds, int nthreads)
{
    for (int i = threadIdx.x; i < 2 * tab::N; i += nthreads) lds[i] = tab::LOG_TAB[i];
    for (int i = threadIdx.x; i < tab::N; i += nthreads) lds[2 * tab::N + i] = tab::EXP_TAB[i];
    __syncthreads();
    return Tables{lds, lds + 2 * tab::N};
}

__device__ __forceinline__ double fast_log(double x, const Tables& T)
{
    // Straight-line main path (the table read can be scheduled early, among independent work); operands that
    // are not positive normal numbers (<= 0, subnormal, inf, NaN) are redone by OCML afterwards - a branch no
    // wave takes on a valid state.
    const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    const int32_t t = (int32_t)(hi - 0x3fe60000u);
    const int i = (t >> 13) & (tab::N - 1);
    const int k = t >> 20;
    const double z = __hiloint2double((int)(hi - ((uint32_t)t & 0xfff00000u)), (int)lo);
    const double invc = T.log_tab[2 * i], logc = T.log_tab[2 * i + 1];
    const double r = __builtin_fma(z, invc, -1.0);
    const double r2 = r * r;
    // log1p(r) - r = r^2 (-1/2 + r/3 - r^2/4 + r^3/5 - r^4/6),  |r| < 2^-8: truncation < 2e-18
    const double p = __builtin_fma(r, 1.0 / 3, -0.5) + r2 * (__builtin_fma(r, 1.0 / 5, -0.25) + r2 * (-1.0 / 6));
    double res = __builtin_fma((double)k, tab::LN2, logc) + __builtin_fma(r2, p, r);
    if (__builtin_expect(!__builtin_amdgcn_class(x, 1 << 8), 0)) res = log(x);
    return res;
}

__device__ __forceinline__ double fast_exp(double x, const Tables& T)
{
    // exp x = 2^k 2^(i/128) e^r,  x = (128 k + i) ln2/128 + r,  |r| <= ln2/256.  ldexp gives gradual
    // underflow and 0 for very negative x (down to about -1e7); NaN stays NaN.  Not for x = -inf, x > 709.
    const double kd = __builtin_rint(x * tab::INV_LN2N);
    const int ki = (int)kd;
    double r = __builtin_fma(kd, -tab::LN2N_HI, x);
    r = __builtin_fma(kd, -tab::LN2N_LO, r);
    const double s = ldexp(T.exp_tab[ki & (tab::N - 1)], ki >> 7);
    const double r2 = r * r;
    // e^r - 1 = r + r^2 (1/2 + r/6) + r^4 (1/24 + r/120),  truncation r^6/720 < 6e-19
    const double tmp = r + r2 * __builtin_fma(r, 1.0 / 6, 0.5) + (r2 * r2) * __builtin_fma(r, 1.0 / 120, 1.0 / 24);
    return __builtin_fma(s, tmp, s);
}

// pow(b, e) for b >= 0 (a clamped saturation distance) and a real kinetics exponent: exp(e log b).
// Relative error ~ (2 + 2.5 |e ln b|) ulp: ~1e-15 for b > 1e-3; for smaller b the value b^e itself is
// negligible against the O(1) terms it is added to.  b = 0 gives 0 (right for e > 0; the
// general pow(0, e) is patched in by the caller when an exponent is <= 0).
__device__ __forceinline__ double pow_sat(double b, double e, const Tables& T)
{
    const double r = fast_exp(e * fast_log(b, T), T);
    return (b == 0.0) ? 0.0 : r;
}

// Fiadeiro-Veronis weight sigma(Pe) for |Pe| >= PECLET_MIN; LHeureux_model.py:437-442 (= calculate_sigma :147-160)
// SERIES: the fused explicit integrators.  The stand-alone RHS (and with it the implicit path, whose Newton / step-size
// decisions are compared with scipy's run on the reference's RHS decision by decision) keeps the closed form throughout.
template <bool SERIES>
__device__ __forceinline__ double fv_sigma(double Pe, double W, const Tables& T)
{
    const double a = fabs(Pe);
    double s = 0.0;
    if (a > PECLET_MAX) {
        s = (W > 0.0) ? 1.0 : ((W < 0.0) ? -1.0 : W);  // np.sign(W) incl. 0 and NaN
    } else if (!(a < PECLET_MIN)) {
        if (SERIES && a <= FV_SERIES_MAX) {
            // coth x - 1/x = sum_k B_2k 4^k x^(2k-1) / (2k)!  (ratio of terms -> -x^2/pi^2): eleven terms are exact to
            // < 1e-16 relative for |x| <= 0.5 - where the closed form below cancels 1/x against coth x and loses
            // eps/x^2 - at a third of its cost (no exp, no reciprocal).  The one-workgroup sweeps (N = 1024,
            // |Pe_Phi| ~ 0.1 - 0.2) live here.
            const double x2 = Pe * Pe;
            double p = 2.3106432599002624e-11;
            p = __builtin_fma(p, x2, -2.2805151204592183e-10);
            p = __builtin_fma(p, x2, 2.2507846516808994e-09);
            p = __builtin_fma(p, x2, -2.2214608789979678e-08);
            p = __builtin_fma(p, x2, 2.1925947851873778e-07);
            p = __builtin_fma(p, x2, -2.1644042808063972e-06);
            p = __builtin_fma(p, x2, 2.1377799155576935e-05);
            p = __builtin_fma(p, x2, -1.0 / 4725);
            p = __builtin_fma(p, x2, 2.0 / 945);
            p = __builtin_fma(p, x2, -1.0 / 45);
            p = __builtin_fma(p, x2, 1.0 / 3);
            s = p * Pe;
        } else {
            // cosh/sinh - 1/Pe  ==  1 + 2/(e^(2 Pe) - 1) - 1/Pe ;  |Pe| in (0.5, 1e2]
            s = (1.0 + 2.0 * rcp_nr(fast_exp(2.0 * Pe, T) - 1.0)) - rcp_nr(Pe);
        }
    }
    return s;
}

// Virtual cell below x = 0: {"value": v} -> 2 v - u[0]   (all five fields)
__device__ __forceinline__ double ghost_lower(double bc, double u0) { return 2.0 * bc - u0; }
// Virtual cell above x = L: {"curvature": 0} for CA, CC -> 2 u[N-1] - u[N-2]; {"derivative": 0} -> u[N-1]
__device__ __forceinline__ double ghost_upper(int f, double uN1, double uN2) { return (f < 2) ? 2.0 * uN1 - uN2 : uN1; }

struct PointAux {
    double U, W;  // by-products needed by the monitors zeros_U / zeros_W (LHeureux_model.py:567-593)
};

// Transcendental reuse across Runge-Kutta stages (and steps).  The stage states of an explicit step on a fine grid
// differ from the step's first state by dt*rate - relatively 1e-10 at N = 2^20, 1e-6 at N = 65 536 - so log(Phi),
// the reciprocals, exp(10 - 10/Phi) and the calcite saturation power of a LATER evaluation are given to full fp64
// accuracy by 2nd/3rd-order expansions around the values an EARLIER evaluation (the "centre") computed; truncation
// < 1e-17 relative while every expansion variable is below REUSE_LIMIT.  Modes of one evaluation:
//   TR_PLAIN  no cache (stand-alone RHS, one-workgroup sweeps on coarse grids)
//   TR_FILL   full evaluation; becomes the centre
//   TR_REUSE  expand around the centre if it is `live`; any lane of the wave out of range (coarse grid, large step,
//             non-finite state) -> the whole wave evaluates in full and stops trying (`live` = false)
//   TR_AUTO   as TR_REUSE, but a wave that falls back becomes the new centre (first stage of every step: the centre
//             then survives from step to step inside one launch and is renewed only when it has drifted out of range)
// Storage: what the range check needs sits in registers; of the six cached values the last MARL_CACHE_LDS_SLOTS sit
// in LDS (one slot column per cell, STRIDE doubles between slots).  All in registers, a centre that lives across the
// step loop spills at 4 waves per SIMD (scratch reloads in every stage: -25 %); all in LDS, the extra ds_reads
// cost more than they save (the neighbour exchange already loads the LDS pipe) - profiles/r01_lab_cache_placement.log.
enum : int { TR_PLAIN = 0, TR_FILL = 1, TR_REUSE = 2, TR_AUTO = 3 };
constexpr double REUSE_LIMIT = 5e-5;   // 3rd-order expansions: truncation x^4 < 1e-17
constexpr double REUSE_TINY = 2e-6;    // below this the 3rd-order terms themselves are < 1e-17: 2nd order suffices

#ifndef MARL_CACHE_LDS_SLOTS   // how many of the slots (counted from the end of the list) live in LDS; the rest in registers
#define MARL_CACHE_LDS_SLOTS 4
#endif
// slots: 1/Phi, 1/(1-Phi), 1/den, 1/(O2-1), exp(10-10/Phi), calcite term
enum : int { PC_INVPHI, PC_INVOM, PC_INVDEN, PC_IB, PC_E, PC_TC, PC_SLOTS };
constexpr int PC_FIRST_LDS = PC_SLOTS - (MARL_CACHE_LDS_SLOTS < PC_SLOTS ? MARL_CACHE_LDS_SLOTS : PC_SLOTS);
constexpr int PC_LDS_SLOTS = PC_SLOTS - PC_FIRST_LDS;

template <int STRIDE>   // doubles between the LDS slots of one cell; 0: every slot in registers
struct PointCache {
    static constexpr int FIRST_LDS = STRIDE > 0 ? PC_FIRST_LDS : PC_SLOTS;
    // centre: Phi, O2 = cCa*cCO3; cPhi bounds every porosity expansion variable per unit |Phi - centre|;
    // nib = n/(O2-1): first-order variable of the calcite saturation power (n: the exponent in use)
    double Phi, O2, cPhi, nib;
    double* s;                       // this cell's LDS slots (slot j at s[(j - FIRST_LDS) * STRIDE])
    double v[FIRST_LDS > 0 ? FIRST_LDS : 1];
    bool fv_quiet;                   // all three Peclet numbers were < 0.9 PECLET_MIN at the centre
    __device__ __forceinline__ double get(int j) const { return j < FIRST_LDS ? v[j] : s[(j - FIRST_LDS) * STRIDE]; }
    __device__ __forceinline__ void set(int j, double x) { if (j < FIRST_LDS) v[j] = x; else s[(j - FIRST_LDS) * STRIDE] = x; }
};

// One RHS evaluation of a cell is split in two phases, so that the neighbour values (10 doubles) are not held in
// registers while the transcendental part runs, and so that a stencil kernel can do the own-cell work BEFORE the
// barrier of its neighbour exchange:
//   point_local  - everything that depends on the cell's own five values (porosity functions, velocities, reaction
//                  terms, Peclet test): results in PointLocal
//   point_rates  - the stencil part: differences against u[i-1], u[i+1] and the five rates
struct PointLocal {
    double Ux, R0, R1;          // |U|/dx and the reaction parts of the two solid rates
    double W, invPhi, G2, G3;   // solutes: advection velocity, 1/Phi, Da (1-Phi)(coA - lambda coC)(delta - c)/Phi
    double h1x, h2f;            // Phi/den/dx^2, (2+den)/den^2  (common_helper1/2, :470-473)
    double Q4, DaR;             // porosity: W - Phi dW/dPhi-part (:495), Da (1-Phi)(coA - lambda coC)
    double Wd;                  // W den (Peclet numbers; only read when fv_active)
    bool upw, fv_active;
    bool fv_solutes;            // wave-uniform: some lane of the wave has a solute Peclet number >= PECLET_MIN
};

// uc: the cell's five values.  in_mask: cell inside the dissolution zone.  K: hot constants (registers); C: the
// instance's full constant block (cold parts are read from memory only on rare paths).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"  // den ... tA are set on exactly one of the two paths below
// VD: the time-varying porosity diffusion coefficient (marl_params.dPhi_variable) - compiled in only where asked for.
// LEGACY: the operation order of rounds 1 - 2 (reaction prefactors applied after the products, F formed explicitly, (delta - c) differences).
// The stand-alone RHS keeps it: the implicit path built on that RHS is pinned to scipy decision by decision, and a last-bit change of the
// RHS flips Newton / step-size decisions.  The fused explicit integrators take the shorter forms (round 3: six
// instructions less per evaluation - prefactors folded into the saturation terms, rhorat F and DaRi (delta - c) as one fma each, the
// second-order reciprocal updates as x (1 - x) forms).
template <int MODE, int STRIDE, bool VD = false, bool LEGACY = false>
__device__ __forceinline__ void point_local(const double (&uc)[NF], bool in_mask, const HotConsts& K, const DevConsts* __restrict__ C,
                                            const Tables& T, PointLocal& pl, PointAux& aux, PointCache<STRIDE>& pc, bool& live)
{
    static_assert(!LEGACY || MODE == TR_PLAIN, "the legacy operation order is the stand-alone RHS's (no transcendental cache)");
#ifdef MARL_ABLATE_CORE  // kernel-lab builds only: the skeleton (loads, LDS exchange, barriers, RK combinations)
    aux.U = aux.W = 0.0;
    return;
#endif
    const double CA = uc[0], CC = uc[1], c = uc[2], o = uc[3], Phi = uc[4];
    const double omPhi = 1.0 - Phi;
    const double O2 = c * o;

    // ---- porosity-only quantities (:414-429) and the saturation terms (:479-493):
    //      tC = (O2-1)^n1 if O2 > 1 else -nu2 (1-O2)^n2 ;  tA = (1-O3)^m2 mask if O3 < 1 else -nu1 (O3-1)^m1
    //      (of each clamp pair (min(x,1), max(x,1)) one power has base exactly 0 and the other has base |x - 1|;
    //      with positive exponents the zero-base member vanishes)
    double den, invPhi, invom, invden, ex, tC, tA;
    bool fv_check = K.fv != 0;
    bool reuse = false;
    if constexpr (MODE == TR_FILL) live = false;
    if ((MODE == TR_REUSE || MODE == TR_AUTO) && live) {   // `live` is wave-uniform
        const double d = Phi - pc.Phi, dO = O2 - pc.O2;
        const double v = dO * pc.nib;
        // every expansion variable below is bounded by this sum (NaN compares false -> falls back)
        const double big = __builtin_fma(fabs(d), pc.cPhi, fabs(v));
        reuse = __builtin_amdgcn_ballot_w64(!(big < REUSE_LIMIT)) == 0;
        live = reuse;
        if (reuse) {
            const bool tiny = __builtin_amdgcn_ballot_w64(!(big < REUSE_TINY)) == 0;
            const double invPhi0 = pc.get(PC_INVPHI), invom0 = pc.get(PC_INVOM), invden0 = pc.get(PC_INVDEN), ib0 = pc.get(PC_IB);
            const double e0 = pc.get(PC_E), tC0 = pc.get(PC_TC);
            const double x = d * invPhi0, y = d * invom0, u = dO * ib0;
            double l1p, ip, w, gy;
            double da;                                                                    // change of 10 - 10/Phi = -10 (ip - invPhi0)
            if (tiny) {
                l1p = x * __builtin_fma(x, -0.5, 1.0);                                    // log1p(x)
                const double t10 = invPhi0 * __builtin_fma(-x, x, x);                     // invPhi0 x (1 - x)
                ip = invPhi0 - t10;                                                       // 1/(Phi0 (1+x)) = invPhi0 (1 - x + x^2)
                da = 10.0 * t10;
                w = v * __builtin_fma(u, -0.5, 1.0);                                      // n log1p(u)
                gy = __builtin_fma(y, 1.0 + y, 1.0);                                      // 1/(1-y)
            } else {
                l1p = x * __builtin_fma(x, __builtin_fma(x, 1.0 / 3, -0.5), 1.0);
                ip = invPhi0 * __builtin_fma(-x, __builtin_fma(-x, 1.0 - x, 1.0), 1.0);
                da = -10.0 * (ip - invPhi0);
                w = v * __builtin_fma(u, __builtin_fma(u, 1.0 / 3, -0.5), 1.0);
                gy = __builtin_fma(y, __builtin_fma(y, 1.0 + y, 1.0), 1.0);
            }
            const double z = -2.0 * l1p * invden0;                                        // (den - den0)/den0
            invPhi = ip;
            invom = invom0 * gy;
            if (tiny) {
                invden = __builtin_fma(-invden0, __builtin_fma(-z, z, z), invden0);       // invden0 (1 - z + z^2)
                ex = e0 * __builtin_fma(da, __builtin_fma(da, 0.5, 1.0), 1.0);
                tC = tC0 * __builtin_fma(w, __builtin_fma(w, 0.5, 1.0), 1.0);
            } else {
                invden = invden0 * __builtin_fma(-z, __builtin_fma(-z, 1.0 - z, 1.0), 1.0);
                ex = e0 * __builtin_fma(da, __builtin_fma(da, __builtin_fma(da, 1.0 / 6, 0.5), 1.0), 1.0);
                tC = tC0 * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.0 / 6, 0.5), 1.0), 1.0);
            }
            // wave-uniform on purpose: a per-lane condition is if-converted, and the Peclet test (with the reciprocal
            // that recovers den) would then run in every evaluation
            fv_check = fv_check && __builtin_amdgcn_ballot_w64(!pc.fv_quiet) != 0;
            if (fv_check) {
                asm volatile("");
                den = rcp_nr(invden);   // only the Peclet numbers need den itself
            }
        }
    }
    if (!reuse) {
        // ONE reciprocal serves 1/Phi, 1/(1-Phi), 1/den.
#ifdef MARL_ABLATE_LOG  // kernel-lab builds only (tools/rk4_lab.hip): price of the pieces
        const double L = Phi - 1.0;
#else
        const double L = fast_log(Phi, T);
#endif
        den = __builtin_fma(-2.0, L, 1.0);
        const double pd = Phi * den, od = omPhi * den, po = Phi * omPhi;
#ifdef MARL_ABLATE_RCP
        const double rall = 2.0 - pd * omPhi;
#else
        const double rall = rcp_nr(pd * omPhi);
#endif
        invPhi = rall * od, invom = rall * pd, invden = rall * po;
#ifdef MARL_ABLATE_EXP
        ex = 0.01 * __builtin_fma(-10.0, invPhi, 10.0);
#else
        ex = fast_exp(__builtin_fma(-10.0, invPhi, 10.0), T);
#endif
        const bool over = O2 > 1.0;
        const double nsel = over ? K.n1 : K.n2;
#ifdef MARL_ABLATE_POW
        const double pwC = fabs(O2 - 1.0) * nsel;
#else
        const double pwC = pow_sat(fabs(O2 - 1.0), nsel, T);
#endif
        tC = (LEGACY ? (over ? 1.0 : -K.nu2) : (over ? K.Dal : -K.Dal_nu2)) * pwC;   // (the prefactor Da lambda of the calcite term rides along: DC = CC tC below)
        if (K.generic_p0) {  // an exponent <= 0: pow(0, e) is 1 or inf instead of 0 (rare; constants from memory)
            asm volatile("");  // keep this a (uniform) branch: if-converted, its arithmetic would run on every evaluation
            const double y1 = C->p0_n1, y2 = C->p0_n2, nu2 = C->hot.nu2;
            const double pcw = (O2 == 1.0) ? (over ? y1 : y2) : pwC;
            tC = (LEGACY ? 1.0 : K.Dal) * (over ? pcw - nu2 * y2 : y1 - nu2 * pcw);
        }
        if constexpr (MODE == TR_FILL || MODE == TR_AUTO) {
            // generic_p0 (an exponent < 1 or <= 0: the clamp-pair shortcut / the bound |u| <= |n u| do not hold): never reuse
            // (option no_reuse: the same NaN makes every later range check fail - the fallback path of every wave, always)
            const double ib = (K.generic_p0 | K.no_reuse) ? __builtin_nan("") : rcp_nr(O2 - 1.0);
            pc.Phi = Phi; pc.O2 = O2; pc.nib = nsel * ib;
            // |x| = |d| invPhi, |y| = |d| invom, |10 d(1/Phi)| <= 12 |d| invPhi^2, |z| <= 2.2 |d| invPhi invden
            pc.cPhi = fmax(fmax(fabs(invPhi), fabs(invom)), fmax(12.0 * (invPhi * invPhi), 2.2 * fabs(invPhi * invden)));
            pc.set(PC_INVPHI, invPhi); pc.set(PC_INVOM, invom); pc.set(PC_INVDEN, invden); pc.set(PC_E, ex);
            pc.set(PC_IB, ib); pc.set(PC_TC, tC);
            live = true;
        }
    }
    {   // aragonite term: always evaluated in full - it vanishes outside the dissolution zone while undersaturated
        // (whole waves skip the power), and caching it as well costs more registers / LDS reads than it saves
        const double O3 = O2 * K.KRat;
        const bool under = O3 < 1.0;
        tA = O3 - O3;                               // 0, or NaN for a non-finite O3 (keeps the reference's NaN visible)
        if (!under || in_mask) {
#ifdef MARL_ABLATE_POWA   // kernel-lab builds only
            const double pwA = fabs(O3 - 1.0) * (under ? K.m2 : K.m1);
#else
            const double pwA = pow_sat(fabs(O3 - 1.0), under ? K.m2 : K.m1, T);
#endif
            tA = (LEGACY ? (under ? 1.0 : -K.nu1) : (under ? K.Da : -K.Da_nu1)) * pwA;  // [Da] ((1-O3)^m2 * mask  |  -nu1 (O3-1)^m1): DA = CA tA below
        }
        if (K.generic_p0) {
            asm volatile("");  // a real branch, as above
            const double z1 = C->p0_m1, z2 = C->p0_m2, nu1 = C->hot.nu1;
            const double mask = in_mask ? 1.0 : 0.0;
            const double pa = (O3 == 1.0) ? (under ? z2 : z1) : ((!under || in_mask) ? fast_exp((under ? K.m2 : K.m1) * fast_log(fabs(O3 - 1.0), T), T) : 0.0);
            tA = (LEGACY ? 1.0 : K.Da) * (under ? pa * mask - nu1 * z1 : z2 * mask - nu1 * pa);
        }
    }
    const double rF = LEGACY ? K.rhorat * (1.0 - ex) : __builtin_fma(-K.rhorat, ex, K.rhorat);   // rhorat F,  F = 1 - exp(10 - 10/Phi)
    const double t2 = rF * (Phi * Phi);
    const double W = K.presum - t2;
    const double U = __builtin_fma(t2 * Phi, invom, K.presum);
    aux.U = U;
    aux.W = W;
    double wpe = fabs(W) * K.pe_Phi;   // |Peclet_Phi| (:452)
    if constexpr (VD) {   // dPhi = auxcon F Phi^3/(1-Phi) (:430, commented out in the reference) instead of dPhi_fixed
        const double dPhi = K.auxcon * ((1.0 - ex) * (Phi * Phi)) * (Phi * invom);
        wpe = fabs(W) * (K.pe_Phi * (K.dPhi * rcp_nr(dPhi)));   // delta_x / (2 dPhi)
    }

    // ---- reaction terms (:479-493)
    const double DA = LEGACY ? K.Da * (CA * tA) : CA * tA;     // Da coA         (tA, tC carry Da / Da lambda unless LEGACY)
    const double DC = LEGACY ? K.Dal * (CC * tC) : CC * tC;    // Da lambda coC
    const double DmD = DA - DC;
    const double DaR = omPhi * DmD;              // Da (1-Phi) (coA - lambda coC)
    // solids (:498-503): (1-CA) DA + CA DC = DA - CA (DA - DC);  CC DA + (1-CC) DC = DC + CC (DA - DC)
    pl.upw = U > 0.0;
    pl.Ux = fabs(U * K.inv_dx);
    pl.R0 = __builtin_fma(CA, DmD, -DA);
    pl.R1 = __builtin_fma(CC, DmD, DC);

    // ---- Fiadeiro-Veronis weights (:433-462) all vanish when every |Pe| < PECLET_MIN (always on fine grids)
    bool fv_active = false;
    pl.Wd = 0.0;
#ifdef MARL_ABLATE_FV   // kernel-lab builds only: central gradients throughout
    fv_check = false;
#endif
    pl.fv_solutes = MODE != TR_PLAIN;   // (cached modes: always the three-sigma form, see below)
    if (fv_check) {   // wave-uniform
        asm volatile("");   // a real branch (its five cheap operations would otherwise be speculated into every evaluation)
        const double Wd = W * den;
        const double psol = fabs(Wd) * K.pe_smax;   // the larger of the two solutes' |Peclet|
        const double pmax = fmax(psol, wpe);
        fv_active = !(pmax < PECLET_MIN);
        // wave-uniform: does ANY lane need a weight for cCa / cCO3?  Their diffusion coefficients are ~200x the porosity's, so from
        // N ~ 100 cells on only the porosity's Peclet number reaches PECLET_MIN (every BASELINE size) - point_rates then evaluates one
        // sigma instead of three (a sigma of 0 gives the central gradient bit for bit, so nothing changes but the instruction count)
        // Only without the transcendental cache (MODE == TR_PLAIN: the one-workgroup sweeps and the stand-alone RHS - the coarse grids,
        // where the weights are live in every evaluation): the fused fine-grid kernels sit exactly at their 128-VGPR cap, and one more
        // wave-uniform value carried across the exchange barrier tipped rk45_attempt_kernel into spilling inside its stage sequence
        // (0 -> 28 B/lane of scratch, rk45_single -9 %) for a branch those kernels never take.
        if constexpr (MODE == TR_PLAIN) pl.fv_solutes = __builtin_amdgcn_ballot_w64(!(psol < PECLET_MIN)) != 0;
        pl.Wd = Wd;
        if constexpr (MODE == TR_FILL || MODE == TR_AUTO) { if (!reuse) pc.fv_quiet = pmax < 0.9 * PECLET_MIN; }  // in range, Pe moves by < 1e-3 relative
    } else if constexpr (MODE == TR_FILL || MODE == TR_AUTO) {
        if (!reuse) pc.fv_quiet = true;  // FV_switch off
    }
    pl.fv_active = fv_active;
    pl.W = W;
    pl.invPhi = invPhi;
    const double DaRi = DaR * invPhi;
    if constexpr (LEGACY) {
        pl.G2 = DaRi * (K.delta - c);                               // :506-509, :512-515
        pl.G3 = DaRi * (K.delta - o);
    } else {
        const double Dd = DaRi * K.delta;
        pl.G2 = __builtin_fma(-DaRi, c, Dd);                        // DaRi (delta - c)
        pl.G3 = __builtin_fma(-DaRi, o, Dd);
    }
    pl.h1x = (Phi * invden) * K.inv_dx2;                            // Phi/den / dx^2
    pl.h2f = invden * __builtin_fma(2.0, invden, 1.0);              // (2+den)/den^2
    const double q = __builtin_fma(rF, __builtin_fma(2.0, Phi, 10.0), -K.rr10);  // rhorat (2 Phi F + 10 (F-1))  (:495)
    pl.Q4 = __builtin_fma(-Phi, q, W);                              // -(dWdx Phi + W Phi') = -Phi' (W - Phi q)  (:518-520)
    pl.DaR = DaR;
}
#pragma clang diagnostic pop

// uc/um/up: values at cell i, i-1, i+1 (ghosts already substituted).  r: the five rates (LHeureux_model.py:498-520).
// mixed_upwind (wave-uniform): false promises pl.upw in every lane of the wave - up[0], up[1] are then not read.
// SOLUTE_SKIP: honour pl.fv_solutes (callers without the transcendental cache); false: the three-sigma form, untouched since round 3
template <bool VD = false, bool SERIES = true, bool SOLUTE_SKIP = false>
__device__ __forceinline__ void point_rates(const double (&uc)[NF], const double (&um)[NF], const double (&up)[NF],
                                            const HotConsts& K, const Tables& T, const PointLocal& pl, double (&r)[NF], bool mixed_upwind = true)
{
#ifdef MARL_ABLATE_CORE
    for (int f = 0; f < NF; f++) r[f] = (up[f] - um[f]) * K.hdx * 1e-9;
    return;
#endif
    const double CA = uc[0], CC = uc[1], c = uc[2], o = uc[3], Phi = uc[4];
    // dPhi = auxcon F Phi^3/(1-Phi) of this cell (VD only)
    auto dPhi_cell = [&]() {
        const double F = 1.0 - fast_exp(__builtin_fma(-10.0, pl.invPhi, 10.0), T);
        return K.auxcon * (F * (Phi * Phi)) * (Phi * rcp_nr(1.0 - Phi));
    };
    // ---- solids: upwinded one-sided difference (:372-384, :418-423).
    //   U > 0: -U (u - u[i-1])/dx;  else: -U (u[i+1] - u)/dx = -|U| (u - u[i+1])/dx  -> one difference, upwind neighbour
    if (mixed_upwind) {   // wave-uniform: some lane has U <= 0
        asm volatile("");
        r[0] = __builtin_fma(-pl.Ux, CA - (pl.upw ? um[0] : up[0]), pl.R0);
        r[1] = __builtin_fma(-pl.Ux, CC - (pl.upw ? um[1] : up[1]), pl.R1);
    } else {              // burial everywhere (the normal case): no per-lane selects
        r[0] = __builtin_fma(-pl.Ux, CA - um[0], pl.R0);
        r[1] = __builtin_fma(-pl.Ux, CC - um[1], pl.R1);
    }

    // ---- solutes and porosity.
    // one-sided differences (x dx): back = u - u[i-1], forw = u[i+1] - u (:372-384) - both exact for smooth fields
    // (Sterbenz), so forw - back is a second difference WITHOUT the 1-ulp-of-u noise of u[i-1] - 2u + u[i+1]
    // (which, times 1/dx^2, is what limits the reference's own RHS to ~1e-10 relative on fine grids).
    const double c_b = c - um[2], c_f = up[2] - c;
    const double o_b = o - um[3], o_f = up[3] - o;
    const double p_b = Phi - um[4], p_f = up[4] - Phi;
    const double p_d = p_f - p_b, c_d = c_f - c_b, o_d = o_f - o_b;
    double pg, cg, og;  // gradients (already divided by dx)
    if (!pl.fv_active) {   // weighted gradient 0.5*((1-s) forw + (1+s) back) with s = 0: the central one
        pg = (p_f + p_b) * K.hdx;
        cg = (c_f + c_b) * K.hdx;
        og = (o_f + o_b) * K.hdx;
    } else {
        const double W = pl.W, Wd = pl.Wd;
        const double pe_Phi = VD ? K.pe_Phi * (K.dPhi * rcp_nr(dPhi_cell())) : K.pe_Phi;   // delta_x / (2 dPhi)
        if constexpr (!SOLUTE_SKIP) {
            const double s_c = fv_sigma<SERIES>(Wd * K.pe_cCa, W, T), s_o = fv_sigma<SERIES>(Wd * K.pe_cCO3, W, T), s_p = fv_sigma<SERIES>(W * pe_Phi, W, T);
            cg = ((1.0 - s_c) * c_f + (1.0 + s_c) * c_b) * K.hdx;
            og = ((1.0 - s_o) * o_f + (1.0 + s_o) * o_b) * K.hdx;
            pg = ((1.0 - s_p) * p_f + (1.0 + s_p) * p_b) * K.hdx;
        } else {
            const double s_p = fv_sigma<SERIES>(W * pe_Phi, W, T);
            pg = ((1.0 - s_p) * p_f + (1.0 + s_p) * p_b) * K.hdx;
            if (pl.fv_solutes) {   // wave-uniform (see point_local)
                asm volatile("");
                const double s_c = fv_sigma<SERIES>(Wd * K.pe_cCa, W, T), s_o = fv_sigma<SERIES>(Wd * K.pe_cCO3, W, T);
                cg = ((1.0 - s_c) * c_f + (1.0 + s_c) * c_b) * K.hdx;
                og = ((1.0 - s_o) * o_f + (1.0 + s_o) * o_b) * K.hdx;
            } else {               // both sigmas are 0 in every lane: ((1 - 0) c_f + (1 + 0) c_b) hdx, the same bits
                cg = (c_f + c_b) * K.hdx;
                og = (o_f + o_b) * K.hdx;
            }
        }
    }
    const double h2 = pg * pl.h2f;                                      // common_helper2 (:472-473)
    const double Hc = K.dCa * __builtin_fma(h2, cg, pl.h1x * c_d);      // (:474-475)
    const double Ho = K.dCO3 * __builtin_fma(h2, og, pl.h1x * o_d);     // (:476-477)
    r[2] = __builtin_fma(-pl.W, cg, __builtin_fma(pl.invPhi, Hc, pl.G2));   // :506-509
    r[3] = __builtin_fma(-pl.W, og, __builtin_fma(pl.invPhi, Ho, pl.G3));   // :512-515
    double diffusion;   // dPhi * Phi_laplace + Da (1-Phi)(coA - lambda coC)
    if constexpr (VD) {
        diffusion = __builtin_fma(dPhi_cell() * K.inv_dx2, p_d, pl.DaR);
    } else {
        diffusion = __builtin_fma(K.dPhi_dx2, p_d, pl.DaR);
    }
    r[4] = __builtin_fma(-pg, pl.Q4, diffusion);  // :518-520
}

// Both phases back to back (stand-alone RHS).
template <int MODE, int STRIDE, bool VD = false>
__device__ __forceinline__ void rhs_point(const double (&uc)[NF], const double (&um)[NF], const double (&up)[NF],
                                          bool in_mask, const HotConsts& K, const DevConsts* __restrict__ C,
                                          const Tables& T, double (&r)[NF], PointAux& aux, PointCache<STRIDE>& pc, bool& live)
{
    PointLocal pl;
    point_local<MODE, STRIDE, VD, true>(uc, in_mask, K, C, T, pl, aux, pc, live);   // (LEGACY operation order: see point_local)
    point_rates<VD, false, true>(uc, um, up, K, T, pl, r);
}


}  // namespace marl
