// marl_math.h - device-side point evaluation of the five-field L'Heureux RHS (gfx950 only).
//
// One call of rhs_point() evaluates what one iteration of the reference's depth loop
// computes (marlpde/LHeureux_model.py:413-520, Appendix A of SURVEY.md) for one cell, given the
// cell's five values and those of its two neighbours.  The 13 separate stencil passes of the
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
//     branch is wave-uniform in practice (hard part 1c).
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
    int32_t fv;                      // FV_switch
    int32_t generic_p0;              // some exponent <= 0, i.e. some pow(0, e) != 0: take the general combination
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
    asm volatile("" : "+s"(k.fv), "+s"(k.generic_p0));
    return k;
}

__device__ __forceinline__ double rcp_nr(double x)
{
    // v_rcp_f64 seed + two Newton-Raphson steps: <= 1 ulp for normal x; x = 0 -> inf, inf -> 0, NaN -> NaN
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r0, 1.0);
    const double r1 = __builtin_fma(r0, e, r0);
    e = __builtin_fma(-x, r1, 1.0);
    const double r2 = __builtin_fma(r1, e, r1);
    // keep the seed's inf/0 (the correction turns them into NaN); a NaN seed stays NaN
    return (r2 == r2) ? r2 : r0;
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
    // positive normal numbers only; everything else (<= 0, subnormal, inf, NaN) through OCML
    if (!__builtin_amdgcn_class(x, 1 << 8)) return log(x);
    const uint32_t hi = (uint32_t)__double2hiint(x), lo = (uint32_t)__double2loint(x);
    const int32_t t = (int32_t)(hi - 0x3fe60000u);
    const int i = (t >> 13) & (tab::N - 1);
    const int k = t >> 20;
    const double z = __hiloint2double((int)(hi - ((uint32_t)t & 0xfff00000u)), (int)lo);
    const double invc = T.log_tab[2 * i], logc = T.log_tab[2 * i + 1];
    const double r = __builtin_fma(z, invc, -1.0);
    const double kd = (double)k;
    const double w = __builtin_fma(kd, tab::LN2_HI, logc);
    const double hi_ = w + r;
    const double lo_ = (w - hi_) + r + kd * tab::LN2_LO;
    const double r2 = r * r;
    // log1p(r) - r = r^2 (-1/2 + r/3 - r^2/4 + r^3/5 - r^4/6),  |r| < 2^-8: truncation < 2e-18
    const double p = __builtin_fma(r, 1.0 / 3, -0.5) + r2 * (__builtin_fma(r, 1.0 / 5, -0.25) + r2 * (-1.0 / 6));
    return __builtin_fma(r2, p, lo_) + hi_;
}

__device__ __forceinline__ double fast_exp(double x, const Tables& T)
{
    // exp x = 2^k 2^(i/128) e^r,  x = (128 k + i) ln2/128 + r,  |r| <= ln2/256.  Arguments are clamped to
    // [-1000, 710] first (0 resp. inf come out of ldexp / the final scaling, gradual underflow too; the
    // scale is applied as 2^(k-2) * 4 so that s itself never overflows before the fma); NaN is restored.
    const double xc = fmin(fmax(x, -1000.0), 710.0);
    const double kd = __builtin_rint(xc * tab::INV_LN2N);
    const int ki = (int)kd;
    double r = __builtin_fma(kd, -tab::LN2N_HI, xc);
    r = __builtin_fma(kd, -tab::LN2N_LO, r);
    const double s = ldexp(T.exp_tab[ki & (tab::N - 1)], (ki >> 7) - 2);
    const double r2 = r * r;
    // e^r - 1 = r + r^2 (1/2 + r/6) + r^4 (1/24 + r/120),  truncation r^6/720 < 6e-19
    const double tmp = r + r2 * __builtin_fma(r, 1.0 / 6, 0.5) + (r2 * r2) * __builtin_fma(r, 1.0 / 120, 1.0 / 24);
    const double res = __builtin_fma(s, tmp, s) * 4.0;
    return (x != x) ? x : res;
}

// pow(b, e) for b >= 0 (a clamped saturation distance) and a real kinetics exponent: exp(e log b).
// Relative error ~ (2 + 2.5 |e ln b|) ulp: ~1e-15 for b > 1e-3; for smaller b the value b^e itself is
// negligible against the O(1) terms it is added to.  b = 0 gives exp(-inf) = 0 for e > 0;
// the general pow(0, e) is patched in by the caller when an exponent is <= 0.
__device__ __forceinline__ double pow_sat(double b, double e, const Tables& T)
{
    return fast_exp(e * fast_log(b, T), T);
}

// Fiadeiro-Veronis weight sigma(Pe) for |Pe| >= PECLET_MIN; LHeureux_model.py:437-442 (= calculate_sigma :147-160)
__device__ __forceinline__ double fv_sigma(double Pe, double W, const Tables& T)
{
    const double a = fabs(Pe);
    double s = 0.0;
    if (a > PECLET_MAX) {
        s = (W > 0.0) ? 1.0 : ((W < 0.0) ? -1.0 : W);  // np.sign(W) incl. 0 and NaN
    } else if (!(a < PECLET_MIN)) {
        // cosh/sinh - 1/Pe  ==  1 + 2/(e^(2 Pe) - 1) - 1/Pe ;  |Pe| in [1e-2, 1e2]
        s = (1.0 + 2.0 * rcp_nr(fast_exp(2.0 * Pe, T) - 1.0)) - rcp_nr(Pe);
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

// uc/um/up: values at cell i, i-1, i+1 (ghosts already substituted).  in_mask: cell inside the
// dissolution zone.  K: hot constants (registers); C: the instance's full constant block (cold parts are
// read from memory only on rare paths).  r: the five rates (LHeureux_model.py:498-520).
__device__ __forceinline__ void rhs_point(const double (&uc)[NF], const double (&um)[NF], const double (&up)[NF],
                                          bool in_mask, const HotConsts& K, const DevConsts* __restrict__ C,
                                          const Tables& T, double (&r)[NF], PointAux& aux)
{
    const double CA = uc[0], CC = uc[1], c = uc[2], o = uc[3], Phi = uc[4];

    // ---- porosity-only quantities: F, U, W, den (:414-429).  ONE reciprocal serves 1/Phi, 1/(1-Phi), 1/den.
    const double omPhi = 1.0 - Phi;
    const double den = 1.0 - 2.0 * fast_log(Phi, T);
    const double pd = Phi * den, od = omPhi * den, po = Phi * omPhi;
    const double rall = rcp_nr(pd * omPhi);
    const double invPhi = rall * od, invom = rall * pd, invden = rall * po;
    const double F = 1.0 - fast_exp(10.0 - 10.0 * invPhi, T);
    const double Phi2 = Phi * Phi;
    const double rF = K.rhorat * F;
    const double U = K.presum + rF * (Phi2 * Phi) * invom;
    const double W = K.presum - rF * Phi2;
    aux.U = U;
    aux.W = W;

    // ---- upwinded solid gradients (:418-423)
    const bool upw = U > 0.0;
    const double CAg = (upw ? (CA - um[0]) : (up[0] - CA)) * K.inv_dx;
    const double CCg = (upw ? (CC - um[1]) : (up[1] - CC)) * K.inv_dx;

    // ---- Fiadeiro-Veronis weights (:433-462); all three vanish when every |Pe| < PECLET_MIN (fine grids)
    double s_c = 0.0, s_o = 0.0, s_p = 0.0;
    if (K.fv) {
        const double Wd = W * den;
        const double Pc = Wd * K.pe_cCa, Po = Wd * K.pe_cCO3, Pp = W * K.pe_Phi;
        if (!(fmax(fmax(fabs(Pc), fabs(Po)), fabs(Pp)) < PECLET_MIN)) {
            s_c = fv_sigma(Pc, W, T);
            s_o = fv_sigma(Po, W, T);
            s_p = fv_sigma(Pp, W, T);
        }
    }
    // weighted gradients 0.5*((1-s) forw + (1+s) back) and Laplacians (:464-469, :372-384)
    const double c_b = c - um[2], c_f = up[2] - c;
    const double o_b = o - um[3], o_f = up[3] - o;
    const double p_b = Phi - um[4], p_f = up[4] - Phi;
    const double hdx = 0.5 * K.inv_dx;
    const double cg = ((1.0 - s_c) * c_f + (1.0 + s_c) * c_b) * hdx;
    const double og = ((1.0 - s_o) * o_f + (1.0 + s_o) * o_b) * hdx;
    const double pg = ((1.0 - s_p) * p_f + (1.0 + s_p) * p_b) * hdx;
    const double c_lap = (c_f - c_b) * K.inv_dx2;
    const double o_lap = (o_f - o_b) * K.inv_dx2;
    const double p_lap = (p_f - p_b) * K.inv_dx2;

    // ---- porosity-coupled diffusion helpers (:471-477)
    const double h1 = Phi * invden;
    const double h2 = pg * (2.0 + den) * (invden * invden);
    const double Hc = K.dCa * (h2 * cg + h1 * c_lap);
    const double Ho = K.dCO3 * (h2 * og + h1 * o_lap);

    // ---- reaction terms (:479-493).  Of each clamp pair (min(x,1), max(x,1)) one power has base exactly 0
    // and the other is evaluated; with positive exponents (generic_p0 == 0) the zero-base member vanishes.
    const double O2 = c * o;
    const double O3 = O2 * K.KRat;
    const bool under = O3 < 1.0;
    const bool over = O2 > 1.0;
    double pwA = O3 - O3;  // 0, or NaN for a non-finite O3 (keeps the reference's NaN visible)
    if (!under || in_mask) pwA = pow_sat(under ? 1.0 - O3 : O3 - 1.0, under ? K.m2 : K.m1, T);
    const double pwC = pow_sat(over ? O2 - 1.0 : 1.0 - O2, over ? K.n1 : K.n2, T);
    double tA, tC;
    if (!K.generic_p0) {
        tA = under ? pwA : -K.nu1 * pwA;   // (1-O3)^m2 * mask   |  -nu1 (O3-1)^m1
        tC = over ? pwC : -K.nu2 * pwC;    // (O2-1)^n1          |  -nu2 (1-O2)^n2
    } else {
        const double z1 = C->p0_m1, z2 = C->p0_m2, y1 = C->p0_n1, y2 = C->p0_n2;
        const double mask = in_mask ? 1.0 : 0.0;
        const double a_under = ((under ? 1.0 - O3 : 1.0) == 0.0) ? z2 : pwA;   // pow(0, m2) when O3 == 1 exactly
        const double a_over = ((under ? 0.0 : O3 - 1.0) == 0.0) ? z1 : pwA;
        tA = under ? (in_mask ? a_under : pwA) * mask - K.nu1 * z1 : z2 * mask - K.nu1 * a_over;
        const double c_over = ((over ? O2 - 1.0 : 0.0) == 0.0) ? y1 : pwC;
        const double c_under = ((over ? 1.0 : 1.0 - O2) == 0.0) ? y2 : pwC;
        tC = over ? c_over - K.nu2 * y2 : y1 - K.nu2 * c_under;
    }
    const double coA = CA * tA;
    const double coC = CC * tC;
    const double R = coA - K.lambda_ * coC;

    const double dWdx = -K.rhorat * pg * (2.0 * Phi * F + 10.0 * (F - 1.0));  // :495
    const double DaR = K.Da * omPhi * R;

    r[0] = -U * CAg - K.Da * ((1.0 - CA) * coA + K.lambda_ * CA * coC);           // :498-499
    r[1] = -U * CCg + K.Da * (K.lambda_ * (1.0 - CC) * coC + CC * coA);           // :502-503
    r[2] = (Hc + DaR * (K.delta - c)) * invPhi - W * cg;                          // :506-509
    r[3] = (Ho + DaR * (K.delta - o)) * invPhi - W * og;                          // :512-515
    r[4] = -(dWdx * Phi + W * pg) + K.dPhi * p_lap + DaR;                         // :518-520
}

}  // namespace marl
