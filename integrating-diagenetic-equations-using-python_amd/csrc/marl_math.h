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
//   * pow(b, e) for the kinetics exponents is exp(e*log b) (b in [0, ~1]; error analysis in DESIGN.md);
//   * coth(Pe) - 1/Pe is 1 + 2/expm1(2 Pe) - 1/Pe and only taken in the mid Peclet range; the
//     branch is wave-uniform in practice (hard part 1c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace marl {

constexpr int NF = 5;  // CA, CC, cCa, cCO3, Phi  (marlpde/Evolve_scenario.py:76-86)

// Uniform (per model instance) constants, derived on the host in marl_api.hip:derive_consts()
// from marl_params exactly as the reference's constructor does (LHeureux_model.py:36-72,130-133).
struct DevConsts {
    double inv_dx, inv_dx2;          // 1/dx, 1/dx^2, dx = length/N (py-pde grid.discretization)
    double pe_cCa, pe_cCO3, pe_Phi;  // delta_x/(2 dCa), delta_x/(2 dCO3), delta_x/(2 dPhi_fixed)  (:436,444,452)
    double presum, rhorat;           // :71-72, :69-70
    double KRat, nu1, nu2;           // :51, :36, :39
    double m1, m2, n1, n2;           // kinetics exponents
    double p0_m1, p0_m2, p0_n1, p0_n2;  // pow(0, exponent): value of the clamp-pair member whose base is exactly 0
    double lambda_, Da, delta;       // :64, :63, :62
    double dCa, dCO3, dPhi;          // :60, :61, :132-133 (dPhi_fixed)
    double bc[NF];                   // Dirichlet values at x = 0  (:26-30)
    double pe_min, pe_max;           // :87-88
    int64_t N;                       // cells of the (global) grid
    int64_t mask_lo, mask_hi;        // cells [mask_lo, mask_hi) have not_too_shallow*not_too_deep == 1 (Evolve_scenario.py:51-54)
    int32_t fv;                      // FV_switch
    int32_t pad;
};

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

// pow(b, e) for b >= 0 (a clamped saturation distance) and a real kinetics exponent.
__device__ __forceinline__ double pow_sat(double b, double e, double pow0)
{
    double r = exp(e * log(b));
    return (b == 0.0) ? pow0 : r;
}

// Fiadeiro-Veronis weight sigma(Pe); LHeureux_model.py:437-442 (= calculate_sigma :147-160)
__device__ __forceinline__ double fv_sigma(double Pe, double W, double pe_min, double pe_max)
{
    const double a = fabs(Pe);
    double s = 0.0;
    if (a > pe_max) {
        s = (W > 0.0) ? 1.0 : ((W < 0.0) ? -1.0 : W);  // np.sign(W) incl. 0 and NaN
    } else if (!(a < pe_min)) {
        // cosh/sinh - 1/Pe  ==  1 + 2/expm1(2 Pe) - 1/Pe
        s = (1.0 + 2.0 * rcp_nr(expm1(2.0 * Pe))) - rcp_nr(Pe);
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
// dissolution zone.  r: the five rates (LHeureux_model.py:498-520).
__device__ __forceinline__ void rhs_point(const double (&uc)[NF], const double (&um)[NF], const double (&up)[NF],
                                          bool in_mask, const DevConsts& C, double (&r)[NF], PointAux& aux)
{
    const double CA = uc[0], CC = uc[1], c = uc[2], o = uc[3], Phi = uc[4];

    // ---- porosity-only quantities: F, U, W, den (:414-429)
    const double invPhi = rcp_nr(Phi);
    const double F = 1.0 - exp(10.0 - 10.0 * invPhi);
    const double omPhi = 1.0 - Phi;
    const double Phi2 = Phi * Phi;
    const double rF = C.rhorat * F;
    const double U = C.presum + rF * (Phi2 * Phi) * rcp_nr(omPhi);
    const double W = C.presum - rF * Phi2;
    const double den = 1.0 - 2.0 * log(Phi);
    const double invden = rcp_nr(den);
    aux.U = U;
    aux.W = W;

    // ---- upwinded solid gradients (:418-423)
    const bool upw = U > 0.0;
    const double CAg = (upw ? (CA - um[0]) : (up[0] - CA)) * C.inv_dx;
    const double CCg = (upw ? (CC - um[1]) : (up[1] - CC)) * C.inv_dx;

    // ---- Fiadeiro-Veronis weights (:433-462)
    double s_c = 0.0, s_o = 0.0, s_p = 0.0;
    if (C.fv) {
        const double Wd = W * den;
        s_c = fv_sigma(Wd * C.pe_cCa, W, C.pe_min, C.pe_max);
        s_o = fv_sigma(Wd * C.pe_cCO3, W, C.pe_min, C.pe_max);
        s_p = fv_sigma(W * C.pe_Phi, W, C.pe_min, C.pe_max);
    }
    // weighted gradients 0.5*((1-s) forw + (1+s) back) and Laplacians (:464-469, :372-384)
    const double c_b = c - um[2], c_f = up[2] - c;
    const double o_b = o - um[3], o_f = up[3] - o;
    const double p_b = Phi - um[4], p_f = up[4] - Phi;
    const double hdx = 0.5 * C.inv_dx;
    const double cg = ((1.0 - s_c) * c_f + (1.0 + s_c) * c_b) * hdx;
    const double og = ((1.0 - s_o) * o_f + (1.0 + s_o) * o_b) * hdx;
    const double pg = ((1.0 - s_p) * p_f + (1.0 + s_p) * p_b) * hdx;
    const double c_lap = (c_f - c_b) * C.inv_dx2;
    const double o_lap = (o_f - o_b) * C.inv_dx2;
    const double p_lap = (p_f - p_b) * C.inv_dx2;

    // ---- porosity-coupled diffusion helpers (:471-477)
    const double h1 = Phi * invden;
    const double h2 = pg * (2.0 + den) * (invden * invden);
    const double Hc = C.dCa * (h2 * cg + h1 * c_lap);
    const double Ho = C.dCO3 * (h2 * og + h1 * o_lap);

    // ---- reaction terms (:479-493); one pow per clamp pair, see header
    const double O2 = c * o;
    const double O3 = O2 * C.KRat;
    double tA;
    {
        const bool under = O3 < 1.0;
        if (under && !in_mask) {
            // (1-O3)^m2 * 0 - nu1 * 0^m1 ; keep NaN/Inf of O3 visible
            tA = (O3 - O3) - C.nu1 * C.p0_m1;
        } else {
            const double base = under ? 1.0 - O3 : O3 - 1.0;
            const double pw = pow_sat(base, under ? C.m2 : C.m1, under ? C.p0_m2 : C.p0_m1);
            tA = under ? pw - C.nu1 * C.p0_m1 : (in_mask ? C.p0_m2 : 0.0) - C.nu1 * pw;
        }
    }
    double tC;
    {
        const bool over = O2 > 1.0;
        const double base = over ? O2 - 1.0 : 1.0 - O2;
        const double pw = pow_sat(base, over ? C.n1 : C.n2, over ? C.p0_n1 : C.p0_n2);
        tC = over ? pw - C.nu2 * C.p0_n2 : C.p0_n1 - C.nu2 * pw;
    }
    const double coA = CA * tA;
    const double coC = CC * tC;
    const double R = coA - C.lambda_ * coC;

    const double dWdx = -C.rhorat * pg * (2.0 * Phi * F + 10.0 * (F - 1.0));  // :495
    const double DaR = C.Da * omPhi * R;

    r[0] = -U * CAg - C.Da * ((1.0 - CA) * coA + C.lambda_ * CA * coC);           // :498-499
    r[1] = -U * CCg + C.Da * (C.lambda_ * (1.0 - CC) * coC + CC * coA);           // :502-503
    r[2] = (Hc + DaR * (C.delta - c)) * invPhi - W * cg;                          // :506-509
    r[3] = (Ho + DaR * (C.delta - o)) * invPhi - W * og;                          // :512-515
    r[4] = -(dWdx * Phi + W * pg) + C.dPhi * p_lap + DaR;                         // :518-520
}

}  // namespace marl
