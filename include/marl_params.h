/* marl_params.h - plain-old-data parameter block shared by the C-ABI (marl_hip.h) and the
 * CPU oracle (oracle/marl_oracle.c).
 *
 * One marl_params describes ONE model instance.  Its members are, name for name, the keyword
 * arguments of the reference's model constructor
 *     LMAHeureuxPorosityDiff.__init__            marlpde/LHeureux_model.py:12-16
 * (filtered out of the Scenario dict by signature at marlpde/Evolve_scenario.py:59-62) with the
 * three py-pde objects the constructor also receives (grid `Depths`, masks `not_too_shallow`,
 * `not_too_deep`; marlpde/Evolve_scenario.py:40,51-54) replaced by the three numbers that define
 * them.  `slices_all_fields` is implied by N (field-major state, Evolve_scenario.py:64-65).
 */
#ifndef MARL_PARAMS_H
#define MARL_PARAMS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct marl_params {
    /* boundary values at x = 0 (Dirichlet for all five fields, LHeureux_model.py:26-30) */
    double CA0, CC0, cCa0, cCO30, Phi0;
    double sedimentationrate, Xstar, Tstar;
    double k1, k2, k3, k4;
    double m1, m2, n1, n2;
    double b, beta, rhos, rhow, rhos0;
    double KA, KC, muA, D0Ca;
    double PhiNR, PhiInfty, PhiIni;
    double DCa, DCO3;
    /* grid [0, length] with N cell-centred nodes: length = max_depth / Xstar (Evolve_scenario.py:40) */
    double length;
    /* dissolution-zone mask H(x - shallow_limit) * H(deep_limit - x), H(0) = 0, in units of Xstar
     * (ShallowLimit/Xstar, DeepLimit/Xstar; Evolve_scenario.py:51-54) */
    double shallow_limit, deep_limit;
    /* 1 = Fiadeiro-Veronis weighting of the solute/porosity gradients, 0 = central (LHeureux_model.py:433-462) */
    int32_t FV_switch;
    /* 0 = the reference's fixed porosity diffusion coefficient dPhi_fixed (LHeureux_model.py:124-133, :431);
     * 1 = the time-varying coefficient dPhi = auxcon F Phi^3 / (1 - Phi) of the lines the reference keeps commented
     *     out (LHeureux_model.py:222-223, :430): enters the porosity Peclet number (:452) and dPhi * Phi_laplace (:519) */
    int32_t dPhi_variable;
} marl_params;

#define MARL_NFIELDS 5 /* CA, CC, cCa, cCO3, Phi - in this order (Evolve_scenario.py:76-86) */
#define MARL_NEVENTS 7 /* zeros, zeros_CA, zeros_CC, ones_CA_plus_CC, ones_Phi, zeros_U, zeros_W (Evolve_scenario.py:107-109) */

/* Integration statistics; the comparable part of scipy's OdeResult (nfev, status, t_events). */
typedef struct marl_stats {
    int64_t nfev;       /* RHS evaluations, counted as scipy does (RK45: 1 at start + 6 per attempt; Radau: incl. the
                           finite-difference Jacobian columns) */
    int64_t n_accepted; /* accepted steps */
    int64_t n_rejected; /* rejected attempts */
    int32_t status;     /* 0 reached t1; -1 step size too small (scipy status -1); 2 attempt budget exhausted */
    int32_t reserved;
    double t;           /* time reached */
    double h_next;      /* step size the controller would try next */
    double event_value[MARL_NEVENTS];   /* the 7 monitors evaluated at the final state */
    int64_t n_events[MARL_NEVENTS];     /* sign changes seen per monitor (both directions, non-terminal) */
    int64_t njev;       /* Jacobian evaluations (implicit Radau path; 0 for the explicit integrators) - scipy's sol.njev */
    int64_t nlu;        /* LU decompositions, counted as scipy does (real and complex system separately) - sol.nlu */
} marl_stats;

#ifdef __cplusplus
}
#endif
#endif /* MARL_PARAMS_H */
