/* marl_oracle.c - CPU restatement of the reference's five-field RHS and explicit RK time loops.
 *
 * TEST INFRASTRUCTURE.  This file is the parity oracle and the "port" CPU baseline.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * path (libmarl_hip.so) never links, loads or calls anything in oracle/.
 *
 * Pinning: checked against tests/golden/ (vectors produced by the reference's own code, see
 * oracle/make_goldens.py) and against the reference's three HDF5 regression goldens
 * (tests/test_oracle_*.py).  The py-pde stencil/ghost-cell layer and the scipy controller are
 * third-party code that is not under /root/reference; their published algorithms are
 * restated here (py-pde 0.32.2, scipy 1.11.2 pinned by the reference's poetry.lock).
 *
 * Every function cites the reference lines it follows.  Arithmetic is written in the same
 * operation order as the reference loop so that differences stay at the libm level.
 * Compile with -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/marl_params.h"

#define NF MARL_NFIELDS

/* ------------------------------------------------------------------------------------------
 * Derived constants.  marlpde/LHeureux_model.py:23-24 (delta_x), :36-72, :87-88, :130-133.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double dx;       /* grid.discretization = length / N: used by the stencils (py-pde) */
    double delta_x;  /* x[1] - x[0] of the cell centres: used by the Peclet numbers (:23-24) */
    double nu1, nu2, KRat, dCa, dCO3, delta, Da, lambda_, auxcon, rhorat0, rhorat, presum;
    double F_fixed, dPhi_fixed, Peclet_min, Peclet_max;
} orc_consts;

static void orc_derive(const marl_params *p, int64_t N, orc_consts *c)
{
    const double g = 100 * 9.81;                                              /* :59 */
    c->dx = p->length / (double)N;
    c->delta_x = (0.0 + 1.5 * c->dx) - (0.0 + 0.5 * c->dx);                    /* :23-24 */
    c->nu1 = p->k1 / p->k2;                                                    /* :36 */
    c->nu2 = p->k4 / p->k3;                                                    /* :39 */
    c->KRat = p->KC / p->KA;                                                   /* :51 */
    c->dCa = p->DCa / p->D0Ca;                                                 /* :60 */
    c->dCO3 = p->DCO3 / p->D0Ca;                                               /* :61 */
    c->delta = p->rhos / (p->muA * sqrt(p->KC));                               /* :62 */
    c->Da = p->k2 * p->Tstar;                                                  /* :63 */
    c->lambda_ = p->k3 / p->k2;                                                /* :64 */
    c->auxcon = p->beta / (p->D0Ca * p->b * g * p->rhow * (p->PhiNR - p->PhiInfty)); /* :65-66 */
    c->rhorat0 = (p->rhos0 / p->rhow - 1) * p->beta / p->sedimentationrate;    /* :67-68 */
    c->rhorat = (p->rhos / p->rhow - 1) * p->beta / p->sedimentationrate;      /* :69-70 */
    c->presum = 1 - c->rhorat0 * pow(p->Phi0, 3) * (1 - exp(10 - 10 / p->Phi0)) / (1 - p->Phi0); /* :71-72 */
    c->Peclet_min = 1e-2;                                                      /* :87 */
    c->Peclet_max = 1 / c->Peclet_min;                                         /* :88 */
    c->F_fixed = 1 - exp(10 - 10 / p->PhiIni);                                 /* :131 */
    c->dPhi_fixed = c->auxcon * c->F_fixed * pow(p->PhiIni, 3) / (1 - p->PhiIni); /* :132-133 */
}

/* Exported so tests can compare the derived constants with tests/golden/derived_constants.json. */
void marl_oracle_derive(const marl_params *p, int64_t N, double out[17])
{
    orc_consts c;
    orc_derive(p, N, &c);
    const double v[17] = {c.delta_x, c.nu1, c.nu2, c.KRat, c.dCa, c.dCO3, c.delta, c.Da, c.lambda_, c.auxcon,
                          c.rhorat0, c.rhorat, c.presum, c.F_fixed, c.dPhi_fixed, c.Peclet_min, c.Peclet_max};
    memcpy(out, v, sizeof v);
}

/* Depth mask not_too_shallow * not_too_deep, marlpde/Evolve_scenario.py:51-54: Heaviside with
 * H(0) = 0 evaluated at the cell centres (i + 1/2) dx. */
static inline double orc_mask(const marl_params *p, double dx, int64_t i)
{
    const double x = 0.0 + ((double)i + 0.5) * dx;
    const double shallow = (x - p->shallow_limit) > 0 ? 1.0 : 0.0;
    const double deep = (p->deep_limit - x) > 0 ? 1.0 : 0.0;
    return deep * shallow;
}

/* Fiadeiro-Veronis weight, marlpde/LHeureux_model.py:437-442 (= calculate_sigma :147-160). */
static inline double orc_sigma(double Pe, double W, double Pe_min, double Pe_max)
{
    if (fabs(Pe) < Pe_min) return 0.0;
    if (fabs(Pe) > Pe_max) return (W > 0) - (W < 0) + (W != W ? W : 0.0); /* np.sign, NaN -> NaN */
    return cosh(Pe) / sinh(Pe) - 1 / Pe;
}

/* ------------------------------------------------------------------------------------------
 * The RHS.  marlpde/LHeureux_model.py:361-522 (pde_rhs); ghost cells / stencils: py-pde
 * semantics, SURVEY.md App. B (call sites LHeureux_model.py:26-30, 96-111, 372-384).
 *   y, rate: field-major float64[5N] (Evolve_scenario.py:64-65).
 * ---------------------------------------------------------------------------------------- */
static void orc_rhs(const marl_params *p, const orc_consts *c, int64_t N, const double *y, double *rate)
{
    const double *CA = y, *CC = y + N, *cCa = y + 2 * N, *cCO3 = y + 3 * N, *Phi = y + 4 * N;
    const double dx = c->dx;
    const double lap_scale = pow(dx, -2);
    const double bc0[NF] = {p->CA0, p->CC0, p->cCa0, p->cCO30, p->Phi0};
    const double *fld[NF] = {CA, CC, cCa, cCO3, Phi};

#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t i = 0; i < N; i++) {
        /* neighbours incl. the two virtual cells: lower {"value": v} -> 2v - u[0];
         * upper {"curvature": 0} (CA, CC) -> 2u[N-1] - u[N-2]; upper {"derivative": 0} -> u[N-1] */
        double um[NF], up[NF];
        for (int f = 0; f < NF; f++) {
            um[f] = (i > 0) ? fld[f][i - 1] : 2.0 * bc0[f] - fld[f][0];
            if (i < N - 1)
                up[f] = fld[f][i + 1];
            else if (f < 2)
                up[f] = 0.0 * dx * dx + 2.0 * fld[f][N - 1] - fld[f][N - 2];
            else
                up[f] = fld[f][N - 1] + dx * 0.0;
        }
        /* :372-384 */
        const double CA_grad_back = (CA[i] - um[0]) / dx, CA_grad_forw = (up[0] - CA[i]) / dx;
        const double CC_grad_back = (CC[i] - um[1]) / dx, CC_grad_forw = (up[1] - CC[i]) / dx;
        const double cCa_grad_back = (cCa[i] - um[2]) / dx, cCa_grad_forw = (up[2] - cCa[i]) / dx;
        const double cCa_laplace = (um[2] - 2 * cCa[i] + up[2]) * lap_scale;
        const double cCO3_grad_back = (cCO3[i] - um[3]) / dx, cCO3_grad_forw = (up[3] - cCO3[i]) / dx;
        const double cCO3_laplace = (um[3] - 2 * cCO3[i] + up[3]) * lap_scale;
        const double Phi_grad_back = (Phi[i] - um[4]) / dx, Phi_grad_forw = (up[4] - Phi[i]) / dx;
        const double Phi_laplace = (um[4] - 2 * Phi[i] + up[4]) * lap_scale;

        const double F = 1 - exp(10 - 10 / Phi[i]);                                         /* :414 */
        const double U = c->presum + c->rhorat * pow(Phi[i], 3) * F / (1 - Phi[i]);         /* :416 */
        double CA_grad, CC_grad;
        if (U > 0) { CA_grad = CA_grad_back; CC_grad = CC_grad_back; }                      /* :418-423 */
        else       { CA_grad = CA_grad_forw; CC_grad = CC_grad_forw; }
        const double W = c->presum - c->rhorat * pow(Phi[i], 2) * F;                        /* :425 */
        const double denominator = 1 - 2 * log(Phi[i]);                                     /* :428 */
        const double one_minus_Phi = 1 - Phi[i];                                            /* :429 */
        const double dPhi = p->dPhi_variable ? c->auxcon * F * pow(Phi[i], 3) / one_minus_Phi     /* :430 (commented out there) */
                                             : c->dPhi_fixed;                               /* :431 */

        double sigma_cCa = 0, sigma_cCO3 = 0, sigma_Phi = 0;
        if (p->FV_switch) {                                                                 /* :433-458 */
            const double Peclet_cCa = W * c->delta_x * denominator / (2. * c->dCa);
            sigma_cCa = orc_sigma(Peclet_cCa, W, c->Peclet_min, c->Peclet_max);
            const double Peclet_cCO3 = W * c->delta_x * denominator / (2. * c->dCO3);
            sigma_cCO3 = orc_sigma(Peclet_cCO3, W, c->Peclet_min, c->Peclet_max);
            const double Peclet_Phi = W * c->delta_x / (2. * dPhi);
            sigma_Phi = orc_sigma(Peclet_Phi, W, c->Peclet_min, c->Peclet_max);
        }
        const double cCa_grad = 0.5 * ((1 - sigma_cCa) * cCa_grad_forw + (1 + sigma_cCa) * cCa_grad_back);     /* :464 */
        const double cCO3_grad = 0.5 * ((1 - sigma_cCO3) * cCO3_grad_forw + (1 + sigma_cCO3) * cCO3_grad_back); /* :466 */
        const double Phi_grad = 0.5 * ((1 - sigma_Phi) * Phi_grad_forw + (1 + sigma_Phi) * Phi_grad_back);     /* :468 */

        const double common_helper1 = Phi[i] / denominator;                                 /* :471 */
        const double common_helper2 = Phi_grad * (2 + denominator) / pow(denominator, 2);   /* :472-473 */
        const double helper_cCa_grad = c->dCa * (common_helper2 * cCa_grad + common_helper1 * cCa_laplace);    /* :474 */
        const double helper_cCO3_grad = c->dCO3 * (common_helper2 * cCO3_grad + common_helper1 * cCO3_laplace); /* :476 */

        const double two_factors = cCa[i] * cCO3[i];                                        /* :479 */
        const double two_factors_upp_lim = (1.0 < two_factors) ? 1.0 : two_factors;         /* min(x,1) :480 */
        const double two_factors_low_lim = (1.0 > two_factors) ? 1.0 : two_factors;         /* max(x,1) :481 */
        const double three_factors = two_factors * c->KRat;                                 /* :482 */
        const double three_factors_upp_lim = (1.0 < three_factors) ? 1.0 : three_factors;
        const double three_factors_low_lim = (1.0 > three_factors) ? 1.0 : three_factors;

        const double coA = CA[i] * ((pow(1 - three_factors_upp_lim, p->m2)) * orc_mask(p, dx, i)
                                    - c->nu1 * pow(three_factors_low_lim - 1, p->m1));       /* :486-488 */
        const double coC = CC[i] * ((pow(two_factors_low_lim - 1, p->n1))
                                    - c->nu2 * pow(1 - two_factors_upp_lim, p->n2));         /* :490-491 */
        const double common_helper3 = coA - c->lambda_ * coC;                               /* :493 */
        const double dW_dx = -c->rhorat * Phi_grad * (2 * Phi[i] * F + 10 * (F - 1));       /* :495 */

        rate[i] = -U * CA_grad - c->Da * ((1 - CA[i]) * coA + c->lambda_ * CA[i] * coC);    /* :498-499 */
        rate[N + i] = -U * CC_grad + c->Da * (c->lambda_ * (1 - CC[i]) * coC + CC[i] * coA); /* :502-503 */
        rate[2 * N + i] = helper_cCa_grad / Phi[i] - W * cCa_grad
                          + c->Da * one_minus_Phi * (c->delta - cCa[i]) * common_helper3 / Phi[i];   /* :506-509 */
        rate[3 * N + i] = helper_cCO3_grad / Phi[i] - W * cCO3_grad
                          + c->Da * one_minus_Phi * (c->delta - cCO3[i]) * common_helper3 / Phi[i];  /* :512-515 */
        rate[4 * N + i] = -(dW_dx * Phi[i] + W * Phi_grad) + dPhi * Phi_laplace
                          + c->Da * one_minus_Phi * common_helper3;                         /* :518-520 */
    }
}

void marl_oracle_rhs(const marl_params *p, int64_t N, const double *y, double *rate)
{
    orc_consts c;
    orc_derive(p, N, &c);
    orc_rhs(p, &c, N, y, rate);
}

/* ------------------------------------------------------------------------------------------
 * The seven monitors.  marlpde/LHeureux_model.py:524-593.  NaN propagates like np.amin/amax.
 * ---------------------------------------------------------------------------------------- */
static void orc_events(const marl_params *p, const orc_consts *c, int64_t N, const double *y, double out[MARL_NEVENTS])
{
    (void)p;
    double mn_all = INFINITY, mn_CA = INFINITY, mn_CC = INFINITY, mx_sum = -INFINITY, mx_Phi = -INFINITY;
    double mn_U = INFINITY, mx_W = -INFINITY;
    int nan_all = 0, nan_CA = 0, nan_CC = 0, nan_sum = 0, nan_Phi = 0, nan_U = 0, nan_W = 0;
    for (int64_t i = 0; i < NF * N; i++) {
        if (y[i] != y[i]) nan_all = 1;
        if (y[i] < mn_all) mn_all = y[i];
    }
    for (int64_t i = 0; i < N; i++) {
        const double CA = y[i], CC = y[N + i], Phi = y[4 * N + i];
        const double s = CA + CC;
        const double F = 1 - exp(10 - 10 / Phi);                                  /* :574, :588 */
        const double U = c->presum + c->rhorat * pow(Phi, 3) * F / (1 - Phi);     /* :575 */
        const double W = c->presum - c->rhorat * pow(Phi, 2) * F;                 /* :589 */
        if (CA != CA) nan_CA = 1;
        if (CC != CC) nan_CC = 1;
        if (s != s) nan_sum = 1;
        if (Phi != Phi) nan_Phi = 1;
        if (U != U) nan_U = 1;
        if (W != W) nan_W = 1;
        if (CA < mn_CA) mn_CA = CA;
        if (CC < mn_CC) mn_CC = CC;
        if (s > mx_sum) mx_sum = s;
        if (Phi > mx_Phi) mx_Phi = Phi;
        if (U < mn_U) mn_U = U;
        if (W > mx_W) mx_W = W;
    }
    out[0] = nan_all ? NAN : mn_all;          /* zeros            :530 */
    out[1] = nan_CA ? NAN : mn_CA;            /* zeros_CA         :538 */
    out[2] = nan_CC ? NAN : mn_CC;            /* zeros_CC         :546 */
    out[3] = nan_sum ? NAN : mx_sum - 1;      /* ones_CA_plus_CC  :556 */
    out[4] = nan_Phi ? NAN : mx_Phi - 1;      /* ones_Phi         :565 */
    out[5] = nan_U ? NAN : mn_U;              /* zeros_U          :579 */
    out[6] = nan_W ? NAN : mx_W;              /* zeros_W          :593 */
}

void marl_oracle_events(const marl_params *p, int64_t N, const double *y, double out[MARL_NEVENTS])
{
    orc_consts c;
    orc_derive(p, N, &c);
    orc_events(p, &c, N, y, out);
}

/* ------------------------------------------------------------------------------------------
 * Classical fixed-step RK4 (BASELINE config 2; not in the reference - README.md:9 only notes
 * that forward Euler fails).  Stage and update arithmetic is the contract the HIP kernel follows:
 *   k1 = f(y); k2 = f(y + (dt/2) k1); k3 = f(y + (dt/2) k2); k4 = f(y + dt k3)
 *   y <- y + (dt/6) * (((k1 + 2 k2) + 2 k3) + k4)
 * ---------------------------------------------------------------------------------------- */
int marl_oracle_rk4(const marl_params *p, int64_t N, double *y, double dt, int64_t nsteps)
{
    orc_consts c;
    orc_derive(p, N, &c);
    const int64_t n = NF * N;
    double *buf = (double *)malloc(sizeof(double) * n * 3);
    if (!buf) return -1;
    double *k = buf, *acc = buf + n, *ys = buf + 2 * n;
    const double h2 = 0.5 * dt, h6 = dt / 6.0;
#ifdef _OPENMP
#define ORC_PAR _Pragma("omp parallel for schedule(static)")
#else
#define ORC_PAR
#endif
    for (int64_t s = 0; s < nsteps; s++) {
        orc_rhs(p, &c, N, y, k);
        ORC_PAR for (int64_t i = 0; i < n; i++) { acc[i] = k[i]; ys[i] = y[i] + h2 * k[i]; }
        orc_rhs(p, &c, N, ys, k);
        ORC_PAR for (int64_t i = 0; i < n; i++) { acc[i] = acc[i] + 2.0 * k[i]; ys[i] = y[i] + h2 * k[i]; }
        orc_rhs(p, &c, N, ys, k);
        ORC_PAR for (int64_t i = 0; i < n; i++) { acc[i] = acc[i] + 2.0 * k[i]; ys[i] = y[i] + dt * k[i]; }
        orc_rhs(p, &c, N, ys, k);
        ORC_PAR for (int64_t i = 0; i < n; i++) y[i] = y[i] + h6 * (acc[i] + k[i]);
    }
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Adaptive Dormand-Prince 5(4) exactly as scipy drives it for the reference
 * (call site marlpde/Evolve_scenario.py:104-109; algorithm scipy/integrate/_ivp/rk.py:14-71
 * rk_step, :111-176 _step_impl, :377-407 tableau + dense output P, common.py:63-65 RMS norm,
 * ivp.py:654-723 driver incl. events and t_eval).
 * ---------------------------------------------------------------------------------------- */
static const double DP_C[6] __attribute__((unused)) = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1};
static const double DP_A[6][5] = {
    {0, 0, 0, 0, 0},
    {1.0 / 5, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
static const double DP_B[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
static const double DP_E[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
static const double DP_P[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

#define DP_SAFETY 0.9
#define DP_MIN_FACTOR 0.2
#define DP_MAX_FACTOR 10.0

typedef struct {
    const marl_params *p;
    const orc_consts *c;
    int64_t N;
    double t_old, h;
    const double *y_old;
    const double *K; /* 7 x 5N (Dormand-Prince stages), or NULL */
    double *scratch; /* 5N */
    const double *Q; /* Radau: 5N x 3 (RadauDenseOutput, radau.py:549-572), or NULL */
    const double *D; /* BDF: (order + 1) x 5N differences (BdfDenseOutput, bdf.py:456-478), or NULL; then t_old / h hold t / h */
    int order;
} orc_dense;

/* RkDenseOutput._call_impl, rk.py:560-574:  y(t) = y_old + h * Q . [x, x^2, x^3, x^4],  Q = K^T P */
static void orc_dense_eval(const orc_dense *d, double t, double *out)
{
    const int64_t n = NF * d->N;
    if (d->D) { /* BdfDenseOutput._call_impl: x_j = (t - (t_end - h j)) / (h (1 + j)), p = cumprod(x), y = D[0] + D[1:].T . p */
        double pr[8], acc = 1;
        for (int j = 0; j < d->order; j++) {
            acc *= (t - (d->t_old - d->h * j)) / (d->h * (1 + j));
            pr[j] = acc;
        }
        for (int64_t i = 0; i < n; i++) {
            double a = 0;
            for (int j = 0; j < d->order; j++) a += d->D[(int64_t)(j + 1) * n + i] * pr[j];
            out[i] = a + d->D[i];
        }
        return;
    }
    const double x = (t - d->t_old) / d->h;
    if (d->Q) { /* RadauDenseOutput._call_impl, radau.py:557-572: y = Q . [x, x^2, x^3] + y_old  (not multiplied by h) */
        const double p1 = x, p2 = p1 * x, p3 = p2 * x; /* np.cumprod */
        for (int64_t i = 0; i < n; i++)
            out[i] = ((d->Q[3 * i] * p1 + d->Q[3 * i + 1] * p2) + d->Q[3 * i + 2] * p3) + d->y_old[i];
        return;
    }
    double pw[4];
    pw[0] = x;
    for (int m = 1; m < 4; m++) pw[m] = pw[m - 1] * x;
    for (int64_t i = 0; i < n; i++) {
        double acc = 0;
        for (int m = 0; m < 4; m++) {
            double q = 0;
            for (int s = 0; s < 7; s++) q += d->K[s * n + i] * DP_P[s][m];
            acc += q * pw[m];
        }
        out[i] = d->h * acc + d->y_old[i];
    }
}

static double orc_event_at(const orc_dense *d, int which, double t)
{
    double g[MARL_NEVENTS];
    orc_dense_eval(d, t, d->scratch);
    orc_events(d->p, d->c, d->N, d->scratch, g);
    return g[which];
}

/* Brent's method on [a, b] with xtol = rtol = 4 eps (ivp.py:51-76 -> scipy.optimize.brentq). */
static double orc_brent(const orc_dense *d, int which, double a, double b)
{
    const double xtol = 4 * 2.220446049250313e-16, rtol = xtol;
    double fa = orc_event_at(d, which, a), fb = orc_event_at(d, which, b);
    if (fa == 0) return a;
    if (fb == 0) return b;
    double xpre = a, xcur = b, fpre = fa, fcur = fb, xblk = 0, fblk = 0, spre = 0, scur = 0;
    for (int it = 0; it < 100; it++) {
        if (fpre != 0 && fcur != 0 && ((fpre < 0) != (fcur < 0))) {
            xblk = xpre; fblk = fpre; spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        const double delta = (xtol + rtol * fabs(xcur)) / 2;
        const double sbis = (xblk - xcur) / 2;
        if (fcur == 0 || fabs(sbis) < delta) return xcur;
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);
            } else {
                const double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * fabs(stry) < fmin(fabs(spre), 3 * fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else { spre = sbis; scur = sbis; }
        xpre = xcur; fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0 ? delta : -delta);
        fcur = orc_event_at(d, which, xcur);
    }
    return xcur;
}

/* y: in = y(t0), out = state at the time reached.  t_eval (sorted, within [t0,t1]) may be NULL.
 * y_eval: n_eval x 5N row-major (sample-major).  step_times: optional buffer receiving the
 * accepted step end times (capacity max_steps_out; *n_steps_out entries written).
 * t_events: optional 7 x max_events buffer of root times.  Returns stats->status. */
int marl_oracle_rk45(const marl_params *p, int64_t N, double *y, double t0, double t1, double first_step,
                     double rtol, double atol, const double *t_eval, int64_t n_eval, double *y_eval,
                     double *step_times, int64_t max_steps_out, int64_t *n_steps_out,
                     double *t_events, int64_t max_events, int64_t max_attempts, marl_stats *st)
{
    orc_consts c;
    orc_derive(p, N, &c);
    const int64_t n = NF * N;
    double *buf = (double *)malloc(sizeof(double) * n * 11);
    if (!buf) return -2;
    double *K = buf, *ynew = buf + 7 * n, *ys = buf + 8 * n, *yold = buf + 9 * n, *scratch = buf + 10 * n;
    double g[MARL_NEVENTS], g_new[MARL_NEVENTS];
    memset(st, 0, sizeof *st);
    int64_t steps_out = 0, eval_i = 0, attempts = 0;

    if (rtol < 100 * 2.220446049250313e-16) rtol = 100 * 2.220446049250313e-16;   /* validate_tol, common.py:44-51 */
    double t = t0, h_abs = first_step;                   /* rk.py:94-100 (first_step validated) */
    orc_rhs(p, &c, N, y, K);                             /* self.f = fun(t0, y0) */
    st->nfev = 1;
    orc_events(p, &c, N, y, g);                          /* ivp.py:645 g = [event(t0, y0)] */
    int status = 1;                                      /* 1 = running (internal) */

    while (status == 1) {
        if (t == t1) { status = 0; break; }              /* base.py:189-194 */
        /* ---- _step_impl, rk.py:111-176 ---- */
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        if (h_abs < min_step) h_abs = min_step;          /* max_step = inf */
        int accepted = 0, rejected = 0;
        double h = 0, t_new = t;
        while (!accepted) {
            if (h_abs < min_step) { status = -1; break; }
            if (max_attempts > 0 && attempts >= max_attempts) { status = 2; break; }
            attempts++;
            h = h_abs;
            t_new = t + h;
            if (t_new - t1 > 0) t_new = t1;
            h = t_new - t;
            h_abs = fabs(h);
            /* rk_step, rk.py:61-69; K[0] = f is already in place (FSAL) */
            for (int s = 1; s < 6; s++) {
                for (int64_t i = 0; i < n; i++) {
                    double dot = 0;
                    for (int j = 0; j < s; j++) dot += K[j * n + i] * DP_A[s][j];
                    ys[i] = y[i] + dot * h;
                }
                orc_rhs(p, &c, N, ys, K + s * n);
            }
            for (int64_t i = 0; i < n; i++) {
                double dot = 0;
                for (int j = 0; j < 6; j++) dot += K[j * n + i] * DP_B[j];
                ynew[i] = y[i] + h * dot;
            }
            orc_rhs(p, &c, N, ynew, K + 6 * n);
            st->nfev += 6;
            /* error norm, rk.py:103-107, 146-147; common.py:63-65 */
            double ss = 0;
            for (int64_t i = 0; i < n; i++) {
                double dot = 0;
                for (int j = 0; j < 7; j++) dot += K[j * n + i] * DP_E[j];
                const double ay = fabs(y[i]), an = fabs(ynew[i]);
                const double scale = atol + (ay > an || ay != ay ? ay : an) * rtol; /* np.maximum propagates NaN */
                const double e = dot * h / scale;
                ss += e * e;
            }
            const double error_norm = sqrt(ss) / sqrt((double)n);
            if (error_norm < 1) {
                double factor = (error_norm == 0) ? DP_MAX_FACTOR : fmin(DP_MAX_FACTOR, DP_SAFETY * pow(error_norm, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                h_abs *= factor;
                accepted = 1;
            } else {
                const double f = DP_SAFETY * pow(error_norm, -0.2);
                h_abs *= (f > DP_MIN_FACTOR) ? f : DP_MIN_FACTOR; /* max(0.2, nan) -> 0.2 */
                rejected = 1;
                st->n_rejected++;
            }
        }
        if (status != 1) break;
        st->n_accepted++;
        memcpy(yold, y, sizeof(double) * n);
        memcpy(y, ynew, sizeof(double) * n);
        const double t_old = t;
        t = t_new;
        if (t - t1 >= 0) status = 0;                     /* base.py:203-204 */
        if (step_times && steps_out < max_steps_out) step_times[steps_out] = t;
        steps_out++;

        orc_dense dense = {p, &c, N, t_old, h, yold, K, scratch, NULL, NULL, 0};
        /* events, ivp.py:673-694 + find_active_events :131-156 (direction 0, non-terminal) */
        orc_events(p, &c, N, y, g_new);
        for (int e = 0; e < MARL_NEVENTS; e++) {
            const int up = (g[e] <= 0) && (g_new[e] >= 0), down = (g[e] >= 0) && (g_new[e] <= 0);
            if (up || down) {
                if (t_events && st->n_events[e] < max_events)
                    t_events[e * max_events + st->n_events[e]] = orc_brent(&dense, e, t_old, t);
                st->n_events[e]++;
            }
            g[e] = g_new[e];
        }
        /* t_eval, ivp.py:706-723: samples in (t_old, t] (t0 itself is emitted on the first step) */
        while (t_eval && eval_i < n_eval && t_eval[eval_i] <= t) {
            orc_dense_eval(&dense, t_eval[eval_i], y_eval + eval_i * n);
            eval_i++;
        }
        memcpy(K, K + 6 * n, sizeof(double) * n);        /* self.f = f_new */
    }
    st->status = status;
    st->t = t;
    st->h_next = h_abs;
    orc_events(p, &c, N, y, st->event_value);
    if (n_steps_out) *n_steps_out = steps_out;
    free(buf);
    return status;
}

/* Batched sweep = the same integrators over `batch` independent instances (params[b], y + b*5N).
 * Not in the reference (it runs one scenario per process); BASELINE configs 3-4. */
int marl_oracle_rk4_batch(const marl_params *p, int64_t batch, int64_t N, double *y, const double *dt, int64_t nsteps)
{
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) if (batch > 1)
#endif
    for (int64_t b = 0; b < batch; b++) {
        int r = marl_oracle_rk4(&p[b], N, y + b * NF * N, dt[b], nsteps);
        if (r) rc = r;
    }
    return rc;
}


/* ==========================================================================================
 * Implicit Radau IIA (order 5) exactly as scipy drives it for the reference - the reference's DEFAULT
 * solver (marlpde/parameters.py:213; call site marlpde/Evolve_scenario.py:104-109 with
 * jac_sparsity = the 27-diagonal pattern of marlpde/parameters.py:150-199).  Restated from
 * scipy/integrate/_ivp/radau.py (constants :11-44, solve_collocation_system :47-130, predict_factor
 * :133-173, Radau.__init__ :290-343, _step_impl :404-537, RadauDenseOutput :549-572) and
 * scipy/integrate/_ivp/common.py (num_jac :268-344, _sparse_num_jac :389-451, norm :63-65).
 *
 * Jacobian: finite differences over column groups.  `groups[5N]` is the column grouping scipy derives from
 * the sparsity pattern (scipy.optimize._numdiff.group_columns, a greedy colouring in a seeded random column
 * order); the caller may pass scipy's own array, or NULL for the structured
 * 15-colouring (3 cell residues x 5 fields).  (scipy's nfev does not count the finite-difference columns, so the
 * grouping shows in no statistic.)  The VALUE of every Jacobian entry is the same either way: an entry
 * (row, col) is (f(y + h e_group)[row] - f(y)[row]) / h[col], and no other column of a valid group touches `row`.
 * Structure kept: rows (f, i) x columns (f', i-1..i+1), minus the CA/CC rows x Phi columns the reference's
 * pattern zeroes (parameters.py:197).  (The pattern's wrap-around entries - offset +-1 diagonals crossing a field
 * boundary - receive exact zeros from the finite differences and are not stored here.)
 *
 * Linear algebra: scipy factorises MU/h I - J with SuperLU; here the same matrices are ordered cell-major
 * (5i + f: block tridiagonal, half bandwidth 9) and factorised by banded LU with partial pivoting (LAPACK
 * dgbtf2 / dgbtrs restated).  Solutions agree to rounding, not bit for bit.
 * ========================================================================================== */
#include <complex.h>

#define RD_KL 9
#define RD_KU 9
#define RD_KV (RD_KL + RD_KU)
#define RD_LDAB (2 * RD_KL + RD_KU + 1)

#define RD_DEFINE_BAND(T, SUF, ABS1)                                                                      \
    /* LAPACK xGBTF2: ab[(kv + i - j) + j*ldab] = A(i, j); returns 0 or 1 + index of a zero pivot */       \
    static int band_factor_##SUF(int64_t n, T *ab, int32_t *ipiv)                                         \
    {                                                                                                     \
        int info = 0;                                                                                     \
        for (int64_t j = RD_KU + 1; j < (RD_KV < n ? RD_KV : n); j++)                                     \
            for (int64_t i = RD_KV - j; i < RD_KL; i++) ab[i + j * RD_LDAB] = 0;                          \
        int64_t ju = 0;                                                                                   \
        for (int64_t j = 0; j < n; j++) {                                                                 \
            if (j + RD_KV < n)                                                                            \
                for (int64_t i = 0; i < RD_KL; i++) ab[i + (j + RD_KV) * RD_LDAB] = 0;                    \
            const int64_t km = (RD_KL < n - 1 - j) ? RD_KL : n - 1 - j;                                   \
            int64_t jp = 0;                                                                               \
            double best = -1;                                                                             \
            for (int64_t i = 0; i <= km; i++) {                                                           \
                const double a = ABS1(ab[RD_KV + i + j * RD_LDAB]);                                       \
                if (a > best) { best = a; jp = i; }                                                       \
            }                                                                                             \
            ipiv[j] = (int32_t)(jp + j);                                                                  \
            if (ab[RD_KV + jp + j * RD_LDAB] != 0) {                                                      \
                const int64_t cand = (j + RD_KU + jp < n - 1) ? j + RD_KU + jp : n - 1;                   \
                if (cand > ju) ju = cand;                                                                 \
                if (jp != 0)                                                                              \
                    for (int64_t c = j; c <= ju; c++) {                                                   \
                        T *a = &ab[RD_KV + jp + j - c + c * RD_LDAB], *b = &ab[RD_KV + j - c + c * RD_LDAB]; \
                        const T tmp = *a; *a = *b; *b = tmp;                                              \
                    }                                                                                     \
                if (km > 0) {                                                                             \
                    const T r = 1.0 / ab[RD_KV + j * RD_LDAB];                                            \
                    for (int64_t i = 1; i <= km; i++) ab[RD_KV + i + j * RD_LDAB] *= r;                   \
                    for (int64_t c = j + 1; c <= ju; c++) {                                               \
                        const T u = ab[RD_KV + j - c + c * RD_LDAB];                                      \
                        if (u != 0)                                                                       \
                            for (int64_t i = 1; i <= km; i++)                                             \
                                ab[RD_KV + j + i - c + c * RD_LDAB] -= ab[RD_KV + i + j * RD_LDAB] * u;   \
                    }                                                                                     \
                }                                                                                         \
            } else if (!info) info = (int)(j + 1);                                                        \
        }                                                                                                 \
        return info;                                                                                      \
    }                                                                                                     \
    /* LAPACK xGBTRS, no transpose, one right-hand side (in place) */                                      \
    static void band_solve_##SUF(int64_t n, const T *ab, const int32_t *ipiv, T *b)                       \
    {                                                                                                     \
        for (int64_t j = 0; j < n - 1; j++) {                                                             \
            const int64_t lm = (RD_KL < n - 1 - j) ? RD_KL : n - 1 - j;                                   \
            const int64_t l = ipiv[j];                                                                    \
            if (l != j) { const T tmp = b[l]; b[l] = b[j]; b[j] = tmp; }                                  \
            const T bj = b[j];                                                                            \
            for (int64_t i = 1; i <= lm; i++) b[j + i] -= bj * ab[RD_KV + i + j * RD_LDAB];               \
        }                                                                                                 \
        for (int64_t j = n - 1; j >= 0; j--) {                                                            \
            b[j] /= ab[RD_KV + j * RD_LDAB];                                                              \
            const T bj = b[j];                                                                            \
            const int64_t lo = (j - RD_KV > 0) ? j - RD_KV : 0;                                           \
            for (int64_t i = j - 1; i >= lo; i--) b[i] -= bj * ab[RD_KV + i - j + j * RD_LDAB];           \
        }                                                                                                 \
    }

#define RD_ABS_REAL(x) fabs(x)
#define RD_ABS_CPLX(x) (fabs(creal(x)) + fabs(cimag(x)))
RD_DEFINE_BAND(double, d, RD_ABS_REAL)
RD_DEFINE_BAND(double complex, z, RD_ABS_CPLX)

/* radau.py:11-44 */
#define RD_S6 2.449489742783178 /* 6 ** 0.5 */
static const double RD_T[3][3] = {{0.09443876248897524, -0.14125529502095421, 0.03002919410514742},
                                  {0.25021312296533332, 0.20412935229379994, -0.38294211275726192},
                                  {1, 1, 0}};
static const double RD_TI[3][3] = {{4.17871859155190428, 0.32768282076106237, 0.52337644549944951},
                                   {-4.17871859155190428, -0.32768282076106237, 0.47662355450055044},
                                   {0.50287263494578682, -2.57192694985560522, 0.59603920482822492}};
#define RD_NEWTON_MAXITER 6
#define RD_MIN_FACTOR 0.2
#define RD_MAX_FACTOR 10.0
#define RD_EPS 2.220446049250313e-16

typedef struct {
    const marl_params *p;
    const orc_consts *c;
    int64_t N, n;
    double *J;      /* [N][3][5][5]: J[((i*3 + d)*5 + f)*5 + fp] = d rate(f, i) / d y(fp, i + d - 1) */
    double *factor; /* num_jac's per-column step factors (carried from call to call), or NULL before the first call */
    const int32_t *groups;
    int32_t n_groups;
    int32_t *own_groups;
    marl_stats *st;
} rd_ctx;

static void rd_fun(const rd_ctx *R, const double *y, double *out)
{
    orc_rhs(R->p, R->c, R->N, y, out);
    R->st->nfev++;
}

/* The finite-difference columns go through OdeSolver.fun_vectorized, which calls the user function WITHOUT counting
 * (scipy/integrate/_ivp/base.py:150-156): scipy's nfev excludes them. */
static void rd_fun_uncounted(const rd_ctx *R, const double *y, double *out) { orc_rhs(R->p, R->c, R->N, y, out); }

/* is (row field f) x (column field fp) in the reference's pattern?  (parameters.py:197) */
static inline int rd_in_pattern(int f, int fp) { return !(f < 2 && fp == 4); }

/* the |diff| column of one perturbed column j = (fp, ip): rows sorted as scipy's csc stores them (row index f*N + i
 * ascending).  Returns max |diff|, the row index of its FIRST occurrence, and the diffs in dcol[f][di], di = i - ip + 1. */
static double rd_column_diff(const rd_ctx *R, const double *f0, const double *fnew, int fp, int64_t ip, double dcol[NF][3],
                             int64_t *max_row)
{
    const int64_t N = R->N;
    double best = 0;
    int64_t arg = -1;
    for (int f = 0; f < NF; f++)
        for (int di = 0; di < 3; di++) {
            const int64_t i = ip + di - 1;
            dcol[f][di] = 0;
            if (i < 0 || i >= N || !rd_in_pattern(f, fp)) continue;
            const int64_t r = f * N + i;
            const double d = fnew[r] - f0[r];
            dcol[f][di] = d;
            if (arg < 0 || fabs(d) > best) { best = fabs(d); arg = r; } /* np.argmax: first occurrence */
        }
    /* scipy's sparse argmax: an all-zero column reports row 0 (scipy/sparse/_data.py:265-272: min(position 0, first
     * implicit zero)) */
    if (best == 0) arg = 0;
    *max_row = arg;
    return best;
}

/* num_jac + _sparse_num_jac, common.py:268-451.  f0 = fun(t, y). */
static int rd_num_jac(rd_ctx *R, const double *y, const double *f0, double threshold)
{
    const int64_t n = R->n, N = R->N;
    const double REJECT = pow(RD_EPS, 0.875), SMALL = pow(RD_EPS, 0.75), BIG = pow(RD_EPS, 0.25), MINF = 1e3 * RD_EPS;
    const int ng = R->n_groups;
    double *work = (double *)malloc(sizeof(double) * (size_t)(n * (6 + 2 * (size_t)ng)));
    unsigned char *small = (unsigned char *)calloc((size_t)n + (size_t)ng, 1);
    if (!work || !small) { free(work); free(small); return -2; }
    double *h = work, *y_scale = work + n, *max_diff = work + 2 * n, *scale = work + 3 * n, *ys = work + 4 * n, *h_new = work + 5 * n;
    double *fnew = work + 6 * n, *fnew2 = fnew + (size_t)ng * n;
    unsigned char *group_hit = small + n;
    if (!R->factor) {
        R->factor = (double *)malloc(sizeof(double) * n);
        if (!R->factor) { free(work); free(small); return -2; }
        for (int64_t j = 0; j < n; j++) R->factor[j] = sqrt(RD_EPS); /* EPS ** 0.5 */
    }
    double *factor = R->factor;
    R->st->njev++;
    for (int64_t j = 0; j < n; j++) {
        const double f_sign = (f0[j] >= 0) ? 1.0 : -1.0;
        const double ay = fabs(y[j]);
        y_scale[j] = f_sign * (threshold > ay ? threshold : ay); /* np.maximum(threshold, |y|) */
        h[j] = (y[j] + factor[j] * y_scale[j]) - y[j];
        while (h[j] == 0) { factor[j] *= 10; h[j] = (y[j] + factor[j] * y_scale[j]) - y[j]; }
    }
    for (int g = 0; g < ng; g++) {
        for (int64_t j = 0; j < n; j++) ys[j] = y[j] + (R->groups[j] == g ? h[j] : 0.0);
        rd_fun_uncounted(R, ys, fnew + (size_t)g * n);
    }
    int any_small = 0;
    double *J = R->J;
    for (int fp = 0; fp < NF; fp++)
        for (int64_t ip = 0; ip < N; ip++) {
            const int64_t j = fp * N + ip;
            const double *fg = fnew + (size_t)R->groups[j] * n;
            double dcol[NF][3];
            int64_t mr;
            max_diff[j] = rd_column_diff(R, f0, fg, fp, ip, dcol, &mr);
            const double a = fabs(f0[mr]), b = fabs(fg[mr]);
            scale[j] = a > b ? a : b;
            for (int f = 0; f < NF; f++)
                for (int di = 0; di < 3; di++) {
                    const int64_t i = ip + di - 1;
                    if (i >= 0 && i < N) J[((i * 3 + (2 - di)) * NF + f) * NF + fp] = dcol[f][di];
                }
            if (max_diff[j] < REJECT * scale[j]) { small[j] = 1; group_hit[R->groups[j]] = 1; any_small = 1; }
        }
    if (any_small) {
        for (int64_t j = 0; j < n; j++) h_new[j] = small[j] ? (y[j] + 10 * factor[j] * y_scale[j]) - y[j] : 0.0;
        for (int g = 0; g < ng; g++) {
            if (!group_hit[g]) continue;
            for (int64_t j = 0; j < n; j++) ys[j] = y[j] + (R->groups[j] == g ? h_new[j] : 0.0);
            rd_fun_uncounted(R, ys, fnew2 + (size_t)g * n);
        }
        for (int fp = 0; fp < NF; fp++)
            for (int64_t ip = 0; ip < N; ip++) {
                const int64_t j = fp * N + ip;
                if (!small[j]) continue;
                const double *fg = fnew2 + (size_t)R->groups[j] * n;
                double dcol[NF][3];
                int64_t mr;
                const double md_new = rd_column_diff(R, f0, fg, fp, ip, dcol, &mr);
                const double a = fabs(f0[mr]), b = fabs(fg[mr]);
                const double scale_new = a > b ? a : b;
                if (max_diff[j] * scale_new < md_new * scale[j]) {
                    factor[j] = 10 * factor[j];
                    h[j] = h_new[j];
                    scale[j] = scale_new;
                    max_diff[j] = md_new;
                    for (int f = 0; f < NF; f++)
                        for (int di = 0; di < 3; di++) {
                            const int64_t i = ip + di - 1;
                            if (i >= 0 && i < N) J[((i * 3 + (2 - di)) * NF + f) * NF + fp] = dcol[f][di];
                        }
                }
            }
    }
    for (int fp = 0; fp < NF; fp++)
        for (int64_t ip = 0; ip < N; ip++) {
            const int64_t j = fp * N + ip;
            for (int f = 0; f < NF; f++)
                for (int di = 0; di < 3; di++) {
                    const int64_t i = ip + di - 1;
                    if (i >= 0 && i < N) J[((i * 3 + (2 - di)) * NF + f) * NF + fp] /= h[j];
                }
            if (max_diff[j] < SMALL * scale[j]) factor[j] *= 10;
            if (max_diff[j] > BIG * scale[j]) factor[j] *= 0.1;
            if (factor[j] < MINF) factor[j] = MINF;
        }
    free(work);
    free(small);
    return 0;
}

/* band storage of  mu I - J  in the cell-major ordering 5 i + f */
static void rd_assemble_real(const rd_ctx *R, double mu, double *ab)
{
    const int64_t n = R->n, N = R->N;
    memset(ab, 0, sizeof(double) * (size_t)(RD_LDAB * n));
    for (int64_t i = 0; i < N; i++)
        for (int d = 0; d < 3; d++) {
            const int64_t ic = i + d - 1;
            if (ic < 0 || ic >= N) continue;
            for (int f = 0; f < NF; f++)
                for (int fp = 0; fp < NF; fp++) {
                    const int64_t r = NF * i + f, c = NF * ic + fp;
                    ab[RD_KV + r - c + c * RD_LDAB] = (r == c ? mu : 0.0) - R->J[((i * 3 + d) * NF + f) * NF + fp];
                }
        }
}

static void rd_assemble_cplx(const rd_ctx *R, double complex mu, double complex *ab)
{
    const int64_t n = R->n, N = R->N;
    memset(ab, 0, sizeof(double complex) * (size_t)(RD_LDAB * n));
    for (int64_t i = 0; i < N; i++)
        for (int d = 0; d < 3; d++) {
            const int64_t ic = i + d - 1;
            if (ic < 0 || ic >= N) continue;
            for (int f = 0; f < NF; f++)
                for (int fp = 0; fp < NF; fp++) {
                    const int64_t r = NF * i + f, c = NF * ic + fp;
                    ab[RD_KV + r - c + c * RD_LDAB] = (r == c ? mu : 0.0) - R->J[((i * 3 + d) * NF + f) * NF + fp];
                }
        }
}

/* field-major vector <-> cell-major vector */
static inline int64_t rd_perm(int64_t N, int64_t k) { return (k % NF) * N + k / NF; } /* cell-major index k -> field-major index */

/* norm(x) = ||x||_2 / sqrt(size), common.py:63-65 */
static double rd_rms_scaled(const double *x, const double *scale, int64_t n, int rows)
{
    double ss = 0;
    for (int r = 0; r < rows; r++)
        for (int64_t i = 0; i < n; i++) { const double v = x[r * n + i] / scale[i]; ss += v * v; }
    return sqrt(ss) / sqrt((double)(rows * n));
}

/* predict_factor, radau.py:133-173.  *_old < 0 encodes None. */
static double rd_predict_factor(double h_abs, double h_abs_old, double error_norm, double error_norm_old)
{
    double multiplier;
    if (error_norm_old < 0 || h_abs_old < 0 || error_norm == 0) multiplier = 1;
    else multiplier = h_abs / h_abs_old * pow(error_norm_old / error_norm, 0.25);
    return (multiplier < 1 ? multiplier : 1) * pow(error_norm, -0.25); /* error_norm = 0 -> inf (errstate ignore) */
}

int marl_oracle_radau(const marl_params *p, int64_t N, double *y, double t0, double t1, double first_step,
                      double rtol, double atol, const int32_t *groups,
                      const double *t_eval, int64_t n_eval, double *y_eval,
                      double *step_times, int64_t max_steps_out, int64_t *n_steps_out,
                      double *t_events, int64_t max_events, int64_t max_attempts, marl_stats *st)
{
    orc_consts c;
    orc_derive(p, N, &c);
    const int64_t n = NF * N;
    const double S6 = sqrt(6.0);
    const double C3[3] = {(4 - S6) / 10, (4 + S6) / 10, 1};
    const double E3[3] = {(-13 - 7 * S6) / 3, (-13 + 7 * S6) / 3, -1.0 / 3};
    const double MU_REAL = 3 + pow(3, 2.0 / 3) - pow(3, 1.0 / 3);
    const double complex MU_COMPLEX = (3 + 0.5 * (pow(3, 1.0 / 3) - pow(3, 2.0 / 3))) - 0.5 * I * (pow(3, 5.0 / 6) + pow(3, 7.0 / 6));
    const double P3[3][3] = {{13.0 / 3 + 7 * S6 / 3, -23.0 / 3 - 22 * S6 / 3, 10.0 / 3 + 5 * S6},
                             {13.0 / 3 - 7 * S6 / 3, -23.0 / 3 + 22 * S6 / 3, 10.0 / 3 - 5 * S6},
                             {1.0 / 3, -8.0 / 3, 10.0 / 3}};
    memset(st, 0, sizeof *st);
    if (rtol < 100 * RD_EPS) rtol = 100 * RD_EPS; /* validate_tol, common.py:44-51 */

    rd_ctx R = {p, &c, N, n, NULL, NULL, groups, 0, NULL, st};
    if (!groups) { /* structured colouring: columns (f, i) and (f', i') never share a row when i = i' mod 3 and f = f' */
        R.own_groups = (int32_t *)malloc(sizeof(int32_t) * n);
        for (int64_t j = 0; j < n; j++) R.own_groups[j] = (int32_t)(3 * (j / N) + (j % N) % 3);
        R.groups = R.own_groups;
    }
    for (int64_t j = 0; j < n; j++)
        if (R.groups[j] + 1 > R.n_groups) R.n_groups = R.groups[j] + 1;

    double *buf = (double *)malloc(sizeof(double) * (size_t)n * 24);
    R.J = (double *)calloc((size_t)N * 75, sizeof(double));
    double *ab_r = (double *)malloc(sizeof(double) * (size_t)(RD_LDAB * n));
    double complex *ab_c = (double complex *)malloc(sizeof(double complex) * (size_t)(RD_LDAB * n));
    double complex *rhs_c = (double complex *)malloc(sizeof(double complex) * (size_t)n);
    int32_t *piv_r = (int32_t *)malloc(sizeof(int32_t) * n), *piv_c = (int32_t *)malloc(sizeof(int32_t) * n);
    if (!buf || !R.J || !ab_r || !ab_c || !rhs_c || !piv_r || !piv_c) return -2;
    double *f = buf, *fnew = buf + n, *Z = buf + 2 * n /* 3n */, *W = buf + 5 * n /* 3n */, *F = buf + 8 * n /* 3n */;
    double *dW = buf + 11 * n /* 3n */, *Z0 = buf + 14 * n /* 3n */, *scale = buf + 17 * n, *ynew = buf + 18 * n, *err = buf + 19 * n;
    double *yold = buf + 20 * n, *Q = buf + 21 * n /* 3n: [i][3] */;
    double *rhs_r = (double *)malloc(sizeof(double) * (size_t)n * 3);
    double *ys = rhs_r + n, *scratch = rhs_r + 2 * n;
    double g[MARL_NEVENTS], g_new[MARL_NEVENTS];
    int64_t steps_out = 0, eval_i = 0, attempts = 0;

    /* Radau.__init__, radau.py:290-343 */
    double t = t0;
    rd_fun(&R, y, f);
    double S_h_abs = first_step, S_h_abs_old = -1, S_err_old = -1; /* self.h_abs, self.h_abs_old, self.error_norm_old; < 0: None */
    const double newton_tol = fmax(10 * RD_EPS / rtol, fmin(0.03, sqrt(rtol)));
    int have_sol = 0;
    double sol_t_old = t0, sol_h = 0;
    if (rd_num_jac(&R, y, f, atol)) return -2;
    int current_jac = 1, have_lu = 0;
    orc_events(p, &c, N, y, g); /* ivp.py:645 */
    int status = 1;

    while (status == 1) {
        if (t == t1) { status = 0; break; } /* base.py:189-194 */
        /* ---- _step_impl, radau.py:404-537 ---- */
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        double h_abs = S_h_abs, h_abs_o = S_h_abs_old, err_o = S_err_old;
        if (S_h_abs < min_step) { h_abs = min_step; h_abs_o = -1; err_o = -1; } /* max_step = inf */
        int rejected = 0, accepted = 0, n_iter = 0;
        double rate = -1, h = 0, t_new = t, error_norm = 0, safety = 0;
        while (!accepted) {
            if (h_abs < min_step) { status = -1; break; }
            if (max_attempts > 0 && attempts >= max_attempts) { status = 2; break; }
            attempts++;
            h = h_abs;
            t_new = t + h;
            if (t_new - t1 > 0) t_new = t1;
            h = t_new - t;
            h_abs = fabs(h);
            if (!have_sol) {
                memset(Z0, 0, sizeof(double) * 3 * n);
            } else { /* Z0 = self.sol(t + h * C).T - y */
                for (int s = 0; s < 3; s++) {
                    const double x = ((t + h * C3[s]) - sol_t_old) / sol_h;
                    const double p1 = x, p2 = p1 * x, p3 = p2 * x;
                    for (int64_t i = 0; i < n; i++)
                        Z0[s * n + i] = (((Q[3 * i] * p1 + Q[3 * i + 1] * p2) + Q[3 * i + 2] * p3) + yold[i]) - y[i];
                }
            }
            for (int64_t i = 0; i < n; i++) scale[i] = atol + fabs(y[i]) * rtol;

            int converged = 0;
            while (!converged) {
                if (!have_lu) {
                    rd_assemble_real(&R, MU_REAL / h, ab_r);
                    band_factor_d(n, ab_r, piv_r);
                    st->nlu++;
                    rd_assemble_cplx(&R, MU_COMPLEX / h, ab_c);
                    band_factor_z(n, ab_c, piv_c);
                    st->nlu++;
                    have_lu = 1;
                }
                /* ---- solve_collocation_system, radau.py:47-130 ---- */
                const double M_real = MU_REAL / h;
                const double complex M_complex = MU_COMPLEX / h;
                for (int r = 0; r < 3; r++)
                    for (int64_t i = 0; i < n; i++)
                        W[r * n + i] = (RD_TI[r][0] * Z0[i] + RD_TI[r][1] * Z0[n + i]) + RD_TI[r][2] * Z0[2 * n + i];
                memcpy(Z, Z0, sizeof(double) * 3 * n);
                double dW_norm_old = -1;
                rate = -1;
                int k;
                for (k = 0; k < RD_NEWTON_MAXITER; k++) {
                    int finite = 1;
                    for (int s = 0; s < 3; s++) {
                        for (int64_t i = 0; i < n; i++) ys[i] = y[i] + Z[s * n + i];
                        rd_fun(&R, ys, F + s * n);
                    }
                    for (int64_t i = 0; i < 3 * n && finite; i++)
                        if (!isfinite(F[i])) finite = 0;
                    if (!finite) break;
                    for (int64_t kk = 0; kk < n; kk++) { /* right-hand sides in the cell-major ordering of the band matrices */
                        const int64_t i = rd_perm(N, kk);
                        const double fr = (F[i] * RD_TI[0][0] + F[n + i] * RD_TI[0][1]) + F[2 * n + i] * RD_TI[0][2];
                        rhs_r[kk] = fr - M_real * W[i];
                        const double complex fc = (F[i] * (RD_TI[1][0] + I * RD_TI[2][0]) + F[n + i] * (RD_TI[1][1] + I * RD_TI[2][1]))
                                                  + F[2 * n + i] * (RD_TI[1][2] + I * RD_TI[2][2]);
                        rhs_c[kk] = fc - M_complex * (W[n + i] + I * W[2 * n + i]);
                    }
                    band_solve_d(n, ab_r, piv_r, rhs_r);
                    band_solve_z(n, ab_c, piv_c, rhs_c);
                    for (int64_t kk = 0; kk < n; kk++) {
                        const int64_t i = rd_perm(N, kk);
                        dW[i] = rhs_r[kk];
                        dW[n + i] = creal(rhs_c[kk]);
                        dW[2 * n + i] = cimag(rhs_c[kk]);
                    }
                    const double dW_norm = rd_rms_scaled(dW, scale, n, 3);
                    if (dW_norm_old >= 0) rate = dW_norm / dW_norm_old;
                    if (rate >= 0 && (rate >= 1 || pow(rate, RD_NEWTON_MAXITER - k) / (1 - rate) * dW_norm > newton_tol)) break;
                    for (int64_t i = 0; i < 3 * n; i++) W[i] += dW[i];
                    for (int r = 0; r < 3; r++)
                        for (int64_t i = 0; i < n; i++)
                            Z[r * n + i] = (RD_T[r][0] * W[i] + RD_T[r][1] * W[n + i]) + RD_T[r][2] * W[2 * n + i];
                    if (dW_norm == 0 || (rate >= 0 && rate / (1 - rate) * dW_norm < newton_tol)) { converged = 1; break; }
                    dW_norm_old = dW_norm;
                }
                n_iter = (k < RD_NEWTON_MAXITER ? k : RD_NEWTON_MAXITER - 1) + 1; /* python: k + 1 with k the last loop value */
                if (!converged) {
                    if (current_jac) break;
                    if (rd_num_jac(&R, y, f, atol)) return -2;
                    current_jac = 1;
                    have_lu = 0;
                }
            }
            if (!converged) {
                h_abs *= 0.5;
                have_lu = 0;
                st->n_rejected++;
                continue;
            }
            for (int64_t i = 0; i < n; i++) ynew[i] = y[i] + Z[2 * n + i];
            /* error = solve_lu(LU_real, f + Z.T.dot(E) / h) */
            for (int64_t kk = 0; kk < n; kk++) {
                const int64_t i = rd_perm(N, kk);
                const double ZE = ((Z[i] * E3[0] + Z[n + i] * E3[1]) + Z[2 * n + i] * E3[2]) / h;
                rhs_r[kk] = f[i] + ZE;
            }
            band_solve_d(n, ab_r, piv_r, rhs_r);
            for (int64_t kk = 0; kk < n; kk++) err[rd_perm(N, kk)] = rhs_r[kk];
            for (int64_t i = 0; i < n; i++) {
                const double a = fabs(y[i]), b = fabs(ynew[i]);
                scale[i] = atol + ((a > b || a != a) ? a : b) * rtol;
            }
            error_norm = rd_rms_scaled(err, scale, n, 1);
            safety = 0.9 * (2 * RD_NEWTON_MAXITER + 1) / (2 * RD_NEWTON_MAXITER + n_iter);
            if (rejected && error_norm > 1) {
                for (int64_t i = 0; i < n; i++) ys[i] = y[i] + err[i];
                rd_fun(&R, ys, scratch);
                for (int64_t kk = 0; kk < n; kk++) {
                    const int64_t i = rd_perm(N, kk);
                    const double ZE = ((Z[i] * E3[0] + Z[n + i] * E3[1]) + Z[2 * n + i] * E3[2]) / h;
                    rhs_r[kk] = scratch[i] + ZE;
                }
                band_solve_d(n, ab_r, piv_r, rhs_r);
                for (int64_t kk = 0; kk < n; kk++) err[rd_perm(N, kk)] = rhs_r[kk];
                error_norm = rd_rms_scaled(err, scale, n, 1);
            }
            if (error_norm > 1) {
                const double factor = rd_predict_factor(h_abs, h_abs_o, error_norm, err_o);
                const double sf = safety * factor;
                h_abs *= (sf > RD_MIN_FACTOR) ? sf : RD_MIN_FACTOR; /* max(MIN_FACTOR, x); NaN -> MIN_FACTOR */
                have_lu = 0;
                rejected = 1;
                st->n_rejected++;
            } else {
                accepted = 1; /* (a NaN error norm is "not > 1": scipy accepts it too) */
            }
        }
        if (status != 1) break;
        const int recompute_jac = n_iter > 2 && rate > 1e-3;
        double factor = rd_predict_factor(h_abs, h_abs_o, error_norm, err_o);
        { const double sf = safety * factor; factor = (sf < RD_MAX_FACTOR) ? sf : RD_MAX_FACTOR; } /* min(MAX_FACTOR, x) */
        if (!recompute_jac && factor < 1.2) factor = 1;
        else have_lu = 0;
        rd_fun(&R, ynew, fnew);
        if (recompute_jac) {
            if (rd_num_jac(&R, ynew, fnew, atol)) return -2;
            current_jac = 1;
        } else {
            current_jac = 0;
        }
        /* radau.py:512-515: self.h_abs_old receives the value self.h_abs had when this step STARTED (the size proposed by
         * the previous step), not the possibly reduced / clipped h_abs just used */
        S_h_abs_old = S_h_abs;
        S_err_old = error_norm;
        S_h_abs = h_abs * factor;
        st->n_accepted++;
        memcpy(yold, y, sizeof(double) * n);
        memcpy(y, ynew, sizeof(double) * n);
        memcpy(f, fnew, sizeof(double) * n);
        const double t_old = t;
        t = t_new;
        /* _compute_dense_output: Q = Z.T . P */
        for (int64_t i = 0; i < n; i++)
            for (int m = 0; m < 3; m++) Q[3 * i + m] = (Z[i] * P3[0][m] + Z[n + i] * P3[1][m]) + Z[2 * n + i] * P3[2][m];
        have_sol = 1;
        sol_t_old = t_old;
        sol_h = t - t_old;
        if (t - t1 >= 0) status = 0;
        if (step_times && steps_out < max_steps_out) step_times[steps_out] = t;
        steps_out++;

        orc_dense dense = {p, &c, N, t_old, sol_h, yold, NULL, scratch, Q, NULL, 0};
        orc_events(p, &c, N, y, g_new);
        for (int e = 0; e < MARL_NEVENTS; e++) {
            const int up = (g[e] <= 0) && (g_new[e] >= 0), down = (g[e] >= 0) && (g_new[e] <= 0);
            if (up || down) {
                if (t_events && st->n_events[e] < max_events)
                    t_events[e * max_events + st->n_events[e]] = orc_brent(&dense, e, t_old, t);
                st->n_events[e]++;
            }
            g[e] = g_new[e];
        }
        while (t_eval && eval_i < n_eval && t_eval[eval_i] <= t) {
            orc_dense_eval(&dense, t_eval[eval_i], y_eval + eval_i * n);
            eval_i++;
        }
    }
    st->status = status;
    st->t = t;
    st->h_next = S_h_abs;
    orc_events(p, &c, N, y, st->event_value);
    if (n_steps_out) *n_steps_out = steps_out;
    free(buf); free(R.J); free(R.factor); free(R.own_groups); free(ab_r); free(ab_c); free(rhs_c); free(piv_r); free(piv_c); free(rhs_r);
    return status;
}


/* =============================================================================================================================
 * scipy.integrate.solve_ivp(method="BDF", jac_sparsity=...) restated (scipy/integrate/_ivp/bdf.py; driver ivp.py:654-723): the
 * variable-order (1..5) quasi-constant-step NDF method the reference's Solver offers next to its default Radau
 * (marlpde/parameters.py:205-219: "Radau" and "BDF" take the jac_sparsity).  TEST INFRASTRUCTURE like the rest of this file.
 * Shares the finite-difference Jacobian (rd_num_jac) and the banded LU with the Radau restatement above; the matrix is I - c J.
 * ============================================================================================================================= */
#define BDF_MAX_ORDER 5
#define BDF_NEWTON_MAXITER 4

/* compute_R (bdf.py:18-25): M[0][:] = 1, M[i][j] = (i - 1 - factor j) / i for i, j >= 1; R = cumprod(M, axis = 0) */
static void bdf_compute_R(int order, double factor, double R[BDF_MAX_ORDER + 1][BDF_MAX_ORDER + 1])
{
    for (int j = 0; j <= order; j++) R[0][j] = 1;
    for (int i = 1; i <= order; i++) {
        R[i][0] = 0; /* M[i][0] = 0: the cumulative product of column 0 is 1, 0, 0, ... */
        for (int j = 1; j <= order; j++) R[i][j] = R[i - 1][j] * ((i - 1 - factor * j) / i);
    }
}

/* change_D (bdf.py:28-33): D[:order + 1] = (R U)^T D[:order + 1] */
static void bdf_change_D(double *D, int64_t n, int order, double factor)
{
    double R[BDF_MAX_ORDER + 1][BDF_MAX_ORDER + 1], U[BDF_MAX_ORDER + 1][BDF_MAX_ORDER + 1], RU[BDF_MAX_ORDER + 1][BDF_MAX_ORDER + 1];
    bdf_compute_R(order, factor, R);
    bdf_compute_R(order, 1.0, U);
    for (int i = 0; i <= order; i++)
        for (int j = 0; j <= order; j++) {
            double a = 0;
            for (int k = 0; k <= order; k++) a += R[i][k] * U[k][j];
            RU[i][j] = a;
        }
    for (int64_t e = 0; e < n; e++) {
        double d[BDF_MAX_ORDER + 1], o[BDF_MAX_ORDER + 1];
        for (int j = 0; j <= order; j++) d[j] = D[(int64_t)j * n + e];
        for (int k = 0; k <= order; k++) {
            double a = 0;
            for (int j = 0; j <= order; j++) a += RU[j][k] * d[j];
            o[k] = a;
        }
        for (int k = 0; k <= order; k++) D[(int64_t)k * n + e] = o[k];
    }
}

/* band storage of  I - c J  in the cell-major ordering 5 i + f */
static void bdf_assemble(const rd_ctx *R, double cc, double *ab)
{
    const int64_t n = R->n, N = R->N;
    memset(ab, 0, sizeof(double) * (size_t)(RD_LDAB * n));
    for (int64_t i = 0; i < N; i++)
        for (int d = 0; d < 3; d++) {
            const int64_t ic = i + d - 1;
            if (ic < 0 || ic >= N) continue;
            for (int f = 0; f < NF; f++)
                for (int fp = 0; fp < NF; fp++) {
                    const int64_t r = NF * i + f, c = NF * ic + fp;
                    ab[RD_KV + r - c + c * RD_LDAB] = (r == c ? 1.0 : 0.0) - cc * R->J[((i * 3 + d) * NF + f) * NF + fp];
                }
        }
}

/* norm(coef * v / scale), common.py:63-65 */
static double bdf_norm(const double *v, double coef, const double *scale, int64_t n)
{
    double s = 0;
    for (int64_t i = 0; i < n; i++) {
        const double e = coef * v[i] / scale[i];
        s += e * e;
    }
    return sqrt(s) / sqrt((double)n);
}

int marl_oracle_bdf(const marl_params *p, int64_t N, double *y, double t0, double t1, double first_step,
                    double rtol, double atol, const int32_t *groups,
                    const double *t_eval, int64_t n_eval, double *y_eval,
                    double *step_times, int64_t max_steps_out, int64_t *n_steps_out,
                    double *t_events, int64_t max_events, int64_t max_attempts, marl_stats *st)
{
    orc_consts c;
    orc_derive(p, N, &c);
    const int64_t n = NF * N;
    memset(st, 0, sizeof *st);
    if (rtol < 100 * RD_EPS) rtol = 100 * RD_EPS;
    rd_ctx R = {p, &c, N, n, NULL, NULL, groups, 0, NULL, st};
    if (!groups) {
        R.own_groups = (int32_t *)malloc(sizeof(int32_t) * n);
        for (int64_t j = 0; j < n; j++) R.own_groups[j] = (int32_t)(3 * (j / N) + (j % N) % 3);
        R.groups = R.own_groups;
    }
    for (int64_t j = 0; j < n; j++)
        if (R.groups[j] + 1 > R.n_groups) R.n_groups = R.groups[j] + 1;
    /* bdf.py:246-249 */
    const double kappa[6] = {0, -0.1850, -1.0 / 9, -0.0823, -0.0415, 0};
    double gamma_[6], alpha[6], error_const[6];
    gamma_[0] = 0;
    for (int k = 1; k <= BDF_MAX_ORDER; k++) gamma_[k] = gamma_[k - 1] + 1.0 / k; /* np.cumsum */
    for (int k = 0; k <= BDF_MAX_ORDER; k++) {
        alpha[k] = (1 - kappa[k]) * gamma_[k];
        error_const[k] = kappa[k] * gamma_[k] + 1.0 / (k + 1);
    }

    double *D = (double *)calloc((size_t)(BDF_MAX_ORDER + 3) * n, sizeof(double));
    double *buf = (double *)malloc(sizeof(double) * (size_t)n * 10);
    R.J = (double *)calloc((size_t)N * 75, sizeof(double));
    double *ab = (double *)malloc(sizeof(double) * (size_t)(RD_LDAB * n));
    int32_t *piv = (int32_t *)malloc(sizeof(int32_t) * n);
    if (!D || !buf || !R.J || !ab || !piv) return -2;
    double *f = buf, *ypred = buf + n, *scale = buf + 2 * n, *psi = buf + 3 * n, *ynew = buf + 4 * n, *d = buf + 5 * n, *rhs = buf + 6 * n,
           *dy = buf + 7 * n, *scratch = buf + 8 * n, *f0 = buf + 9 * n;
    double g[MARL_NEVENTS], g_new[MARL_NEVENTS];
    int64_t steps_out = 0, eval_i = 0, attempts = 0;

    /* BDF.__init__, bdf.py:197-258 */
    double t = t0;
    rd_fun(&R, y, f);
    double S_h_abs = first_step;
    const double newton_tol = fmax(10 * RD_EPS / rtol, fmin(0.03, sqrt(rtol)));
    rd_fun_uncounted(&R, y, f0); /* jac_wrapped: f = self.fun_single(t, y) - not counted */
    if (rd_num_jac(&R, y, f0, atol)) return -2;
    memcpy(D, y, sizeof(double) * n);
    for (int64_t i = 0; i < n; i++) D[n + i] = f[i] * S_h_abs;
    int order = 1, n_equal_steps = 0, have_lu = 0;
    orc_events(p, &c, N, y, g);
    int status = 1;

    while (status == 1) {
        if (t == t1) { status = 0; break; }
        /* ---- _step_impl, bdf.py:310-450 ---- */
        const double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        double h_abs;
        if (S_h_abs < min_step) {
            h_abs = min_step;
            bdf_change_D(D, n, order, min_step / S_h_abs);
            n_equal_steps = 0;
        } else {
            h_abs = S_h_abs;
        }
        int current_jac = 0; /* self.jac is a callable (the finite-difference wrapper) */
        int accepted = 0, n_iter = 0;
        double h = 0, t_new = t, error_norm = 0, safety = 0;
        while (!accepted) {
            if (h_abs < min_step) { status = -1; break; }
            if (max_attempts > 0 && attempts >= max_attempts) { status = 2; break; }
            attempts++;
            h = h_abs;
            t_new = t + h;
            if (t_new - t1 > 0) {
                t_new = t1;
                bdf_change_D(D, n, order, fabs(t_new - t) / h_abs);
                n_equal_steps = 0;
                have_lu = 0;
            }
            h = t_new - t;
            h_abs = fabs(h);
            for (int64_t i = 0; i < n; i++) { /* np.sum(D[:order + 1], axis=0): rows added in order */
                double a = D[i];
                for (int j = 1; j <= order; j++) a += D[(int64_t)j * n + i];
                ypred[i] = a;
                scale[i] = atol + rtol * fabs(a);
                double q = 0; /* np.dot(D[1:order + 1].T, gamma[1:order + 1]) / alpha[order] */
                for (int j = 1; j <= order; j++) q += D[(int64_t)j * n + i] * gamma_[j];
                psi[i] = q / alpha[order];
            }
            int converged = 0;
            const double cc = h / alpha[order];
            while (!converged) {
                if (!have_lu) {
                    bdf_assemble(&R, cc, ab);
                    band_factor_d(n, ab, piv);
                    st->nlu++;
                    have_lu = 1;
                }
                /* ---- solve_bdf_system, bdf.py:36-68 ---- */
                memset(d, 0, sizeof(double) * n);
                memcpy(ynew, ypred, sizeof(double) * n);
                double dy_norm_old = -1;
                int k;
                for (k = 0; k < BDF_NEWTON_MAXITER; k++) {
                    rd_fun(&R, ynew, scratch);
                    int finite = 1;
                    for (int64_t i = 0; i < n && finite; i++)
                        if (!isfinite(scratch[i])) finite = 0;
                    if (!finite) break;
                    for (int64_t kk = 0; kk < n; kk++) {
                        const int64_t i = rd_perm(N, kk);
                        rhs[kk] = (cc * scratch[i] - psi[i]) - d[i];
                    }
                    band_solve_d(n, ab, piv, rhs);
                    for (int64_t kk = 0; kk < n; kk++) dy[rd_perm(N, kk)] = rhs[kk];
                    const double dy_norm = bdf_norm(dy, 1.0, scale, n);
                    double rate = -1;
                    if (dy_norm_old >= 0) rate = dy_norm / dy_norm_old;
                    if (rate >= 0 && (rate >= 1 || pow(rate, BDF_NEWTON_MAXITER - k) / (1 - rate) * dy_norm > newton_tol)) break;
                    for (int64_t i = 0; i < n; i++) { ynew[i] += dy[i]; d[i] += dy[i]; }
                    if (dy_norm == 0 || (rate >= 0 && rate / (1 - rate) * dy_norm < newton_tol)) { converged = 1; break; }
                    dy_norm_old = dy_norm;
                }
                n_iter = (k < BDF_NEWTON_MAXITER ? k : BDF_NEWTON_MAXITER - 1) + 1;
                if (!converged) {
                    if (current_jac) break;
                    rd_fun_uncounted(&R, ypred, f0); /* J = self.jac(t_new, y_predict) */
                    if (rd_num_jac(&R, ypred, f0, atol)) return -2;
                    have_lu = 0;
                    current_jac = 1;
                }
            }
            if (!converged) {
                h_abs *= 0.5;
                bdf_change_D(D, n, order, 0.5);
                n_equal_steps = 0;
                have_lu = 0;
                st->n_rejected++;
                continue;
            }
            safety = 0.9 * (2 * BDF_NEWTON_MAXITER + 1) / (2 * BDF_NEWTON_MAXITER + n_iter);
            for (int64_t i = 0; i < n; i++) scale[i] = atol + rtol * fabs(ynew[i]);
            error_norm = bdf_norm(d, error_const[order], scale, n);
            if (error_norm > 1) {
                const double sf = safety * pow(error_norm, -1.0 / (order + 1));
                const double factor = (sf > RD_MIN_FACTOR) ? sf : RD_MIN_FACTOR;
                h_abs *= factor;
                bdf_change_D(D, n, order, factor);
                n_equal_steps = 0;
                st->n_rejected++;
                /* (bdf.py:405-406: the LU is NOT reset here) */
            } else {
                accepted = 1;
            }
        }
        if (status != 1) break;
        n_equal_steps++;
        const double t_old = t;
        t = t_new;
        memcpy(y, ynew, sizeof(double) * n);
        S_h_abs = h_abs;
        st->n_accepted++;
        for (int64_t i = 0; i < n; i++) { /* bdf.py:419-422 */
            D[(int64_t)(order + 2) * n + i] = d[i] - D[(int64_t)(order + 1) * n + i];
            D[(int64_t)(order + 1) * n + i] = d[i];
            for (int j = order; j >= 0; j--) D[(int64_t)j * n + i] += D[(int64_t)(j + 1) * n + i];
        }
        if (n_equal_steps >= order + 1) {
            const double em = order > 1 ? bdf_norm(D + (int64_t)order * n, error_const[order - 1], scale, n) : INFINITY;
            const double ep = order < BDF_MAX_ORDER ? bdf_norm(D + (int64_t)(order + 2) * n, error_const[order + 1], scale, n) : INFINITY;
            const double en[3] = {em, error_norm, ep};
            double factors[3];
            int best = 0;
            for (int i = 0; i < 3; i++) {
                factors[i] = pow(en[i], -1.0 / (order + i)); /* inf -> 0, 0 -> inf (errstate ignore) */
                if (factors[i] > factors[best]) best = i; /* np.argmax: the first maximum; NaN handling: see below */
            }
            for (int i = 0; i < 3; i++) /* np.argmax returns the first NaN if there is one */
                if (factors[i] != factors[i]) { best = i; break; }
            order += best - 1;
            const double sf = safety * factors[best];
            const double factor = (sf < RD_MAX_FACTOR) ? sf : RD_MAX_FACTOR; /* python's min(MAX_FACTOR, x): x is returned only if x < MAX_FACTOR, so a NaN x yields MAX_FACTOR (bdf.py:442) */
            S_h_abs *= factor;
            bdf_change_D(D, n, order, factor);
            n_equal_steps = 0;
            have_lu = 0;
        }
        if (t - t1 >= 0) status = 0;
        if (step_times && steps_out < max_steps_out) step_times[steps_out] = t;
        steps_out++;

        /* _dense_output_impl (bdf.py:452-454): built AFTER the order / step-size update, from self.h_abs, self.order, self.D */
        orc_dense dense = {p, &c, N, t, S_h_abs, NULL, NULL, scratch, NULL, D, order};
        orc_events(p, &c, N, y, g_new);
        for (int e = 0; e < MARL_NEVENTS; e++) {
            const int up = (g[e] <= 0) && (g_new[e] >= 0), down = (g[e] >= 0) && (g_new[e] <= 0);
            if (up || down) {
                if (t_events && st->n_events[e] < max_events)
                    t_events[e * max_events + st->n_events[e]] = orc_brent(&dense, e, t_old, t);
                st->n_events[e]++;
            }
            g[e] = g_new[e];
        }
        while (t_eval && eval_i < n_eval && t_eval[eval_i] <= t) {
            orc_dense_eval(&dense, t_eval[eval_i], y_eval + eval_i * n);
            eval_i++;
        }
    }
    st->status = status;
    st->t = t;
    st->h_next = S_h_abs;
    orc_events(p, &c, N, y, st->event_value);
    if (n_steps_out) *n_steps_out = steps_out;
    free(D); free(buf); free(R.J); free(R.factor); free(R.own_groups); free(ab); free(piv);
    return status;
}
