/* marl_hip.h - C ABI of libmarl_hip.so: the MI355X (gfx950) implementation of the reference's hot
 * path - the five-field L'Heureux (2018) RHS and its explicit Runge-Kutta time loops.
 *
 * Every entry point names the reference interface it stands in for (paths relative to the
 * reference repository).  Plain pointers and sizes only; no C++ or torch types cross this ABI.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all floating point data is float64; a state vector is float64[5*N] FIELD-MAJOR
 *     (CA | CC | cCa | cCO3 | Phi, each N contiguous: marlpde/Evolve_scenario.py:64-65,76-86) unless a
 *     `layout` argument says MARL_LAYOUT_TILED (device buffers only);
 *   - functions return 0 when they ran, <0 on error (invalid argument / HIP failure; text from
 *     marl_last_error); how an integration ENDED is in marl_stats.status (0 reached t1, -1 step size
 *     too small = scipy's status -1, 2 attempt budget exhausted) - a failed solve is not an API error,
 *     exactly as solve_ivp reports it in sol.status without raising (scipy/integrate/_ivp/ivp.py:659-661);
 *   - `*_dev` variants take DEVICE pointers (e.g. torch.Tensor.data_ptr()), enqueue on the context's
 *     stream and do not synchronise unless documented; the others take HOST pointers, copy in/out
 *     and return after the stream is idle;
 *   - a context is not thread safe (the reference's RHS object is stateful too:
 *     marlpde/LHeureux_model.py:90,168-172); use one context per thread/GPU/stream;
 *   - NaN/Inf in a state are data, not errors (the reference's numba path behaves the same way).
 */
#ifndef MARL_HIP_H
#define MARL_HIP_H

#include "marl_params.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct marl_ctx marl_ctx;

#define MARL_LAYOUT_FIELD_MAJOR 0 /* y[f*N + i]                                   (reference layout) */
#define MARL_LAYOUT_TILED 1       /* y[(i>>6)*320 + f*64 + (i&63)], ceil(N/64)*320 doubles            */

/* ---- model construction ---------------------------------------------------------------------
 * Replaces LMAHeureuxPorosityDiff.__init__ (marlpde/LHeureux_model.py:12-133): derives the
 * constants (:36-72, :130-133), the boundary rules (:26-30) and the depth mask
 * (marlpde/Evolve_scenario.py:51-54), and uploads them.  `params` holds n_instances blocks
 * (a parameter sweep: one block per instance; the reference runs one instance per process).
 * All instances share N. */
int marl_ctx_create(const marl_params* params, int64_t n_instances, int64_t N, int device, marl_ctx** out);
/* New parameters for an existing context (same N, same number of instances): what constructing the reference's model again with other
 * arguments does (marlpde/LHeureux_model.py:12-133), without giving up the buffers the context has allocated - a loop over scenarios
 * (the reference's tests, marlpde_amd.sweep.run_sweep_bdf) pays for its allocations once. */
int marl_ctx_set_params(marl_ctx* ctx, const marl_params* params, int64_t n_instances);

void marl_ctx_destroy(marl_ctx* ctx);
const char* marl_last_error(const marl_ctx* ctx); /* ctx may be NULL: error of the last failed create */
/* All work of a context is enqueued on one HIP stream: by default the device's default (null) stream -
 * which is also PyTorch's default current stream - or the one given here (a hipStream_t, e.g.
 * torch.cuda.current_stream().cuda_stream; NULL selects the default stream again). */
int marl_set_stream(marl_ctx* ctx, void* hip_stream);
int marl_synchronize(marl_ctx* ctx);
/* Tuning knobs; unknown names are an error.  See DESIGN.md.
 *   rk4_variant, rk45_variant, sweep_variant (kernel shapes; -1 = default), host_layout (device layout used behind the
 *   host-pointer entry points), poll_interval (attempts enqueued between status reads), no_reuse (1: every RHS evaluation of
 *   the fused kernels takes its full transcendental path - the input-independent worst case, for benchmarks), radau_solver
 *   (linear systems of the implicit path: 0 block parallel cyclic reduction - the default; 1 sequential block Thomas, a cross-check that
 *   only lab builds compile in, -DMARL_LAB_BLOCK_THOMAS: an error otherwise),
 *   rk45_stream (the adaptive loop of ONE grid: 1, the default: one launch for a whole batch of attempts - resident workgroups meeting at a
 *   barrier in memory, rk45_stream_kernel - for grids of up to three rounds of resident workgroups (~750 000 cells on an MI355X), one launch
 *   per attempt above; 0 never; 2 always), rk45_stream_attempts (attempts per launch of that loop at most; 4096),
 *   dd_stream (library-side domain-decomposed loop: 0, the default: attempt + reduce + pack launches per attempt; 1 / 2: the slab's attempt as
 *   one launch of the persistent kernel whose last workgroup packs the rank's message - measured not faster),
 *   implicit_zero_copy (1, the default: marl_integrate_radau / marl_integrate_bdf read their per-iteration scalars from coherent host
 *   memory that the kernels write and the host polls; 0: a copy and a stream synchronisation per read - bit-identical),
 *   radau_fused_solve (systems of up to 2048 unknowns: 0 = one launch per cyclic-reduction level; 1 = every level of a solve in one
 *   launch; 2 = a Radau Newton iteration's linear algebra in one workgroup; 3, the default = 1 plus, for single Radau runs, two
 *   launches per Newton iteration - all bit-identical),
 *   radau_sweep_wg (Radau sweeps of grids of up to 409 cells: 3, the default: a persistent workgroup per instance runs the instance's
 *   sequential work and its Jacobians, launch kernels over work lists do the factorisations; 1: Jacobians by launch kernels too -
 *   bit-identical; 2: everything in the workgroup; 0: one launch cycle per action),
 *   radau_cr (single Radau / BDF runs: levels of block cyclic reduction in front of the parallel cyclic reduction; -1, the default:
 *   automatic for grids of radau_cr_min_n (2048) cells or more, down to a compact system of at most 204 rows; 0: none; k > 0: k levels),
 *   radau_cr_small / radau_cr_small_min_n (grids solved in one workgroup, up to 409 cells: 3 levels of cyclic reduction in front of PCR
 *   inside the one-launch solves, by default from 205 cells; min_n = 32 turns it on for the reference's N = 200 - faster sweeps, but no
 *   longer decision-by-decision equal to scipy in every pinned case),
 *   radau_cr_tail (1, the default: the launch-bound levels of such a solve in one launch each way; 0: one launch per level -
 *   bit-identical), bdf_solve_wg (1, the default: on grids of up to 409 cells marl_integrate_bdf runs solve_bdf_system as one launch of
 *   one workgroup; 0: one RHS launch, one linear-algebra launch and one wait per Newton iteration - bit-identical),
 *   rk4_stream (fixed-step RK4 of one grid as ONE dataflow launch over (level, tile) work items instead of one launch per
 *   fused level: 0 never, 1 - the default - for grids of 196 608 cells or more, 2 always; results are bit-identical),
 *   rk4_stream_third (1, the default: a streamed call with an odd number of levels runs through a third state buffer instead of
 *   copying the state back afterwards; 0: copy - bit-identical),
 *   rk4_stream_test_raise (test hook: the next streamed run starts with its give-up flag raised - marl_synchronize must
 *   report error -2 and the context must recover), rk4_stream_max_items (test hook: work items per streamed launch, default
 *   2^31 - 1: a call with more (level, tile) items is split into several launches of an even number of levels). */
int marl_set_option(marl_ctx* ctx, const char* name, int64_t value);
/* Derived constants of instance `inst` in the order of tests/golden/derived_constants.json:
 * delta_x nu1 nu2 KRat dCa dCO3 delta Da lambda_ auxcon rhorat0 rhorat presum F_fixed dPhi_fixed
 * Peclet_min Peclet_max, then mask_lo, mask_hi (as doubles). */
int marl_get_constants(const marl_ctx* ctx, int64_t inst, double out[19]);
/* doubles needed for one instance's state in `layout` */
int64_t marl_state_doubles(const marl_ctx* ctx, int layout);

/* ---- RHS ---------------------------------------------------------------------------------------
 * Replaces the solve_ivp callable  fun(t, y, progress_proxy, progress_dt, t0) / fun_numba(...)
 * (marlpde/LHeureux_model.py:162-288, :290-359 -> pde_rhs :361-522).  `t` is accepted and ignored
 * (the system is autonomous; the reference only uses t for its progress bar).  y and dydt must not
 * alias.  For n_instances > 1 the buffers hold the instances one after another. */
int marl_rhs(marl_ctx* ctx, double t, const double* y, double* dydt);
int marl_rhs_dev(marl_ctx* ctx, double t, const double* y_dev, double* dydt_dev, int layout);

/* ---- monitors ----------------------------------------------------------------------------------
 * Replaces the seven event functions zeros, zeros_CA, zeros_CC, ones_CA_plus_CC, ones_Phi,
 * zeros_U, zeros_W (marlpde/LHeureux_model.py:524-593), in that order.  out: [n_instances][7] (host). */
int marl_events(marl_ctx* ctx, const double* y, double* out);
int marl_events_dev(marl_ctx* ctx, const double* y_dev, int layout, double* out); /* synchronises */

int marl_convert_layout_dev(marl_ctx* ctx, const double* src_dev, double* dst_dev, int src_layout, int dst_layout);

/* Test hook: y[i] = op(x[i]) with the kernels' own math primitives: op 0 log, 1 exp, 2 pow(x, e),
 * 3 reciprocal, 4 Fiadeiro-Veronis sigma(Pe = x, W = e).  Device pointers; asynchronous. */
int marl_debug_math(marl_ctx* ctx, int op, const double* x_dev, double* y_dev, int64_t n, double e);

/* ---- fixed-step classical RK4 (BASELINE config 2; no counterpart in the reference, which only
 * remarks that forward Euler fails: README.md:9).  y is advanced in place by nsteps steps of dt. */
int marl_integrate_rk4(marl_ctx* ctx, double* y, double dt, int64_t nsteps);
int marl_integrate_rk4_dev(marl_ctx* ctx, double* y_dev, int layout, double dt, int64_t nsteps);
/* batched sweep: one dt per instance (host array, n_instances entries); FIELD-MAJOR device state */
int marl_sweep_rk4_dev(marl_ctx* ctx, double* y_dev, const double* dt, int64_t nsteps);

/* ---- adaptive Dormand-Prince RK45 ---------------------------------------------------------------
 * Replaces  scipy.integrate.solve_ivp(fun, t_span, y0, method="RK45", first_step=, rtol=, atol=,
 * t_eval=, events=[7 monitors])  as called at marlpde/Evolve_scenario.py:104-109 (controller:
 * scipy/integrate/_ivp/rk.py:111-176; driver: ivp.py:654-723).  y: in y(t0), out y(stats->t).
 * t_eval (may be NULL): sorted sample times within [t0, t1]; y_eval receives n_eval x 5N doubles,
 * sample-major (dense output, rk.py:560-574).  t_events (may be NULL): 7 x max_events root times of the
 * monitors' sign changes, located like scipy does (dense output + Brent, ivp.py:51-76); the counts are
 * stats->n_events.  max_attempts = 0: unlimited. */
int marl_integrate_rk45(marl_ctx* ctx, double* y, double t0, double t1, double first_step, double rtol, double atol,
                        const double* t_eval, int64_t n_eval, double* y_eval, double* t_events, int64_t max_events,
                        int64_t max_attempts, marl_stats* stats);
int marl_integrate_rk45_dev(marl_ctx* ctx, double* y_dev, int layout, double t0, double t1, double first_step,
                            double rtol, double atol, int64_t max_attempts, marl_stats* stats); /* synchronises */
/* batched sweep: per-instance controller; all instances share t0, t1, first_step, tolerances.
 * stats: host array of n_instances entries.  FIELD-MAJOR device state.  Synchronises. */
int marl_sweep_rk45_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol,
                        double atol, int64_t max_attempts, marl_stats* stats);

/* ---- implicit Radau IIA (order 5): the reference's DEFAULT solver ------------------------------------------------
 * Replaces  scipy.integrate.solve_ivp(fun, t_span, y0, method="Radau", jac_sparsity=jacobian_sparsity(), first_step=,
 * rtol=, atol=, t_eval=, events=[7 monitors])  as called at marlpde/Evolve_scenario.py:104-109 with the defaults of
 * marlpde/parameters.py:201-240 (method "Radau" :213; the 27-diagonal sparsity pattern :150-199).  Step logic:
 * scipy/integrate/_ivp/radau.py; Jacobian: scipy's finite-difference num_jac (common.py:268-451) over column groups, with
 * the reference's pattern (rows (f, i) x columns (f', i-1..i+1) minus CA/CC rows x Phi columns) - evaluated on the device,
 * as are the block-tridiagonal factorisations of the real and the complex collocation system and every vector operation.
 * groups (may be NULL): int32[5N], the column grouping scipy derives from the pattern (scipy.optimize._numdiff.group_columns);
 * NULL selects a structured 15-colouring - the Jacobian entries are the same either way (and scipy's nfev does not count the
 * finite-difference columns).  stats->nfev / njev / nlu are counted as scipy counts them.  Other arguments as for
 * marl_integrate_rk45.  Single-instance contexts only. */
int marl_integrate_radau(marl_ctx* ctx, double* y, double t0, double t1, double first_step, double rtol, double atol,
                         const int32_t* groups, const double* t_eval, int64_t n_eval, double* y_eval, double* t_events,
                         int64_t max_events, int64_t max_attempts, marl_stats* stats);
/* The same with scipy's BDF (scipy/integrate/_ivp/bdf.py: variable order 1..5, quasi-constant step NDF) - the other implicit method
 * the reference's Solver names for its jac_sparsity (marlpde/parameters.py:205-219); replaces
 * solve_ivp(fun, method="BDF", jac_sparsity=...) at marlpde/Evolve_scenario.py:104-109.  The matrix factorised is I - c J (one real
 * block-tridiagonal system per step size / order change, by cyclic reduction); arguments and statistics as for marl_integrate_radau. */
int marl_integrate_bdf(marl_ctx* ctx, double* y, double t0, double t1, double first_step, double rtol, double atol,
                         const int32_t* groups, const double* t_eval, int64_t n_eval, double* y_eval, double* t_events,
                         int64_t max_events, int64_t max_attempts, marl_stats* stats);

/* A SWEEP of Radau integrations (the reference integrates one scenario per process with this solver; its tests loop over
 * scenarios: tests/Regression_test/test_regression.py:44,74,115-116): every instance of the context with its own parameters, step-size
 * history and Newton convergence, advanced together - each instance's step logic runs as a state machine on the device, every
 * launch works on all instances that currently need that kind of work (csrc/marl_radau_batch.h).  y_dev: [n_instances][5N] FIELD-MAJOR
 * device states, advanced in place; stats: host array of n_instances entries (nfev / njev / nlu / status / t as scipy counts them;
 * monitor sign changes are counted in n_events, root times are not located in a sweep).  N <= 1638.  Synchronises. */
int marl_sweep_radau_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol, double atol,
                         const int32_t* groups, int64_t max_attempts, marl_stats* stats);
/* The same sweep with the monitors' ROOT TIMES located inside it, as solve_ivp returns them for every run (sol.t_events,
 * marlpde/Evolve_scenario.py:118-145, 175-177; scipy ivp.py:673-694 with solve_event_equation :51-76): after an accepted step in
 * which a monitor changed sign, the instance's device-side controller runs Brent's method on the step's dense output - one
 * dense-output evaluation + monitors per function evaluation.  t_events: host array [instances][7][max_events]; entry k of
 * monitor e of instance b is valid for k < stats[b].n_events[e], the rest is NaN; sign changes beyond max_events are counted only. */
int marl_sweep_radau_events_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol, double atol,
                                const int32_t* groups, int64_t max_attempts, double* t_events, int64_t max_events, marl_stats* stats);

/* ---- 1-D domain decomposition of ONE large grid (BASELINE config 5; the reference never decomposes the depth
 * axis).  One process per GPU holds a slab [g_begin, g_end) of the N_global cells in a slab context; the host
 * layer moves halo strips between neighbours and the per-rank reduction records to everybody (RCCL
 * send/recv + all-gather on the context's stream) between these calls - see domain.py.  Slab state lives in the
 * context (FIELD-MAJOR, `halo` >= 6 extra cells on each interior side).  Strips are 2 x 5 x halo doubles:
 * [y | f][field][cell].  `which`: 0/1 an explicit state buffer, -1 the buffer the attempt in flight wrote,
 * -2 the current one.  Sequence:
 *   load -> pack(0) <-> unpack(0) -> rhs0 -> pack(0) <-> unpack(0) -> monitors -> [all-gather] -> init_control
 *   per attempt: attempt -> pack(-1) <-> unpack(-1) -> [all-gather records] -> control;   finally store.
 * Every rank feeds the SAME gathered records (in rank order) to init_control / control, so all ranks take
 * bit-identical accept/reject decisions without further synchronisation. */
int marl_ctx_create_slab(const marl_params* params, int64_t N_global, int64_t g_begin, int64_t g_end, int64_t halo,
                         int device, marl_ctx** out);
int marl_slab_load(marl_ctx* ctx, const double* y_owned_dev);  /* [5][g_end - g_begin] */
int marl_slab_store(marl_ctx* ctx, double* y_owned_dev);
int marl_slab_pack(marl_ctx* ctx, int which, double* send_lo_dev, double* send_hi_dev);
int marl_slab_unpack(marl_ctx* ctx, int which, const double* recv_lo_dev, const double* recv_hi_dev);
int marl_slab_rhs0(marl_ctx* ctx);                             /* f(t0, y0) on the owned cells */
int marl_slab_monitors(marl_ctx* ctx, double* rec_dev);        /* 8-double record of y0's monitors (owned cells) */
int marl_slab_init_control(marl_ctx* ctx, const double* recs_dev, int64_t nrec, double t0, double t1,
                           double first_step, double rtol, double atol, int64_t max_attempts);
int marl_slab_attempt(marl_ctx* ctx, double* rec_dev);         /* one fused Dormand-Prince attempt + its record */
int marl_slab_control(marl_ctx* ctx, const double* recs_dev, int64_t nrec);
int marl_slab_status(marl_ctx* ctx, marl_stats* stats);        /* synchronises */
/* The same loop with the exchange INSIDE the library (no host language in the per-attempt path): the ranks' messages
 * [record (8) | lower strip | upper strip] travel in one ncclAllGather on the context's stream.  The RCCL API is taken by
 * dlopen from the librccl the process already has (`rccl_path`, e.g. the one PyTorch-ROCm bundles; NULL: the loader's
 * default) - this library does not link RCCL.  Rank 0 makes the id (marl_slab_comm_id) and the host layer sends it to every
 * rank (any transport); comm_init with world = 1 and id = NULL runs one slab with no communicator.
 * Sequence:  comm_init -> load -> exchange(0) -> rhs0 -> monitors(NULL) -> exchange(0) -> init_control(NULL, world, ...) -> run
 * -> store.  marl_slab_run enqueues  attempt -> reduce + pack -> all-gather -> unpack + control  `poll_interval` attempts at a
 * time (never more than the attempt budget has left) and reads the status once per batch.
 * ncclCommInitRank is collective: marl_slab_comm_probe checks everything that is local (librccl loadable, entry points
 * present) so that the host layer can agree on the transport BEFORE any rank enters it; comm_init rejects an id that
 * marl_slab_comm_id cannot have produced (all zero) without touching RCCL. */
int marl_slab_comm_probe(const char* rccl_path);
int marl_slab_comm_id(const char* rccl_path, char id_out[128]);
int marl_slab_comm_init(marl_ctx* ctx, const char* rccl_path, const char id[128], int rank, int world);
int marl_slab_exchange(marl_ctx* ctx, int which);
int marl_slab_run(marl_ctx* ctx, marl_stats* stats);           /* synchronises */

#ifdef __cplusplus
}
#endif
#endif /* MARL_HIP_H */
