// marl_api.hip - host side of libmarl_hip.so: context management, constant derivation and kernel
// launches behind the C ABI declared in include/marl_hip.h.  gfx950 only; no CPU fallback - every
// entry point needs a HIP device and fails with an error text otherwise.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/marl_hip.h"
#include "marl_kernels.h"
#include "marl_radau.h"
#include "marl_radau_cr.h"
#include "marl_radau_batch.h"
#include "marl_radau_wg.h"
#include "marl_bdf.h"
#include "marl_bdf_wg.h"

using namespace marl;

static_assert(MARL_NFIELDS == NF, "field count");
static thread_local std::string g_create_error;   // marl_ctx_create failures, per calling thread (contexts are created from worker threads too)

struct marl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // nullptr = the device's default (null) stream, which is also torch's default current stream
    int64_t N = 0, batch = 0;
    Slab slab{};
    int halo = 0;  // > 0: a slab context of a domain-decomposed grid
    bool var_dphi = false;  // marl_params.dPhi_variable (the same for every instance): selects the VD kernel instantiations
    std::vector<marl_params> params;
    std::vector<DevConsts> hconsts;
    std::vector<double> extra;  // per instance: delta_x, auxcon, rhorat0, F_fixed (not needed on device)
    DevConsts* dconsts = nullptr;
    // device scratch (grown on demand)
    double* buf[4] = {nullptr, nullptr, nullptr, nullptr};  // Y0, Y1, F0, F1 (or staging)
    size_t buf_cap[4] = {0, 0, 0, 0};
    double* part = nullptr;
    size_t part_cap = 0;
    // streamed RK4 (rk4_stream_kernel): item counter (+ one spare word) + one published-level counter per tile; never reset
    // between launches - sq_item_base / sq_level_base say where they stand
    unsigned* sq = nullptr;
    size_t sq_cap = 0;
    unsigned sq_item_base = 0, sq_level_base = 0;
    bool sq_test_raise = false;    // test hook (option rk4_stream_test_raise): the next streamed run starts with the flag raised
    int64_t sq_max_items = 0x7fffffff;   // items per launch (the counter is 32 bits wide); test hook rk4_stream_max_items lowers it
    unsigned* sq_sticky = nullptr; // device: raised by a streamed run that gave up waiting; cleared by the host only
    unsigned* sq_host = nullptr;   // pinned: copy of sq_sticky
    bool sq_pending = false;       // streamed runs since sq_sticky was last looked at
    int cus = 0;
    // persistent adaptive loop (rk45_stream_kernel): barrier / publication words, one record per resident workgroup; the ticket and
    // barrier counters are never reset between launches - rs_arrive_base / rs_epoch_base say where they stand
    Rk45Stream* rs = nullptr;
    double* rs_grec = nullptr;
    unsigned* rs_host = nullptr;   // pinned: copy of {arrive, sticky}
    unsigned rs_arrive_base = 0, rs_epoch_base = 0, rs_grab_base = 0, rs_G = 0, rs_grabs_per_attempt = 0;
    int rs_occ[3][2] = {{0, 0}, {0, 0}, {0, 0}};   // resident workgroups per CU of the instantiation [field-major / tiled / one attempt per launch][VD]
    int64_t rk45_stream = 1;       // the adaptive loop of one grid as ONE launch per batch of attempts (rk45_stream_kernel): 0 never, 1 where it is faster (grids of up to kRk45StreamRounds rounds of resident workgroups), 2 always
    int64_t dd_stream = 0;         // domain-decomposed loop inside the library: 0 (default) attempt + reduce + pack launches; 1 / 2: the slab's attempt as ONE launch of the persistent kernel whose last workgroup packs the rank's message (1: slabs of up to kRk45StreamRounds rounds, 2: always) - built for the 8-GPU shard size and measured there NOT faster (44.5 us against 34.9 + 4.7 + 4.7 us, profiles/r04_lab_rk45_stream.log)
    int64_t rk45_stream_attempts = 4096;   // attempts per launch at most (the host looks at the status and the sticky flag in between)
    double* rec = nullptr;  // [batch][NQ]
    Rk45Ctrl* dctrl = nullptr;
    Rk45Ctrl* hctrl = nullptr;  // pinned
    double* hrec = nullptr;     // pinned [batch][NQ]
    double* ddt = nullptr;      // [batch] per-instance dt
    double* hdt = nullptr;      // pinned [batch]: staging of the caller's dt array (the caller's buffer may die before the copy runs)
    // domain decomposition: this rank's message / the gathered messages (device), and the RCCL communicator (dlopen'ed API)
    double* dd_send = nullptr;
    double* dd_gathered = nullptr;
    double* dd_recs = nullptr;
    int dd_rank = 0, dd_world = 1;
    int64_t dd_max_attempts = 0;   // the attempt budget given to marl_slab_init_control (marl_slab_run sizes its batches by it)
    void* rccl_lib = nullptr;
    void* rccl_comm = nullptr;
    int (*rccl_allgather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*rccl_destroy)(void*) = nullptr;
    // implicit (Radau) path: one device arena + a small pinned read-back area
    double* rd_arena = nullptr;
    size_t rd_cap = 0;
    double* rd_host = nullptr;  // pinned: [0] norm^2, [1] flags (as int32), [2..] spare
    // zero-copy result words of the single-instance implicit drivers: coherent host memory the kernels write their scalar results
    // into and the host polls - no copy, no stream synchronisation per Newton iteration.  [0] a sum of squares, [1] the non-finite flag
    // (int32), [8..15] a monitors record.  A slot is "armed" with a NaN of a payload no computation produces until its kernel writes it.
    double* zc_h = nullptr;     // host view
    double* zc_d = nullptr;     // device view of the same memory
    bool zc_on = false;         // the run in progress uses it
    // options
    int64_t rk4_variant = -1, rk45_variant = -1, sweep_variant = -1, host_layout = LAYOUT_TILED, poll = 64;
    int64_t rk4_stream = 1;     // the fixed-step loop of one grid as ONE dataflow launch (rk4_stream_kernel): 0 never, 1 large grids, 2 always
    int64_t rk4_small = 1;      // grids of up to two workgroups per CU at 16 steps per launch: the build of marl_rk4_small.hip (0: the common one)
    int64_t rk4_stream_third = 1;   // an odd number of levels goes through a third state buffer, so that no whole-state copy follows (0: copy)
    double* stream_c = nullptr;
    size_t stream_c_cap = 0;
    int64_t implicit_zero_copy = 1;  // scalar results of the implicit drivers through polled host memory (0: copy + synchronise)
    int64_t radau_fused_solve = 3;   // small systems (5 N <= 2048): >= 1: all levels of a solve in one launch (BDF: the whole Newton iteration / solve_bdf_system); Radau single runs: 3 (default) = two launches per Newton iteration (solves with their own right-hand sides | update + norm + the next stage derivatives), 2 = the whole iteration's linear algebra in ONE workgroup (bit-identical, measured slower: the two chains then share one compute unit), 1 = four launches; 0: one launch per level
    int64_t radau_solver = 0;   // 0: block parallel cyclic reduction (parallel over depth); 1: sequential block Thomas
    int64_t radau_cr = -1;      // single runs: levels of cyclic reduction in front of PCR; -1 = automatic (grids of >= radau_cr_min_n cells: down to a
                                // compact system that fits the one-launch solve with one unknown per thread), 0 = none
    int64_t radau_cr_min_n = 2048;
    int64_t radau_cr_small = 3; // grids solved in one workgroup (up to 409 cells): levels of cyclic reduction in front of PCR (0: none)
    int64_t radau_cr_small_min_n = 205;   // ... from this many cells on (below: plain PCR, which reproduces scipy's decisions on the reference's N = 200 runs)
    int64_t bdf_solve_wg = 1;   // small grids: solve_bdf_system as one launch of one workgroup (marl_bdf_wg.h); 0: one launch + wait per Newton iteration
    int64_t radau_cr_tail = 1;  // the launch-bound levels of a cyclic-reduction solve in one launch each way (0: one launch per level)
    int64_t radau_sweep_wg = 3; // sweeps of small grids: 3 hybrid (workgroup per instance for the sequential work and Jacobians, launch kernels for factorisations), 1 hybrid with launch kernels for Jacobians too, 2 all in the workgroup, 0 launch per action
    std::string err;
};

static int fail(marl_ctx* ctx, int code, const char* fmt, ...)
{
    char msg[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(msg, sizeof msg, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_OK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) return fail(ctx, -100 - (int)e_, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

#define LAUNCH_OK(ctx)                                                                            \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(ctx, -100 - (int)e_, "kernel launch failed: %s", hipGetErrorString(e_)); \
    } while (0)

#ifdef MARL_LAB_CLOCK45
extern "C" int marl_lab_read_clock45(unsigned long long* dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(marl::marl_lab_clock45), n * sizeof(unsigned long long));
}
extern "C" int marl_lab_read_clock45t(unsigned long long* dst, size_t n)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(marl::marl_lab_clock45t), n * sizeof(unsigned long long));
}
#endif

// ----------------------------------------------------------------------------------------------
// Derived constants: LMAHeureuxPorosityDiff.__init__, marlpde/LHeureux_model.py:23-24, :36-72,
// :87-88, :130-133; depth mask: marlpde/Evolve_scenario.py:51-54.
// ----------------------------------------------------------------------------------------------
static void derive_consts(const marl_params& p, int64_t N, DevConsts& c, double extra[4])
{
    const double g = 100 * 9.81;
    const double dx = p.length / (double)N;
    const double delta_x = (0.0 + 1.5 * dx) - (0.0 + 0.5 * dx);  // x[1] - x[0] of the cell centres (:23-24)
    const double dCa = p.DCa / p.D0Ca, dCO3 = p.DCO3 / p.D0Ca;
    const double auxcon = p.beta / (p.D0Ca * p.b * g * p.rhow * (p.PhiNR - p.PhiInfty));
    const double rhorat0 = (p.rhos0 / p.rhow - 1) * p.beta / p.sedimentationrate;
    const double F_fixed = 1 - std::exp(10 - 10 / p.PhiIni);
    memset(&c, 0, sizeof c);
    HotConsts& k = c.hot;
    k.inv_dx = 1.0 / dx;
    k.inv_dx2 = std::pow(dx, -2);
    k.dCa = dCa;
    k.dCO3 = dCO3;
    k.dPhi = auxcon * F_fixed * std::pow(p.PhiIni, 3) / (1 - p.PhiIni);
    k.pe_cCa = delta_x / (2. * dCa);
    k.pe_cCO3 = delta_x / (2. * dCO3);
    k.pe_Phi = delta_x / (2. * k.dPhi);
    k.presum = 1 - rhorat0 * std::pow(p.Phi0, 3) * (1 - std::exp(10 - 10 / p.Phi0)) / (1 - p.Phi0);
    k.rhorat = (p.rhos / p.rhow - 1) * p.beta / p.sedimentationrate;
    k.KRat = p.KC / p.KA;
    k.nu1 = p.k1 / p.k2;
    k.nu2 = p.k4 / p.k3;
    k.m1 = p.m1; k.m2 = p.m2; k.n1 = p.n1; k.n2 = p.n2;
    c.p0_m1 = std::pow(0.0, p.m1); c.p0_m2 = std::pow(0.0, p.m2);
    c.p0_n1 = std::pow(0.0, p.n1); c.p0_n2 = std::pow(0.0, p.n2);
    // exponents <= 0: pow(0, e) != 0, the clamp pairs need the general combination; exponents < 1: the stage-reuse
    // expansions' range check (|u| <= |n u|) does not hold - both take the general, cache-less evaluation
    k.generic_p0 = (c.p0_m1 != 0 || c.p0_m2 != 0 || c.p0_n1 != 0 || c.p0_n2 != 0 || !(p.m1 >= 1) || !(p.m2 >= 1) || !(p.n1 >= 1) || !(p.n2 >= 1)) ? 1 : 0;
    k.lambda_ = p.k3 / p.k2;
    k.Da = p.k2 * p.Tstar;
    k.delta = p.rhos / (p.muA * std::sqrt(p.KC));
    k.hdx = 0.5 * k.inv_dx;
    k.pe_smax = k.pe_cCa > k.pe_cCO3 ? k.pe_cCa : k.pe_cCO3;
    k.Dal = k.Da * k.lambda_;
    k.Da_nu1 = k.Da * k.nu1;
    k.Dal_nu2 = k.Dal * k.nu2;
    k.rr10 = 10.0 * k.rhorat;
    k.dPhi_dx2 = k.dPhi * k.inv_dx2;
    k.auxcon = auxcon;
    k.var_dphi = p.dPhi_variable ? 1 : 0;
    const double bc[NF] = {p.CA0, p.CC0, p.cCa0, p.cCO30, p.Phi0};
    for (int f = 0; f < NF; f++) c.bc[f] = bc[f];
    c.N = N;
    k.fv = p.FV_switch;
    // mask = H(x - shallow) * H(deep - x) with H(0) = 0 at the cell centres: a contiguous index range
    int64_t lo = N, hi = N;
    bool open = false;
    for (int64_t i = 0; i < N; i++) {
        const double x = 0.0 + ((double)i + 0.5) * dx;
        const bool in = (x - p.shallow_limit > 0) && (p.deep_limit - x > 0);
        if (in && !open) { lo = i; open = true; }
        if (!in && open) { hi = i; break; }
    }
    if (!open) lo = hi = 0;
    c.mask_lo = lo;
    c.mask_hi = hi;
    extra[0] = delta_x; extra[1] = auxcon; extra[2] = rhorat0; extra[3] = F_fixed;
}

static int ensure(marl_ctx* ctx, int which, size_t doubles)
{
    if (ctx->buf_cap[which] >= doubles) return 0;
    if (ctx->buf[which]) HIP_OK(ctx, hipFree(ctx->buf[which]));
    ctx->buf[which] = nullptr;
    ctx->buf_cap[which] = 0;
    HIP_OK(ctx, hipMalloc((void**)&ctx->buf[which], doubles * sizeof(double)));
    ctx->buf_cap[which] = doubles;
    return 0;
}

static int ensure_part(marl_ctx* ctx, size_t records)
{
    if (ctx->part_cap >= records) return 0;
    if (ctx->part) HIP_OK(ctx, hipFree(ctx->part));
    ctx->part = nullptr;
    ctx->part_cap = 0;
    HIP_OK(ctx, hipMalloc((void**)&ctx->part, records * NQ * sizeof(double)));
    ctx->part_cap = records;
    return 0;
}

// validate_tol (scipy/integrate/_ivp/common.py:44-51): rtol below 100 eps is raised to it
static double clamp_rtol(double rtol) { const double lo = 100 * 2.220446049250313e-16; return rtol < lo ? lo : rtol; }

static int64_t state_doubles(int64_t n, int layout)
{
    return layout == LAYOUT_TILED ? ((n + 63) / 64) * (int64_t)(NF * 64) : (int64_t)NF * n;
}

extern "C" {

int marl_ctx_create(const marl_params* params, int64_t n_instances, int64_t N, int device, marl_ctx** out)
{
    if (!params || !out || n_instances < 1 || N < 2) return fail(nullptr, -1, "marl_ctx_create: invalid argument");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, -2, "marl_ctx_create: no HIP device (%s); this library has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, -1, "marl_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
    marl_ctx* ctx = new marl_ctx;
    ctx->device = device;
    ctx->N = N;
    ctx->batch = n_instances;
    ctx->slab = Slab{N, 0, N, 0, N};
    ctx->params.assign(params, params + n_instances);
    ctx->hconsts.resize(n_instances);
    ctx->extra.resize(4 * n_instances);
    for (int64_t b = 0; b < n_instances; b++) {
        const marl_params& p = params[b];
        if (!(p.length > 0)) { delete ctx; return fail(nullptr, -1, "marl_ctx_create: instance %lld: length must be > 0", (long long)b); }
        if ((p.dPhi_variable != 0) != (params[0].dPhi_variable != 0)) {
            delete ctx;
            return fail(nullptr, -1, "marl_ctx_create: dPhi_variable must be the same for every instance of a sweep");
        }
        derive_consts(p, N, ctx->hconsts[b], &ctx->extra[4 * b]);
    }
    ctx->var_dphi = params[0].dPhi_variable != 0;
#define CREATE_OK(call)                                                                     \
    do {                                                                                    \
        hipError_t e2_ = (call);                                                            \
        if (e2_ != hipSuccess) {                                                            \
            fail(nullptr, -100 - (int)e2_, "%s failed: %s", #call, hipGetErrorString(e2_)); \
            marl_ctx_destroy(ctx);                                                          \
            return -100 - (int)e2_;                                                         \
        }                                                                                   \
    } while (0)
    CREATE_OK(hipSetDevice(device));
    CREATE_OK(hipMalloc((void**)&ctx->dconsts, sizeof(DevConsts) * n_instances));
    CREATE_OK(hipMemcpy(ctx->dconsts, ctx->hconsts.data(), sizeof(DevConsts) * n_instances, hipMemcpyHostToDevice));
    CREATE_OK(hipMalloc((void**)&ctx->rec, sizeof(double) * NQ * n_instances));
    CREATE_OK(hipMalloc((void**)&ctx->dctrl, sizeof(Rk45Ctrl) * n_instances));
    CREATE_OK(hipMalloc((void**)&ctx->ddt, sizeof(double) * n_instances));
    CREATE_OK(hipHostMalloc((void**)&ctx->hctrl, sizeof(Rk45Ctrl) * n_instances, hipHostMallocDefault));
    CREATE_OK(hipHostMalloc((void**)&ctx->hrec, sizeof(double) * NQ * n_instances, hipHostMallocDefault));
    CREATE_OK(hipHostMalloc((void**)&ctx->hdt, sizeof(double) * n_instances, hipHostMallocDefault));
#undef CREATE_OK
    *out = ctx;
    return 0;
}

// New parameters for an existing context (same N, same number of instances): the derived constants are recomputed and uploaded; every
// buffer the context has grown stays (a loop over scenarios pays for its allocations once).  The time-varying dPhi switch may change.
int marl_ctx_set_params(marl_ctx* ctx, const marl_params* params, int64_t n_instances)
{
    if (!ctx || !params) return ctx ? fail(ctx, -1, "marl_ctx_set_params: invalid argument") : -1;
    if (n_instances != ctx->batch) return fail(ctx, -1, "marl_ctx_set_params: the context holds %lld instance(s)", (long long)ctx->batch);
    if (ctx->halo > 0) return fail(ctx, -1, "marl_ctx_set_params: not for slab contexts");
    for (int64_t b = 0; b < n_instances; b++) {
        if (!(params[b].length > 0)) return fail(ctx, -1, "marl_ctx_set_params: instance %lld: length must be > 0", (long long)b);
        if ((params[b].dPhi_variable != 0) != (params[0].dPhi_variable != 0))
            return fail(ctx, -1, "marl_ctx_set_params: dPhi_variable must be the same for every instance of a sweep");
    }
    HIP_OK(ctx, hipSetDevice(ctx->device));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));   // nothing in flight reads the old constants
    ctx->params.assign(params, params + n_instances);
    const int no_reuse = ctx->hconsts[0].hot.no_reuse;   // (an option of the context, kept)
    for (int64_t b = 0; b < n_instances; b++) {
        derive_consts(params[b], ctx->N, ctx->hconsts[b], &ctx->extra[4 * b]);
        ctx->hconsts[b].hot.no_reuse = no_reuse;
    }
    ctx->var_dphi = params[0].dPhi_variable != 0;
    HIP_OK(ctx, hipMemcpyAsync(ctx->dconsts, ctx->hconsts.data(), sizeof(DevConsts) * n_instances, hipMemcpyHostToDevice, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

void marl_ctx_destroy(marl_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 4; i++)
        if (ctx->buf[i]) (void)hipFree(ctx->buf[i]);
    if (ctx->part) (void)hipFree(ctx->part);
    if (ctx->stream_c) (void)hipFree(ctx->stream_c);
    if (ctx->sq) (void)hipFree(ctx->sq);
    if (ctx->sq_sticky) (void)hipFree(ctx->sq_sticky);
    if (ctx->sq_host) (void)hipHostFree(ctx->sq_host);
    if (ctx->rs) (void)hipFree(ctx->rs);
    if (ctx->rs_grec) (void)hipFree(ctx->rs_grec);
    if (ctx->rs_host) (void)hipHostFree(ctx->rs_host);
    if (ctx->rec) (void)hipFree(ctx->rec);
    if (ctx->dctrl) (void)hipFree(ctx->dctrl);
    if (ctx->ddt) (void)hipFree(ctx->ddt);
    if (ctx->dconsts) (void)hipFree(ctx->dconsts);
    if (ctx->hctrl) (void)hipHostFree(ctx->hctrl);
    if (ctx->hrec) (void)hipHostFree(ctx->hrec);
    if (ctx->hdt) (void)hipHostFree(ctx->hdt);
    if (ctx->rccl_comm && ctx->rccl_destroy) (void)ctx->rccl_destroy(ctx->rccl_comm);
    if (ctx->dd_send) (void)hipFree(ctx->dd_send);
    if (ctx->dd_gathered) (void)hipFree(ctx->dd_gathered);
    if (ctx->dd_recs) (void)hipFree(ctx->dd_recs);
    if (ctx->rd_arena) (void)hipFree(ctx->rd_arena);
    if (ctx->rd_host) (void)hipHostFree(ctx->rd_host);
    if (ctx->zc_h) (void)hipHostFree(ctx->zc_h);
    delete ctx;
}

const char* marl_last_error(const marl_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int marl_set_stream(marl_ctx* ctx, void* hip_stream)
{
    if (!ctx) return -1;
    ctx->stream = (hipStream_t)hip_stream;
    return 0;
}

// A streamed RK4 run (rk4_stream_kernel) that gave up waiting raises a sticky device flag; the device entry points stay
// asynchronous, so the flag is looked at where the caller synchronises anyway: marl_synchronize and the host-pointer entry points.
static int stream_check(marl_ctx* ctx)
{
    if (!ctx->sq_pending) return 0;
    HIP_OK(ctx, hipMemcpyAsync(ctx->sq_host, ctx->sq_sticky, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sq_pending = false;
    if (*ctx->sq_host) {
        HIP_OK(ctx, hipMemsetAsync(ctx->sq_sticky, 0, sizeof(unsigned), ctx->stream));
        if (ctx->sq) HIP_OK(ctx, hipMemsetAsync(ctx->sq, 0, ctx->sq_cap * sizeof(unsigned), ctx->stream));
        ctx->sq_item_base = ctx->sq_level_base = 0;
        return fail(ctx, -2, "rk4: the streamed time loop gave up waiting for a neighbouring tile (rk4_stream_kernel); the state is invalid");
    }
    return 0;
}

int marl_synchronize(marl_ctx* ctx)
{
    if (!ctx) return -1;
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return stream_check(ctx);
}

int marl_set_option(marl_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return -1;
    const std::string n(name);
    if (n == "rk4_variant") ctx->rk4_variant = value;
    else if (n == "rk45_variant") ctx->rk45_variant = value;
    else if (n == "sweep_variant") ctx->sweep_variant = value;
    else if (n == "host_layout") ctx->host_layout = value ? LAYOUT_TILED : LAYOUT_FIELD_MAJOR;
    else if (n == "poll_interval") ctx->poll = value > 0 ? value : 1;
    else if (n == "radau_solver") {
#ifdef MARL_LAB_BLOCK_THOMAS
        ctx->radau_solver = value ? 1 : 0;
#else
        if (value) return fail(ctx, -1, "marl_set_option: radau_solver = 1 (sequential block Thomas, the cross-check of the cyclic-reduction solver) is compiled into lab builds only (-DMARL_LAB_BLOCK_THOMAS)");
        ctx->radau_solver = 0;
#endif
    }
    else if (n == "radau_cr") ctx->radau_cr = (value < 0) ? -1 : std::min<int64_t>(value, radau::CR_MAX_LEVELS);
    else if (n == "radau_cr_min_n") ctx->radau_cr_min_n = std::max<int64_t>(value, 4);
    else if (n == "radau_cr_small") ctx->radau_cr_small = value < 0 ? 0 : std::min<int64_t>(value, radau::CR_WG_MAX_LEVELS);
    else if (n == "radau_cr_small_min_n") ctx->radau_cr_small_min_n = std::max<int64_t>(value, 32);
    else if (n == "radau_cr_tail") ctx->radau_cr_tail = value ? 1 : 0;
    else if (n == "bdf_solve_wg") ctx->bdf_solve_wg = value ? 1 : 0;
    else if (n == "radau_fused_solve") ctx->radau_fused_solve = value < 0 ? 0 : (value > 3 ? 3 : value);
    else if (n == "radau_sweep_wg") ctx->radau_sweep_wg = value < 0 ? 0 : (value > 3 ? 3 : value);
    else if (n == "implicit_zero_copy") ctx->implicit_zero_copy = value ? 1 : 0;
    else if (n == "rk4_stream") ctx->rk4_stream = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "rk45_stream") ctx->rk45_stream = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "dd_stream") ctx->dd_stream = value < 0 ? 0 : (value > 2 ? 2 : value);
    else if (n == "rk45_stream_attempts") ctx->rk45_stream_attempts = value < 1 ? 1 : std::min<int64_t>(value, 1 << 20);
    else if (n == "rk4_small") ctx->rk4_small = value ? 1 : 0;
    else if (n == "rk4_stream_third") ctx->rk4_stream_third = value ? 1 : 0;
    else if (n == "rk4_stream_test_raise") ctx->sq_test_raise = value != 0;
    else if (n == "rk4_stream_max_items") ctx->sq_max_items = value > 0 ? std::min<int64_t>(value, 0x7fffffff) : 0x7fffffff;
    else if (n == "no_reuse") {
        // every evaluation of the fused kernels takes its full path (what a rough state does wave by wave): re-upload the constants
        for (auto& c : ctx->hconsts) c.hot.no_reuse = value ? 1 : 0;
        HIP_OK(ctx, hipSetDevice(ctx->device));
        HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
        HIP_OK(ctx, hipMemcpy(ctx->dconsts, ctx->hconsts.data(), sizeof(DevConsts) * ctx->hconsts.size(), hipMemcpyHostToDevice));
    }
    else return fail(ctx, -1, "marl_set_option: unknown option '%s'", name);
    return 0;
}

int marl_get_constants(const marl_ctx* ctx, int64_t inst, double out[19])
{
    if (!ctx || inst < 0 || inst >= ctx->batch || !out) return -1;
    const DevConsts& d = ctx->hconsts[inst];
    const HotConsts& c = d.hot;
    const double* x = &ctx->extra[4 * inst];
    const double v[19] = {x[0], c.nu1, c.nu2, c.KRat, c.dCa, c.dCO3, c.delta, c.Da, c.lambda_, x[1], x[2], c.rhorat,
                          c.presum, x[3], c.dPhi, PECLET_MIN, PECLET_MAX, (double)d.mask_lo, (double)d.mask_hi};
    memcpy(out, v, sizeof v);
    return 0;
}

int64_t marl_state_doubles(const marl_ctx* ctx, int layout) { return ctx ? state_doubles(ctx->slab.n_buf, layout) : -1; }

}  // extern "C"

#ifndef MARL_LAB  // tools/rk4_lab.hip includes this file for context creation only
// ----------------------------------------------------------------------------------------------
// launch helpers
// ----------------------------------------------------------------------------------------------
static inline int64_t inst_stride(const marl_ctx* ctx, int layout) { return state_doubles(ctx->slab.n_buf, layout); }

// nstates > 0: that many states of the ONE model of the context, one after another (implicit path); 0: one state per instance
static int launch_rhs(marl_ctx* ctx, const double* y, double* dydt, int layout, int64_t nstates = 0)
{
    const dim3 grid((unsigned)((ctx->slab.n_buf + 255) / 256), (unsigned)(nstates > 0 ? nstates : ctx->batch));
    const int64_t stride = inst_stride(ctx, layout);
    const int cs = nstates > 0 ? 0 : 1;
    if (ctx->var_dphi) {
        if (layout == LAYOUT_TILED)
            hipLaunchKernelGGL((rhs_kernel<LAYOUT_TILED, true>), grid, dim3(256), 0, ctx->stream, y, dydt, ctx->dconsts, ctx->slab, stride, cs);
        else
            hipLaunchKernelGGL((rhs_kernel<LAYOUT_FIELD_MAJOR, true>), grid, dim3(256), 0, ctx->stream, y, dydt, ctx->dconsts, ctx->slab, stride, cs);
    } else if (layout == LAYOUT_TILED)
        hipLaunchKernelGGL(rhs_kernel<LAYOUT_TILED>, grid, dim3(256), 0, ctx->stream, y, dydt, ctx->dconsts, ctx->slab, stride, cs);
    else
        hipLaunchKernelGGL(rhs_kernel<LAYOUT_FIELD_MAJOR>, grid, dim3(256), 0, ctx->stream, y, dydt, ctx->dconsts, ctx->slab, stride, cs);
    LAUNCH_OK(ctx);
    return 0;
}

static int launch_convert(marl_ctx* ctx, const double* src, double* dst, int sl, int dl)
{
    const int64_t n = ctx->slab.n_buf;
    if (sl == dl) {
        HIP_OK(ctx, hipMemcpyAsync(dst, src, sizeof(double) * state_doubles(n, sl) * ctx->batch, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)ctx->batch);
    if (sl == LAYOUT_FIELD_MAJOR)
        hipLaunchKernelGGL((convert_kernel<LAYOUT_FIELD_MAJOR, LAYOUT_TILED>), grid, dim3(256), 0, ctx->stream, src, dst, n,
                           ctx->slab.ld, (int64_t)0, state_doubles(n, sl), state_doubles(n, dl));
    else
        hipLaunchKernelGGL((convert_kernel<LAYOUT_TILED, LAYOUT_FIELD_MAJOR>), grid, dim3(256), 0, ctx->stream, src, dst, n,
                           (int64_t)0, ctx->slab.ld, state_doubles(n, sl), state_doubles(n, dl));
    LAUNCH_OK(ctx);
    return 0;
}

// monitors of `y` -> ctx->rec[batch][NQ] (device)
static int launch_monitors(marl_ctx* ctx, const double* y, int layout, double* rec_out = nullptr)
{
    const int64_t n = ctx->slab.out_hi - ctx->slab.out_lo;
    int64_t nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (int rc = ensure_part(ctx, (size_t)(nb * ctx->batch))) return rc;
    const dim3 grid((unsigned)nb, (unsigned)ctx->batch);
    double* rec = rec_out ? rec_out : ctx->rec;
    double* first = nb == 1 ? rec : ctx->part;   // one workgroup per instance (N <= 256): its record IS the result - no second level
    if (layout == LAYOUT_TILED)
        hipLaunchKernelGGL(monitors_kernel<LAYOUT_TILED>, grid, dim3(256), 0, ctx->stream, y, ctx->dconsts, ctx->slab, inst_stride(ctx, layout), first);
    else
        hipLaunchKernelGGL(monitors_kernel<LAYOUT_FIELD_MAJOR>, grid, dim3(256), 0, ctx->stream, y, ctx->dconsts, ctx->slab, inst_stride(ctx, layout), first);
    LAUNCH_OK(ctx);
    if (nb == 1) return 0;
    hipLaunchKernelGGL(reduce_records_kernel, dim3((unsigned)ctx->batch), dim3(256), 0, ctx->stream, ctx->part, nb, rec);
    LAUNCH_OK(ctx);
    return 0;
}

static void record_to_events(const double* r, double* g)
{
    g[0] = r[1]; g[1] = r[2]; g[2] = r[3]; g[3] = r[5] - 1.0; g[4] = r[6] - 1.0; g[5] = r[4]; g[6] = r[7];
}

// ---- fused RK4 variants ------------------------------------------------------------------------
struct Rk4Variant { int blk, cpt, nsteps; };
// One shape ships: 256-thread blocks, one cell per thread (4 waves per SIMD), at 1 / 2 / 4 / 8 / 16 steps per launch - the default
// depth depends on the grid size, the shallower ones also serve as the remainder chain.  (Other block shapes and cells per thread
// were measured in round 1 and live in tools/rk4_lab.hip only.)
static const Rk4Variant kRk4Variants[] = {{256, 1, 1}, {256, 1, 2}, {256, 1, 4}, {256, 1, 8}, {256, 1, 16}};
constexpr int kNumRk4Variants = sizeof(kRk4Variants) / sizeof(kRk4Variants[0]);

template <int BLK, int CPT, int NSTEPS, bool VD = false>
static void launch_rk4_t(marl_ctx* ctx, const double* yin, double* yout, int layout, double dt)
{
    constexpr int V = BLK * CPT - 8 * NSTEPS;
    const int64_t n = ctx->slab.out_hi - ctx->slab.out_lo;
    const dim3 grid((unsigned)((n + V - 1) / V));
    if (layout == LAYOUT_TILED)
        hipLaunchKernelGGL((rk4_fused_kernel<BLK, CPT, LAYOUT_TILED, NSTEPS, VD>), grid, dim3(BLK), 0, ctx->stream, yin, yout, ctx->dconsts, ctx->slab, dt);
    else
        hipLaunchKernelGGL((rk4_fused_kernel<BLK, CPT, LAYOUT_FIELD_MAJOR, NSTEPS, VD>), grid, dim3(BLK), 0, ctx->stream, yin, yout, ctx->dconsts, ctx->slab, dt);
}

// The kernel sets instantiated with the time-varying porosity diffusion coefficient (dPhi_variable): one
// one-cell-per-thread shape per integrator - the option is a model variant, not a tuning surface.
constexpr int kVdRk4Variant = 2;    // {256, 1, 4}
constexpr int kVdRk45Variant = 0;   // {256, 1}

extern "C" int marl_small_rk4_16(int tiled, unsigned grid, hipStream_t stream, const double* yin, double* yout, const void* consts, const int64_t slab5[5],
                                 double dt);   // marl_rk4_small.hip

// nsteps: steps fused in this launch - the variant's own depth, or a smaller instantiated one for the remainder
static int launch_rk4(marl_ctx* ctx, int v, int nsteps, const double* yin, double* yout, int layout, double dt)
{
    (void)v;
    switch (nsteps) {
        case 1: if (ctx->var_dphi) launch_rk4_t<256, 1, 1, true>(ctx, yin, yout, layout, dt); else launch_rk4_t<256, 1, 1>(ctx, yin, yout, layout, dt); break;
        case 2: launch_rk4_t<256, 1, 2>(ctx, yin, yout, layout, dt); break;
        case 4: if (ctx->var_dphi) launch_rk4_t<256, 1, 4, true>(ctx, yin, yout, layout, dt); else launch_rk4_t<256, 1, 4>(ctx, yin, yout, layout, dt); break;
        case 8: launch_rk4_t<256, 1, 8>(ctx, yin, yout, layout, dt); break;
        case 16: {
            // grids whose workgroups fit the chip at TWO per CU (N <= 65 536 on 256 CUs: BASELINE configs[1]) take the build of this kernel
            // that trades occupancy for registers and ILP (marl_rk4_small.hip: 3.43 against 3.65 us per step at N = 65 536); with a third
            // workgroup per CU that build needs a second round (N = 98 304: 6.39 against 4.47 us) - those keep the four-waves-per-SIMD one
            const int64_t n = ctx->slab.out_hi - ctx->slab.out_lo, V = 256 - 8 * 16, tiles = (n + V - 1) / V;
            if (!ctx->cus) {
                hipDeviceProp_t prop;
                HIP_OK(ctx, hipGetDeviceProperties(&prop, ctx->device));
                ctx->cus = prop.multiProcessorCount;
            }
            if (ctx->rk4_small && tiles <= 2 * (int64_t)ctx->cus) {
                const int64_t slab5[5] = {ctx->slab.n_buf, ctx->slab.goff, ctx->slab.ld, ctx->slab.out_lo, ctx->slab.out_hi};
                const int e = marl_small_rk4_16(layout == LAYOUT_TILED, (unsigned)tiles, ctx->stream, yin, yout, ctx->dconsts, slab5, dt);
                if (e != 0) return fail(ctx, -100 - e, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
                return 0;
            }
            launch_rk4_t<256, 1, 16>(ctx, yin, yout, layout, dt);
            break;
        }
        default: return fail(ctx, -1, "rk4: %d steps per launch not instantiated", nsteps);
    }
    LAUNCH_OK(ctx);
    return 0;
}

static int default_rk4_variant(const marl_ctx* ctx)
{
    if (ctx->var_dphi) return kVdRk4Variant;
    if (ctx->rk4_variant >= 0 && ctx->rk4_variant < kNumRk4Variants) return (int)ctx->rk4_variant;
    const int64_t n = ctx->slab.out_hi - ctx->slab.out_lo;
    // 256-thread blocks; the smaller the grid, the more the launch boundary matters against the recomputed halo:
    // 16 / 8 / 4 steps per launch (tools/variant_sweep.sh)
    return n <= 98304 ? 4 : (n <= 262144 ? 3 : 2);
}

#ifndef MARL_LAB_STREAM_BLK   // (kernel-lab switch: threads per workgroup of the streamed loop; 512 = half the halo share, twice the waves per barrier)
#define MARL_LAB_STREAM_BLK 256
#endif
// One dataflow launch for `levels` x `per` steps (rk4_stream_kernel): bufA -> ... -> (levels odd ? bufB : bufA)
template <int NSTEPS, bool VD = false>
static void launch_stream_t(marl_ctx* ctx, double* a, double* b, int layout, double dt, unsigned levels, unsigned tiles, unsigned blocks, double* c = nullptr)
{
    if (layout == LAYOUT_TILED)
        hipLaunchKernelGGL((rk4_stream_kernel<MARL_LAB_STREAM_BLK, LAYOUT_TILED, NSTEPS, VD>), dim3(blocks), dim3(MARL_LAB_STREAM_BLK), 0, ctx->stream, a, b, ctx->dconsts, ctx->slab, dt,
                           levels, tiles, ctx->sq, ctx->sq + 2, ctx->sq_sticky, ctx->sq_item_base, ctx->sq_level_base, c);
    else
        hipLaunchKernelGGL((rk4_stream_kernel<MARL_LAB_STREAM_BLK, LAYOUT_FIELD_MAJOR, NSTEPS, VD>), dim3(blocks), dim3(MARL_LAB_STREAM_BLK), 0, ctx->stream, a, b, ctx->dconsts, ctx->slab,
                           dt, levels, tiles, ctx->sq, ctx->sq + 2, ctx->sq_sticky, ctx->sq_item_base, ctx->sq_level_base, c);
}

// *result: where the state is afterwards (a or b)
static int rk4_stream(marl_ctx* ctx, double* a, double* b, int layout, double dt, int per, int64_t levels, double** result)
{
    const int64_t n = ctx->slab.out_hi - ctx->slab.out_lo, V = MARL_LAB_STREAM_BLK - 8 * per, tiles = (n + V - 1) / V;
    if (!ctx->cus) {
        hipDeviceProp_t prop;
        HIP_OK(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->cus = prop.multiProcessorCount;
    }
    if (!ctx->sq_host) HIP_OK(ctx, hipHostMalloc((void**)&ctx->sq_host, sizeof(unsigned), hipHostMallocDefault));
    if (!ctx->sq_sticky) {
        HIP_OK(ctx, hipMalloc((void**)&ctx->sq_sticky, sizeof(unsigned)));
        HIP_OK(ctx, hipMemsetAsync(ctx->sq_sticky, 0, sizeof(unsigned), ctx->stream));
    }
    if (ctx->sq_cap < (size_t)tiles + 2 || ctx->sq_level_base > (1u << 30)) {   // (re)start the counters
        if (ctx->sq_cap < (size_t)tiles + 2) {
            if (ctx->sq) HIP_OK(ctx, hipFree(ctx->sq));
            ctx->sq = nullptr;
            ctx->sq_cap = 0;
            HIP_OK(ctx, hipMalloc((void**)&ctx->sq, ((size_t)tiles + 2) * sizeof(unsigned)));
            ctx->sq_cap = (size_t)tiles + 2;
        }
        HIP_OK(ctx, hipMemsetAsync(ctx->sq, 0, ctx->sq_cap * sizeof(unsigned), ctx->stream));
        ctx->sq_item_base = ctx->sq_level_base = 0;
    }
    if (ctx->sq_test_raise) {   // as if a workgroup of an earlier launch had given up: waiting workgroups leave, marl_synchronize reports
        HIP_OK(ctx, hipMemsetAsync(ctx->sq_sticky, 1, sizeof(unsigned), ctx->stream));
        ctx->sq_test_raise = false;
    }
    while (levels > 0) {
        // (the item counter is 32 bits wide; an even number of levels per launch keeps the ping-pong orientation)
        const int64_t cap = (ctx->sq_max_items / tiles) & ~(int64_t)1;
        const int64_t lv = std::min<int64_t>(levels, std::max<int64_t>(cap, 2));
        const unsigned blocks = (unsigned)std::min<int64_t>(lv * tiles, (4 * 256 / MARL_LAB_STREAM_BLK) * (int64_t)ctx->cus);   // 4 workgroups of 256 per CU are resident
        // the LAST launch with an odd number (>= 3) of levels goes A -> B, B <-> C, ... -> A through a third buffer: the result lands where
        // the caller's state is and the whole-state copy afterwards goes away (12 us of a 20-step call at N = 2^20; option rk4_stream_third)
        double* c = nullptr;
        if (ctx->rk4_stream_third && levels == lv && (lv & 1) && lv >= 3) {
            const size_t need = (size_t)state_doubles(ctx->slab.n_buf, layout);
            if (ctx->stream_c_cap < need) {
                if (ctx->stream_c) HIP_OK(ctx, hipFree(ctx->stream_c));
                ctx->stream_c = nullptr; ctx->stream_c_cap = 0;
                HIP_OK(ctx, hipMalloc((void**)&ctx->stream_c, need * sizeof(double)));
                ctx->stream_c_cap = need;
            }
            c = ctx->stream_c;
        }
        switch (per) {
            case 1: if (ctx->var_dphi) launch_stream_t<1, true>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c);
                    else launch_stream_t<1>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c);
                    break;
            case 2: launch_stream_t<2>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c); break;
            case 4: if (ctx->var_dphi) launch_stream_t<4, true>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c);
                    else launch_stream_t<4>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c);
                    break;
            case 8: launch_stream_t<8>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c); break;
            case 16: launch_stream_t<16>(ctx, a, b, layout, dt, (unsigned)lv, (unsigned)tiles, blocks, c); break;
            default: return fail(ctx, -1, "rk4 stream: %d steps per level not instantiated", per);
        }
        LAUNCH_OK(ctx);
        ctx->sq_pending = true;
        ctx->sq_item_base += (unsigned)(lv * tiles) + blocks;   // its items + one failing grab per workgroup (wraps with the counter)
        ctx->sq_level_base += (unsigned)lv;
        levels -= lv;
        if ((lv & 1) && !c) std::swap(a, b);   // (now `a` holds the state)
    }
    *result = a;
    return 0;
}

constexpr int64_t kStreamMinCells = 196608;

// y (device, `layout`) advanced in place; `tmp` is a second buffer of the same size
static int rk4_run(marl_ctx* ctx, double* y, double* tmp, int layout, double dt, int64_t nsteps)
{
    const int v = default_rk4_variant(ctx);
    const int per = kRk4Variants[v].nsteps;
    double* a = y;
    double* b = tmp;
    int64_t left = nsteps;
    // One dataflow launch instead of one launch per level where it pays (tools/lab record in profiles/r02_lab_rk4_stream.log):
    // from ~200 000 cells (below, the tiles do not fill the resident workgroups and the state is served from the L2s anyway).
    const bool stream = ctx->rk4_stream == 2 ? left >= per
                                             : (ctx->rk4_stream == 1 && ctx->slab.out_hi - ctx->slab.out_lo >= kStreamMinCells && left >= 2 * per);
    if (stream && ctx->halo == 0) {
        const int64_t levels = left / per;
        double* now = nullptr;
        if (int rc = rk4_stream(ctx, a, b, layout, dt, per, levels, &now)) return rc;
        if (now != a) std::swap(a, b);
        left -= levels * per;
    }
    while (left >= per) {
        if (int rc = launch_rk4(ctx, v, per, a, b, layout, dt)) return rc;
        std::swap(a, b);
        left -= per;
    }
    // remainder: every power of two below the depth is instantiated (the dPhi_variable set: depths 4 and 1 only)
    const bool family = !ctx->var_dphi;
    for (int chunk = per / 2; chunk >= 1 && left > 0; chunk /= 2) {
        const int c = family ? chunk : 1;
        while (left >= c) {
            if (int rc = launch_rk4(ctx, v, c, a, b, layout, dt)) return rc;
            std::swap(a, b);
            left -= c;
            if (family) break;   // at most one launch per power of two
        }
    }
    if (a != y) HIP_OK(ctx, hipMemcpyAsync(y, a, sizeof(double) * state_doubles(ctx->slab.n_buf, layout), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// ---- sweep variants ----------------------------------------------------------------------------
struct SweepVariant { int blk, cpt; };
// one cell per thread; the smallest window that holds the grid is taken
static const SweepVariant kSweepVariants[] = {{256, 1}, {512, 1}, {1024, 1}};
constexpr int kNumSweepVariants = sizeof(kSweepVariants) / sizeof(kSweepVariants[0]);

static int default_sweep_variant(marl_ctx* ctx)
{
    if (ctx->sweep_variant >= 0 && ctx->sweep_variant < kNumSweepVariants && (int64_t)kSweepVariants[ctx->sweep_variant].blk >= ctx->N)
        return (int)ctx->sweep_variant;
    for (int i = 0; i < kNumSweepVariants; i++)
        if ((int64_t)kSweepVariants[i].blk * kSweepVariants[i].cpt >= ctx->N) return i;
    return -1;
}

#define SWEEP_DISPATCH(KERNEL, ...)                                                                           \
    switch (v) {                                                                                              \
        case 0: if (ctx->var_dphi) hipLaunchKernelGGL((KERNEL<256, 1, true>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); \
                else hipLaunchKernelGGL((KERNEL<256, 1>), grid, dim3(256), 0, ctx->stream, __VA_ARGS__); break;    \
        case 1: if (ctx->var_dphi) hipLaunchKernelGGL((KERNEL<512, 1, true>), grid, dim3(512), 0, ctx->stream, __VA_ARGS__); \
                else hipLaunchKernelGGL((KERNEL<512, 1>), grid, dim3(512), 0, ctx->stream, __VA_ARGS__); break;    \
        case 2: if (ctx->var_dphi) hipLaunchKernelGGL((KERNEL<1024, 1, true>), grid, dim3(1024), 0, ctx->stream, __VA_ARGS__); \
                else hipLaunchKernelGGL((KERNEL<1024, 1>), grid, dim3(1024), 0, ctx->stream, __VA_ARGS__); break;  \
        default: return fail(ctx, -1, "sweep variant %d not instantiated", v);                                \
    }

// ---- fused RK45 variants -----------------------------------------------------------------------
struct Rk45Variant { int blk, cpt; };
static const Rk45Variant kRk45Variants[] = {{256, 1}};   // (other shapes were measured in round 1; none faster)
constexpr int kNumRk45Variants = sizeof(kRk45Variants) / sizeof(kRk45Variants[0]);

static int default_rk45_variant(const marl_ctx* ctx)
{
    if (ctx->var_dphi) return kVdRk45Variant;
    if (ctx->rk45_variant >= 0 && ctx->rk45_variant < kNumRk45Variants) return (int)ctx->rk45_variant;
    return 0;
}

static int64_t rk45_blocks(const marl_ctx* ctx, int v)
{
    const int V = kRk45Variants[v].blk * kRk45Variants[v].cpt - 12;
    const int64_t n = ctx->slab.out_hi - ctx->slab.out_lo;
    return (n + V - 1) / V;
}

constexpr int64_t kReduceGroups = 64;
template <int BLK, int CPT, bool VD = false>
static void launch_attempt_t(marl_ctx* ctx, int64_t nb, int layout)
{
    if (layout == LAYOUT_TILED)
        hipLaunchKernelGGL((rk45_attempt_kernel<BLK, CPT, LAYOUT_TILED, VD>), dim3((unsigned)nb), dim3(BLK), 0, ctx->stream, ctx->buf[0], ctx->buf[1],
                           ctx->buf[2], ctx->buf[3], ctx->dconsts, ctx->slab, ctx->dctrl, ctx->part);
    else
        hipLaunchKernelGGL((rk45_attempt_kernel<BLK, CPT, LAYOUT_FIELD_MAJOR, VD>), dim3((unsigned)nb), dim3(BLK), 0, ctx->stream, ctx->buf[0],
                           ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dconsts, ctx->slab, ctx->dctrl, ctx->part);
}

// Records of a large grid are reduced in two levels: `*nrec` per-workgroup records at ctx->part -> at most kReduceGroups
// records behind them (same buffer, offset `*nrec`); returns where the second level reads and updates *nrec.
static const double* reduce_first_level(marl_ctx* ctx, int64_t* nrec)
{
    const int64_t nb = *nrec;
    if (nb <= 4 * kReduceGroups) return ctx->part;
    const int64_t chunk = (nb + kReduceGroups - 1) / kReduceGroups, groups = (nb + chunk - 1) / chunk;
    double* out = ctx->part + nb * NQ;
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)groups), dim3(256), 0, ctx->stream, ctx->part, nb, chunk, out);
    *nrec = groups;
    return out;
}

static int launch_attempt(marl_ctx* ctx, int v, int layout)
{
    const int64_t nb = rk45_blocks(ctx, v);
    switch (v) {
        case 0: if (ctx->var_dphi) launch_attempt_t<256, 1, true>(ctx, nb, layout); else launch_attempt_t<256, 1>(ctx, nb, layout); break;
        default: return fail(ctx, -1, "rk45 variant %d not instantiated", v);
    }
    LAUNCH_OK(ctx);
    int64_t nrec = nb;
    const double* recs = reduce_first_level(ctx, &nrec);
    LAUNCH_OK(ctx);
    hipLaunchKernelGGL(rk45_control_kernel, dim3(1), dim3(CONTROL_THREADS), 0, ctx->stream, recs, nrec, ctx->dctrl);
    LAUNCH_OK(ctx);
    return 0;
}

// ---- the adaptive loop of one grid as ONE launch per batch of attempts (rk45_stream_kernel) ----------------------------------
// barrier words zeroed, the grid-wide extrema armed (+inf); synchronises
static int rk45_stream_reset(marl_ctx* ctx)
{
    std::vector<double> inf((size_t)GREC_DOUBLES - MON_OFFSET, (double)INFINITY);
    HIP_OK(ctx, hipMemsetAsync(ctx->rs, 0, sizeof(Rk45Stream), ctx->stream));
    HIP_OK(ctx, hipMemsetAsync(ctx->rs_grec, 0, sizeof(double) * MON_OFFSET, ctx->stream));
    HIP_OK(ctx, hipMemcpyAsync(ctx->rs_grec + MON_OFFSET, inf.data(), inf.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->rs_arrive_base = ctx->rs_epoch_base = ctx->rs_grab_base = 0;
    return 0;
}

static int rk45_stream_setup(marl_ctx* ctx)
{
    if (!ctx->cus) {
        hipDeviceProp_t prop;
        HIP_OK(ctx, hipGetDeviceProperties(&prop, ctx->device));
        ctx->cus = prop.multiProcessorCount;
    }
    if (!ctx->rs) {
        HIP_OK(ctx, hipMalloc((void**)&ctx->rs, sizeof(Rk45Stream)));
        HIP_OK(ctx, hipMalloc((void**)&ctx->rs_grec, sizeof(double) * GREC_DOUBLES));
        HIP_OK(ctx, hipHostMalloc((void**)&ctx->rs_host, 2 * sizeof(unsigned), hipHostMallocDefault));
        if (int rc = rk45_stream_reset(ctx)) return rc;
    }
    if (ctx->rs_arrive_base > (1u << 30) || ctx->rs_epoch_base > (1u << 30) || ctx->rs_grab_base > (1u << 30))   // (the stream is idle between batches: the caller has synchronised)
        if (int rc = rk45_stream_reset(ctx)) return rc;
    return 0;
}

// dd: one attempt of a slab; the last workgroup packs the rank's message into ctx->dd_send and re-arms the counters (bases 0)
template <bool VD>
static int launch_rk45_stream_t(marl_ctx* ctx, int layout, int64_t tiles, int64_t max_attempts, bool dd = false)
{
    if (dd && layout != LAYOUT_FIELD_MAJOR) return fail(ctx, -1, "rk45 stream: slabs are field-major");
    auto kern = dd ? rk45_stream_kernel<256, LAYOUT_FIELD_MAJOR, VD, true>
                   : (layout == LAYOUT_TILED ? rk45_stream_kernel<256, LAYOUT_TILED, VD, false> : rk45_stream_kernel<256, LAYOUT_FIELD_MAJOR, VD, false>);
    int& occ = ctx->rs_occ[dd ? 2 : (layout == LAYOUT_TILED ? 1 : 0)][VD ? 1 : 0];   // (per instantiation: the resident-grid bound below rests on it)
    if (!occ) {
        HIP_OK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0));
        if (occ < 1) return fail(ctx, -3, "rk45 stream: the kernel does not fit a compute unit");
        if (occ > 16) occ = 16;
    }
    // every workgroup of the grid must be resident at once (they meet at a barrier in memory): at most what the device holds
    ctx->rs_G = (unsigned)std::min<int64_t>(std::min<int64_t>(tiles, (int64_t)occ * ctx->cus), MON_OFFSET);
    hipLaunchKernelGGL(kern, dim3(ctx->rs_G), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dconsts, ctx->slab,
                       ctx->dctrl, ctx->rs_grec, ctx->rs, (unsigned)tiles, dd ? 0u : ctx->rs_arrive_base, dd ? 0u : ctx->rs_epoch_base, dd ? 0u : ctx->rs_grab_base,
                       dd ? 1u : (unsigned)max_attempts, dd ? ctx->dd_send : (double*)nullptr, dd ? ctx->halo : 0);
    const unsigned R = (unsigned)(tiles % ctx->rs_G);
    ctx->rs_grabs_per_attempt = R ? R + ctx->rs_G : 0;   // (what the kernel's workgroups take from the remainder counter per attempt)
    LAUNCH_OK(ctx);
    return 0;
}

// Where the persistent loop pays (profiles/r04_lab_rk45_stream.log; MI355X, 1024 resident workgroups): what it removes is the fixed
// cost per attempt - two small launches and every workgroup's prologue, ~8 us against a ~5 us barrier - and what it loses is the
// dispatcher's balancing: its workgroups own STATIC tiles, the SIMDs arbitrate oldest-first, so a CU's four workgroups finish
// one after the other and the last runs alone.  N = 65 536: +35 %, 2^19: +7 %, 2^20: -1 %, 2^22: -10 %.  Default: grids of up to
// three rounds of resident workgroups (N <= ~750 000 cells on 256 CUs); larger grids keep one launch per attempt.
constexpr int64_t kRk45StreamRounds = 3;
static bool rk45_use_stream(marl_ctx* ctx, int v)
{
    if (ctx->rk45_stream == 0) return false;
    if (ctx->rk45_stream >= 2) return true;
    if (!ctx->cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return false;
        ctx->cus = prop.multiProcessorCount;
    }
    return rk45_blocks(ctx, v) <= kRk45StreamRounds * 4 * (int64_t)ctx->cus;
}

static int launch_rk45_stream(marl_ctx* ctx, int v, int layout, int64_t max_attempts, bool dd = false)
{
    if (int rc = rk45_stream_setup(ctx)) return rc;
    const int64_t tiles = rk45_blocks(ctx, v);
    if (tiles >= (1ll << 31)) return fail(ctx, -1, "rk45 stream: grid too large");
    return ctx->var_dphi ? launch_rk45_stream_t<true>(ctx, layout, tiles, max_attempts, dd) : launch_rk45_stream_t<false>(ctx, layout, tiles, max_attempts, dd);
}

// after the batch's status read-back (stream idle): where the ticket / barrier counters stand now, and the sticky flag
static int rk45_stream_account(marl_ctx* ctx, int64_t attempts_in_batch)
{
    HIP_OK(ctx, hipMemcpyAsync(ctx->rs_host, ctx->rs, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    const bool consistent = ctx->rs_host[0] - ctx->rs_arrive_base == ctx->rs_G * (unsigned)attempts_in_batch;
    if (ctx->rs_host[1] || !consistent) {
        const unsigned took = ctx->rs_host[0] - ctx->rs_arrive_base;
        if (int rc = rk45_stream_reset(ctx)) return rc;
        if (ctx->rs_host[1]) return fail(ctx, -2, "rk45: the persistent time loop gave up waiting at its barrier (rk45_stream_kernel); the state is invalid");
        return fail(ctx, -2, "rk45: the persistent time loop took %u tickets for %lld attempts of %u workgroups", took, (long long)attempts_in_batch, ctx->rs_G);
    }
    ctx->rs_arrive_base += ctx->rs_G * (unsigned)attempts_in_batch;
    ctx->rs_epoch_base += (unsigned)attempts_in_batch;
    ctx->rs_grab_base += ctx->rs_grabs_per_attempt * (unsigned)attempts_in_batch;
    return 0;
}

template <int BLK, int CPT, bool VD = false>
static void launch_dense_t(marl_ctx* ctx, int64_t nb, int layout, const double* yold, const double* fold, double h,
                           const DenseWeights& dw, double* yout)
{
    if (layout == LAYOUT_TILED)
        hipLaunchKernelGGL((rk45_dense_kernel<BLK, CPT, LAYOUT_TILED, VD>), dim3((unsigned)nb), dim3(BLK), 0, ctx->stream, yold, fold, ctx->dconsts,
                           ctx->slab, h, dw, yout, ctx->part);
    else
        hipLaunchKernelGGL((rk45_dense_kernel<BLK, CPT, LAYOUT_FIELD_MAJOR, VD>), dim3((unsigned)nb), dim3(BLK), 0, ctx->stream, yold, fold,
                           ctx->dconsts, ctx->slab, h, dw, yout, ctx->part);
}

// Dense output P (scipy/integrate/_ivp/rk.py:393-407): w_j(x) = sum_m P[j][m] x^(m+1)
static const double kDpP[7][4] = {
    {1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
    {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
    {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
    {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
    {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};

// Evaluate the dense output of the LAST accepted step at time t: state into yout (may be NULL) and the
// seven monitors into g (may be NULL; synchronises when given).
static int dense_eval(marl_ctx* ctx, int v, int layout, bool small, const Rk45Ctrl& c, double t, double* yout, double* g)
{

    const double x = (t - c.t_old) / c.h_prev;
    DenseWeights dw;
    double pw[4] = {x, x * x, x * x * x, x * x * x * x};
    for (int j = 0; j < 7; j++) {
        double a = 0;
        for (int m = 0; m < 4; m++) a += kDpP[j][m] * pw[m];
        dw.w[j] = a;
    }
    const double* yold = small ? ctx->buf[1] : ctx->buf[c.cur ^ 1];
    const double* fold = small ? ctx->buf[3] : ctx->buf[2 + (c.cur ^ 1)];
    const int64_t nb = rk45_blocks(ctx, v);
    switch (v) {
        case 0: if (ctx->var_dphi) launch_dense_t<256, 1, true>(ctx, nb, layout, yold, fold, c.h_prev, dw, yout); else launch_dense_t<256, 1>(ctx, nb, layout, yold, fold, c.h_prev, dw, yout); break;
        default: return fail(ctx, -1, "rk45 variant %d not instantiated", v);
    }
    LAUNCH_OK(ctx);
    if (g) {
        hipLaunchKernelGGL(reduce_records_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->part, nb, ctx->rec);
        LAUNCH_OK(ctx);
        HIP_OK(ctx, hipMemcpyAsync(ctx->hrec, ctx->rec, sizeof(double) * NQ, hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
        record_to_events(ctx->hrec, g);
    }
    return 0;
}

// Brent's method for monitor `e` on the last accepted step (solve_event_equation, ivp.py:51-76)
static int brent_event(marl_ctx* ctx, int v, int layout, bool small, const Rk45Ctrl& c, int e, double ga, double gb, double* root)
{
    const double xtol = 4 * 2.220446049250313e-16, rtol = xtol;
    double a = c.t_old, b = c.t, fa = ga, fb = gb, g[7];
    if (fa == 0) { *root = a; return 0; }
    if (fb == 0) { *root = b; return 0; }
    double xpre = a, xcur = b, fpre = fa, fcur = fb, xblk = 0, fblk = 0, spre = 0, scur = 0;
    for (int it = 0; it < 100; it++) {
        if (fpre != 0 && fcur != 0 && ((fpre < 0) != (fcur < 0))) { xblk = xpre; fblk = fpre; spre = scur = xcur - xpre; }
        if (std::fabs(fblk) < std::fabs(fcur)) { xpre = xcur; xcur = xblk; xblk = xpre; fpre = fcur; fcur = fblk; fblk = fpre; }
        const double delta = (xtol + rtol * std::fabs(xcur)) / 2, sbis = (xblk - xcur) / 2;
        if (fcur == 0 || std::fabs(sbis) < delta) break;
        if (std::fabs(spre) > delta && std::fabs(fcur) < std::fabs(fpre)) {
            double stry;
            if (xpre == xblk) stry = -fcur * (xcur - xpre) / (fcur - fpre);
            else {
                const double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * std::fabs(stry) < std::fmin(std::fabs(spre), 3 * std::fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else { spre = sbis; scur = sbis; }
        xpre = xcur; fpre = fcur;
        if (std::fabs(scur) > delta) xcur += scur; else xcur += (sbis > 0 ? delta : -delta);
        if (int rc = dense_eval(ctx, v, layout, small, c, xcur, nullptr, g)) return rc;
        fcur = g[e];
    }
    *root = xcur;
    return 0;
}

static void ctrl_to_stats(const Rk45Ctrl& c, marl_stats* st)
{
    memset(st, 0, sizeof *st);
    st->nfev = c.nfev;
    st->n_accepted = c.n_acc;
    st->n_rejected = c.n_rej;
    st->status = c.status;
    st->t = c.t;
    st->h_next = c.h_abs;
    for (int e = 0; e < 7; e++) { st->event_value[e] = c.g[e]; st->n_events[e] = c.n_events[e]; }
}

// Attempts to enqueue before the next status read: `poll`, but never more than an attempt budget has left - once the
// controller has stopped, every further attempt / reduce / control launch of the batch is a dispatch that does nothing
// (cheap, but it is time, and it dilutes per-launch profile averages).  Pauses for t_eval samples and events still cut a
// batch short; those are rare.
static int64_t attempts_per_batch(int64_t poll, int64_t max_attempts, int64_t executed)
{
    if (max_attempts <= 0) return poll;
    const int64_t left = max_attempts - executed;
    return left < 1 ? 1 : (left < poll ? left : poll);
}

// The adaptive loop on device buffers buf[0..3] (`layout`), state already in buf[0].
// small = true: the grid fits one workgroup -> the persistent sweep kernel runs all attempts on-chip
// (state in buf[0], FIELD-MAJOR; (y_old, f_old) of a paused step in buf[1], buf[3]).
static int rk45_run(marl_ctx* ctx, int layout, bool small, double t0, double t1, double first_step, double rtol, double atol,
                    const double* t_eval, int64_t n_eval, double* y_eval_dev, double* t_events, int64_t max_events,
                    int64_t max_attempts, marl_stats* stats)
{
    if (!(first_step > 0) || !(t1 >= t0)) return fail(ctx, -1, "rk45: need first_step > 0 and t1 >= t0 (forward integration)");
    if (t1 > t0 && first_step > t1 - t0) return fail(ctx, -1, "rk45: `first_step` exceeds bounds");  // common.py:10-16
    if (!(rtol > 0) || !(atol >= 0)) return fail(ctx, -1, "rk45: tolerances must be positive");
    rtol = clamp_rtol(rtol);
    for (int64_t i = 0; i < n_eval; i++)
        if (t_eval[i] < t0 || t_eval[i] > t1 || (i > 0 && t_eval[i] <= t_eval[i - 1]))
            return fail(ctx, -1, "rk45: `t_eval` must be sorted and within t_span");  // ivp.py:603-609
    const int v = small ? 0 : default_rk45_variant(ctx);
    const int sv = small ? default_sweep_variant(ctx) : -1;
    const int64_t nb = rk45_blocks(ctx, v);
    if (int rc = ensure_part(ctx, (size_t)std::max<int64_t>(nb + kReduceGroups, 1024))) return rc;
    const int64_t sd = state_doubles(ctx->slab.n_buf, layout);
    // f(t0, y0) and the monitors at t0
    if (!small)
        if (int rc = launch_rhs(ctx, ctx->buf[0], ctx->buf[2], layout)) return rc;
    if (int rc = launch_monitors(ctx, ctx->buf[0], layout)) return rc;
    hipLaunchKernelGGL(rk45_init_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dctrl, ctx->rec, t0, t1, first_step, rtol, atol,
                       (int64_t)NF * ctx->N, max_attempts, 0);
    LAUNCH_OK(ctx);
    int64_t eval_i = 0;
    // t_eval == t0 is emitted on the first step by scipy (dense output at x = 0 == y_old)
    const bool events_on = t_events != nullptr && max_events > 0;
    auto next_pause = [&]() { return eval_i < n_eval ? t_eval[eval_i] : (double)INFINITY; };
    hipLaunchKernelGGL(rk45_resume_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dctrl, next_pause(), max_attempts);
    LAUNCH_OK(ctx);
    if (events_on) {
        // pause_on_event is a plain field: set it through a tiny host round trip once
        HIP_OK(ctx, hipMemcpyAsync(ctx->hctrl, ctx->dctrl, sizeof(Rk45Ctrl), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->hctrl->pause_on_event = 1;
        HIP_OK(ctx, hipMemcpyAsync(ctx->dctrl, ctx->hctrl, sizeof(Rk45Ctrl), hipMemcpyHostToDevice, ctx->stream));
    }
    int64_t seen_events[7] = {0, 0, 0, 0, 0, 0, 0};
    Rk45Ctrl& hc = *ctx->hctrl;
    int64_t executed = 0;   // attempts the device has finished, as of the last status read
    const bool streamed = !small && rk45_use_stream(ctx, v);   // one launch per batch of attempts (rk45_stream_kernel)
    while (true) {
        if (small) {
            const int v = sv;  // SWEEP_DISPATCH switches on `v`
            const dim3 grid(1);
            if (events_on) {   // pauses on monitor sign changes (root finding): the loop shape that decides with this step's events
                SWEEP_DISPATCH(rk45_sweep_events_kernel, ctx->buf[0], ctx->dconsts, ctx->dctrl, ctx->N, ctx->buf[1], ctx->buf[3])
            } else {
                SWEEP_DISPATCH(rk45_sweep_kernel, ctx->buf[0], ctx->dconsts, ctx->dctrl, ctx->N, ctx->buf[1], ctx->buf[3])
            }
            LAUNCH_OK(ctx);
        } else if (streamed) {
            if (int rc = launch_rk45_stream(ctx, v, layout, ctx->rk45_stream_attempts)) return rc;
        } else {
            const int64_t batch = attempts_per_batch(ctx->poll, max_attempts, executed);
            for (int64_t i = 0; i < batch; i++)
                if (int rc = launch_attempt(ctx, v, layout)) return rc;
        }
        HIP_OK(ctx, hipMemcpyAsync(ctx->hctrl, ctx->dctrl, sizeof(Rk45Ctrl), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
        const int64_t executed_before = executed;
        executed = hc.attempts - (hc.status == ST_RUNNING ? 1 : 0);   // (a running controller has already prepared the next one)
        if (streamed)
            if (int rc = rk45_stream_account(ctx, executed - executed_before)) return rc;
        if (hc.status == ST_RUNNING) continue;
        const bool stepped = hc.n_acc > 0;
        // event roots inside the last accepted step (ivp.py:673-694)
        if (events_on && stepped) {
            for (int e = 0; e < 7; e++) {
                if (hc.n_events[e] > seen_events[e]) {
                    // only the newest sign change can be refined (earlier ones were refined at their own pause)
                    double ga[7];
                    if (int rc = dense_eval(ctx, v, layout, small, hc, hc.t_old, nullptr, ga)) return rc;
                    double root = hc.ev_last[e];
                    if (int rc = brent_event(ctx, v, layout, small, hc, e, ga[e], hc.g[e], &root)) return rc;
                    for (int64_t k = seen_events[e]; k < hc.n_events[e]; k++)
                        if (k < max_events) t_events[e * max_events + k] = root;
                    seen_events[e] = hc.n_events[e];
                }
            }
        }
        // t_eval samples inside (t_old, t]  (ivp.py:706-723)
        if (stepped) {
            while (eval_i < n_eval && t_eval[eval_i] <= hc.t) {
                double* dst = y_eval_dev + eval_i * sd;
                if (t_eval[eval_i] <= hc.t_old && hc.n_acc == 1 && t_eval[eval_i] == t0) {
                    HIP_OK(ctx, hipMemcpyAsync(dst, small ? ctx->buf[1] : ctx->buf[hc.cur ^ 1], sizeof(double) * sd, hipMemcpyDeviceToDevice, ctx->stream));
                } else if (int rc = dense_eval(ctx, v, layout, small, hc, t_eval[eval_i], dst, nullptr)) return rc;
                eval_i++;
            }
        } else {
            while (eval_i < n_eval && t_eval[eval_i] <= hc.t) {  // t0 == t1: no step was taken
                HIP_OK(ctx, hipMemcpyAsync(y_eval_dev + eval_i * sd, ctx->buf[small ? 0 : hc.cur], sizeof(double) * sd, hipMemcpyDeviceToDevice, ctx->stream));
                eval_i++;
            }
        }
        if (hc.status != ST_PAUSED) break;
        hipLaunchKernelGGL(rk45_resume_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dctrl, next_pause(), max_attempts);
        LAUNCH_OK(ctx);
    }
    ctrl_to_stats(hc, stats);
    if (!small && hc.cur != 0) {
        HIP_OK(ctx, hipMemcpyAsync(ctx->buf[0], ctx->buf[1], sizeof(double) * sd, hipMemcpyDeviceToDevice, ctx->stream));
        HIP_OK(ctx, hipMemcpyAsync(ctx->buf[2], ctx->buf[3], sizeof(double) * sd, hipMemcpyDeviceToDevice, ctx->stream));
    }
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" {

int marl_rhs_dev(marl_ctx* ctx, double t, const double* y_dev, double* dydt_dev, int layout)
{
    (void)t;
    if (!ctx || !y_dev || !dydt_dev || y_dev == dydt_dev) return ctx ? fail(ctx, -1, "marl_rhs_dev: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    return launch_rhs(ctx, y_dev, dydt_dev, layout);
}

int marl_rhs(marl_ctx* ctx, double t, const double* y, double* dydt)
{
    (void)t;
    if (!ctx || !y || !dydt) return ctx ? fail(ctx, -1, "marl_rhs: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)NF * ctx->N * ctx->batch;
    if (int rc = ensure(ctx, 0, n)) return rc;
    if (int rc = ensure(ctx, 1, n)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(ctx->buf[0], y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = launch_rhs(ctx, ctx->buf[0], ctx->buf[1], LAYOUT_FIELD_MAJOR)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(dydt, ctx->buf[1], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int marl_events_dev(marl_ctx* ctx, const double* y_dev, int layout, double* out)
{
    if (!ctx || !y_dev || !out) return ctx ? fail(ctx, -1, "marl_events_dev: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    if (int rc = launch_monitors(ctx, y_dev, layout)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(ctx->hrec, ctx->rec, sizeof(double) * NQ * ctx->batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < ctx->batch; b++) record_to_events(ctx->hrec + b * NQ, out + b * MARL_NEVENTS);
    return 0;
}

int marl_events(marl_ctx* ctx, const double* y, double* out)
{
    if (!ctx || !y || !out) return ctx ? fail(ctx, -1, "marl_events: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)NF * ctx->N * ctx->batch;
    if (int rc = ensure(ctx, 0, n)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(ctx->buf[0], y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return marl_events_dev(ctx, ctx->buf[0], LAYOUT_FIELD_MAJOR, out);
}

int marl_debug_math(marl_ctx* ctx, int op, const double* x_dev, double* y_dev, int64_t n, double e)
{
    if (!ctx || !x_dev || !y_dev || n < 0) return ctx ? fail(ctx, -1, "marl_debug_math: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    if (n == 0) return 0;
    hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, x_dev, y_dev, n, e);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_convert_layout_dev(marl_ctx* ctx, const double* src_dev, double* dst_dev, int src_layout, int dst_layout)
{
    if (!ctx || !src_dev || !dst_dev || src_dev == dst_dev) return ctx ? fail(ctx, -1, "marl_convert_layout_dev: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    return launch_convert(ctx, src_dev, dst_dev, src_layout, dst_layout);
}

int marl_integrate_rk4_dev(marl_ctx* ctx, double* y_dev, int layout, double dt, int64_t nsteps)
{
    if (!ctx || !y_dev || nsteps < 0) return ctx ? fail(ctx, -1, "marl_integrate_rk4_dev: invalid argument") : -1;
    if (ctx->batch != 1) return fail(ctx, -1, "marl_integrate_rk4_dev: single-instance context required (use marl_sweep_rk4_dev)");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    if (int rc = ensure(ctx, 1, (size_t)state_doubles(ctx->slab.n_buf, layout))) return rc;
    return rk4_run(ctx, y_dev, ctx->buf[1], layout, dt, nsteps);
}

int marl_sweep_rk4_dev(marl_ctx* ctx, double* y_dev, const double* dt, int64_t nsteps)
{
    if (!ctx || !y_dev || !dt || nsteps < 0) return ctx ? fail(ctx, -1, "marl_sweep_rk4_dev: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const int v = default_sweep_variant(ctx);
    if (v < 0) return fail(ctx, -1, "marl_sweep_rk4_dev: N = %lld exceeds the largest one-workgroup window (1024 cells)", (long long)ctx->N);
    // through the context's pinned buffer: `dt` may be a temporary of the caller (the previous use of hdt has long completed
    // in stream order only if we wait for it - a sweep launch is milliseconds, the wait is free)
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(ctx->hdt, dt, sizeof(double) * ctx->batch);
    HIP_OK(ctx, hipMemcpyAsync(ctx->ddt, ctx->hdt, sizeof(double) * ctx->batch, hipMemcpyHostToDevice, ctx->stream));
    const dim3 grid((unsigned)ctx->batch);
    SWEEP_DISPATCH(rk4_sweep_kernel, y_dev, ctx->dconsts, ctx->ddt, ctx->N, nsteps)
    LAUNCH_OK(ctx);
    return 0;
}

int marl_integrate_rk4(marl_ctx* ctx, double* y, double dt, int64_t nsteps)
{
    if (!ctx || !y || nsteps < 0) return ctx ? fail(ctx, -1, "marl_integrate_rk4: invalid argument") : -1;
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)NF * ctx->N * ctx->batch;
    if (int rc = ensure(ctx, 2, n)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(ctx->buf[2], y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (ctx->batch > 1 || default_sweep_variant(ctx) >= 0) {
        // small grids (one workgroup each, on-chip for all steps) and sweeps
        std::vector<double> dts((size_t)ctx->batch, dt);
        if (int rc = marl_sweep_rk4_dev(ctx, ctx->buf[2], dts.data(), nsteps)) return rc;
    } else {
        const int layout = (int)ctx->host_layout;
        const size_t sd = (size_t)state_doubles(ctx->N, layout);
        if (int rc = ensure(ctx, 0, sd)) return rc;
        if (int rc = ensure(ctx, 1, sd)) return rc;
        if (int rc = launch_convert(ctx, ctx->buf[2], ctx->buf[0], LAYOUT_FIELD_MAJOR, layout)) return rc;
        if (int rc = rk4_run(ctx, ctx->buf[0], ctx->buf[1], layout, dt, nsteps)) return rc;
        if (int rc = launch_convert(ctx, ctx->buf[0], ctx->buf[2], layout, LAYOUT_FIELD_MAJOR)) return rc;
    }
    HIP_OK(ctx, hipMemcpyAsync(y, ctx->buf[2], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return stream_check(ctx);
}

int marl_sweep_rk45_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol, double atol,
                        int64_t max_attempts, marl_stats* stats)
{
    if (!ctx || !y_dev || !stats) return ctx ? fail(ctx, -1, "marl_sweep_rk45_dev: invalid argument") : -1;
    if (!(first_step > 0) || !(t1 >= t0)) return fail(ctx, -1, "rk45: need first_step > 0 and t1 >= t0 (forward integration)");
    if (t1 > t0 && first_step > t1 - t0) return fail(ctx, -1, "rk45: `first_step` exceeds bounds");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const int v = default_sweep_variant(ctx);
    if (v < 0) return fail(ctx, -1, "marl_sweep_rk45_dev: N = %lld exceeds the largest one-workgroup window (1024 cells)", (long long)ctx->N);
    if (int rc = launch_monitors(ctx, y_dev, LAYOUT_FIELD_MAJOR)) return rc;
    hipLaunchKernelGGL(rk45_init_kernel, dim3((unsigned)ctx->batch), dim3(1), 0, ctx->stream, ctx->dctrl, ctx->rec, t0, t1, first_step, clamp_rtol(rtol),
                       atol, (int64_t)NF * ctx->N, max_attempts, 0);
    LAUNCH_OK(ctx);
    const dim3 grid((unsigned)ctx->batch);
    SWEEP_DISPATCH(rk45_sweep_kernel, y_dev, ctx->dconsts, ctx->dctrl, ctx->N, (double*)nullptr, (double*)nullptr)
    LAUNCH_OK(ctx);
    HIP_OK(ctx, hipMemcpyAsync(ctx->hctrl, ctx->dctrl, sizeof(Rk45Ctrl) * ctx->batch, hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < ctx->batch; b++) ctrl_to_stats(ctx->hctrl[b], &stats[b]);
    return 0;
}

int marl_integrate_rk45_dev(marl_ctx* ctx, double* y_dev, int layout, double t0, double t1, double first_step, double rtol,
                            double atol, int64_t max_attempts, marl_stats* stats)
{
    if (!ctx || !y_dev || !stats) return ctx ? fail(ctx, -1, "marl_integrate_rk45_dev: invalid argument") : -1;
    if (ctx->batch != 1) return fail(ctx, -1, "marl_integrate_rk45_dev: single-instance context required (use marl_sweep_rk45_dev)");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const size_t sd = (size_t)state_doubles(ctx->slab.n_buf, layout);
    for (int i = 0; i < 4; i++)
        if (int rc = ensure(ctx, i, sd)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(ctx->buf[0], y_dev, sd * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (int rc = rk45_run(ctx, layout, false, t0, t1, first_step, rtol, atol, nullptr, 0, nullptr, nullptr, 0, max_attempts, stats)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(y_dev, ctx->buf[0], sd * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int marl_integrate_rk45(marl_ctx* ctx, double* y, double t0, double t1, double first_step, double rtol, double atol,
                        const double* t_eval, int64_t n_eval, double* y_eval, double* t_events, int64_t max_events,
                        int64_t max_attempts, marl_stats* stats)
{
    if (!ctx || !y || !stats || (n_eval > 0 && (!t_eval || !y_eval))) return ctx ? fail(ctx, -1, "marl_integrate_rk45: invalid argument") : -1;
    if (ctx->batch != 1) return fail(ctx, -1, "marl_integrate_rk45: single-instance context required (use marl_sweep_rk45_dev)");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    const bool small = default_sweep_variant(ctx) >= 0;
    const int layout = small ? LAYOUT_FIELD_MAJOR : (int)ctx->host_layout;
    const size_t n = (size_t)NF * ctx->N;
    const size_t sd = (size_t)state_doubles(ctx->N, layout);
    for (int i = 0; i < 4; i++)
        if (int rc = ensure(ctx, i, sd > n ? sd : n)) return rc;
    // upload field-major into buf[1], convert into buf[0]
    HIP_OK(ctx, hipMemcpyAsync(ctx->buf[1], y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = launch_convert(ctx, ctx->buf[1], ctx->buf[0], LAYOUT_FIELD_MAJOR, layout)) return rc;
    double* yev = nullptr;
    if (n_eval > 0) HIP_OK(ctx, hipMalloc((void**)&yev, sizeof(double) * sd * (size_t)(n_eval + 1)));
    int rc = rk45_run(ctx, layout, small, t0, t1, first_step, rtol, atol, t_eval, n_eval, yev, t_events, max_events, max_attempts, stats);
    if (rc == 0) {
        // results back to field-major through a spare slot
        for (int64_t i = 0; i < n_eval && rc == 0; i++) {
            rc = launch_convert(ctx, yev + i * sd, yev + n_eval * sd, layout, LAYOUT_FIELD_MAJOR);
            if (rc == 0 && hipMemcpyAsync(y_eval + i * n, yev + n_eval * sd, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
                rc = fail(ctx, -3, "copy of t_eval sample failed");
            if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, -3, "sync failed");
        }
        if (rc == 0) rc = launch_convert(ctx, ctx->buf[0], ctx->buf[1], layout, LAYOUT_FIELD_MAJOR);
        if (rc == 0 && hipMemcpyAsync(y, ctx->buf[1], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
            rc = fail(ctx, -3, "copy of final state failed");
        if (rc == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, -3, "sync failed");
    }
    if (yev) (void)hipFree(yev);
    return rc;
}


// ---- domain decomposition building blocks --------------------------------------------------------
int marl_ctx_create_slab(const marl_params* params, int64_t N_global, int64_t g_begin, int64_t g_end, int64_t halo, int device,
                         marl_ctx** out)
{
    if (!params || !out || N_global < 2 || g_begin < 0 || g_end > N_global || g_end <= g_begin)
        return fail(nullptr, -1, "marl_ctx_create_slab: invalid argument");
    if (halo < 6) return fail(nullptr, -1, "marl_ctx_create_slab: halo must be >= 6 (one Dormand-Prince attempt consumes 6 cells per side)");
    if (g_end - g_begin < halo) return fail(nullptr, -1, "marl_ctx_create_slab: slab narrower than the halo");
    marl_ctx* ctx = nullptr;
    if (int rc = marl_ctx_create(params, 1, N_global, device, &ctx)) return rc;
    const int64_t hl = g_begin > 0 ? halo : 0, hr = g_end < N_global ? halo : 0, n_own = g_end - g_begin;
    ctx->halo = (int)halo;
    ctx->slab = Slab{hl + n_own + hr, g_begin - hl, hl + n_own + hr, hl, hl + n_own};
    *out = ctx;
    return 0;
}

#define SLAB_OK(ctx, fn)                                                                       \
    if (!(ctx)) return -1;                                                                     \
    if ((ctx)->halo <= 0) return fail(ctx, -1, fn ": not a slab context (marl_ctx_create_slab)"); \
    HIP_OK(ctx, hipSetDevice((ctx)->device));

int marl_slab_load(marl_ctx* ctx, const double* y_owned_dev)
{
    SLAB_OK(ctx, "marl_slab_load")
    if (!y_owned_dev) return fail(ctx, -1, "marl_slab_load: invalid argument");
    const size_t sd = (size_t)NF * ctx->slab.n_buf;
    for (int i = 0; i < 4; i++) {
        if (int rc = ensure(ctx, i, sd)) return rc;
        HIP_OK(ctx, hipMemsetAsync(ctx->buf[i], 0, sd * sizeof(double), ctx->stream));
    }
    const int64_t n_own = ctx->slab.out_hi - ctx->slab.out_lo;
    hipLaunchKernelGGL(slab_copy_kernel, dim3((unsigned)((n_own + 255) / 256)), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->dctrl, 0,
                       ctx->slab, const_cast<double*>(y_owned_dev), 0);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_slab_store(marl_ctx* ctx, double* y_owned_dev)
{
    SLAB_OK(ctx, "marl_slab_store")
    if (!y_owned_dev) return fail(ctx, -1, "marl_slab_store: invalid argument");
    const int64_t n_own = ctx->slab.out_hi - ctx->slab.out_lo;
    hipLaunchKernelGGL(slab_copy_kernel, dim3((unsigned)((n_own + 255) / 256)), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->dctrl, -2,
                       ctx->slab, y_owned_dev, 1);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_slab_pack(marl_ctx* ctx, int which, double* send_lo_dev, double* send_hi_dev)
{
    SLAB_OK(ctx, "marl_slab_pack")
    hipLaunchKernelGGL(slab_pack_kernel, dim3(1), dim3(128), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dctrl, which,
                       ctx->slab, ctx->halo, send_lo_dev, send_hi_dev);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_slab_unpack(marl_ctx* ctx, int which, const double* recv_lo_dev, const double* recv_hi_dev)
{
    SLAB_OK(ctx, "marl_slab_unpack")
    // a side without a neighbour (physical boundary) has no halo cells: ignore its strip
    const double* lo = ctx->slab.out_lo > 0 ? recv_lo_dev : nullptr;
    const double* hi = ctx->slab.out_hi < ctx->slab.n_buf ? recv_hi_dev : nullptr;
    hipLaunchKernelGGL(slab_unpack_kernel, dim3(1), dim3(128), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dctrl, which,
                       ctx->slab, ctx->halo, lo, hi);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_slab_rhs0(marl_ctx* ctx)
{
    SLAB_OK(ctx, "marl_slab_rhs0")
    return launch_rhs(ctx, ctx->buf[0], ctx->buf[2], LAYOUT_FIELD_MAJOR);
}

int marl_slab_monitors(marl_ctx* ctx, double* rec_dev)
{
    SLAB_OK(ctx, "marl_slab_monitors")
    if (!rec_dev) rec_dev = ctx->dd_send;   // library-side exchange: straight into this rank's message
    if (!rec_dev) return fail(ctx, -1, "marl_slab_monitors: invalid argument");
    if (int rc = launch_monitors(ctx, ctx->buf[0], LAYOUT_FIELD_MAJOR)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(rec_dev, ctx->rec, sizeof(double) * NQ, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

int marl_slab_init_control(marl_ctx* ctx, const double* recs_dev, int64_t nrec, double t0, double t1, double first_step, double rtol,
                           double atol, int64_t max_attempts)
{
    SLAB_OK(ctx, "marl_slab_init_control")
    if (!recs_dev) recs_dev = ctx->dd_recs;   // library-side exchange: the records gathered by marl_slab_exchange
    if (!recs_dev || nrec < 1) return fail(ctx, -1, "marl_slab_init_control: invalid argument");
    if (!(first_step > 0) || !(t1 >= t0) || (t1 > t0 && first_step > t1 - t0)) return fail(ctx, -1, "rk45: `first_step` must be in (0, t1 - t0]");
    hipLaunchKernelGGL(reduce_records_kernel, dim3(1), dim3(256), 0, ctx->stream, recs_dev, nrec, ctx->rec);
    LAUNCH_OK(ctx);
    hipLaunchKernelGGL(rk45_init_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->dctrl, ctx->rec, t0, t1, first_step, clamp_rtol(rtol), atol,
                       (int64_t)NF * ctx->N, max_attempts, 0);
    LAUNCH_OK(ctx);
    ctx->dd_max_attempts = max_attempts;
    return 0;
}

int marl_slab_attempt(marl_ctx* ctx, double* rec_dev)
{
    SLAB_OK(ctx, "marl_slab_attempt")
    if (!rec_dev) return fail(ctx, -1, "marl_slab_attempt: invalid argument");
    const int v = default_rk45_variant(ctx);
    const int64_t nb = rk45_blocks(ctx, v);
    if (int rc = ensure_part(ctx, (size_t)std::max<int64_t>(nb + kReduceGroups, 1024))) return rc;
    if (ctx->var_dphi) launch_attempt_t<256, 1, true>(ctx, nb, LAYOUT_FIELD_MAJOR); else launch_attempt_t<256, 1>(ctx, nb, LAYOUT_FIELD_MAJOR);
    LAUNCH_OK(ctx);
    int64_t nrec = nb;
    const double* recs = reduce_first_level(ctx, &nrec);
    LAUNCH_OK(ctx);
    // one record for this rank; harmless (stale partials) when the controller is no longer running: control ignores it
    hipLaunchKernelGGL(reduce_records_kernel, dim3(1), dim3(256), 0, ctx->stream, recs, nrec, rec_dev);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_slab_control(marl_ctx* ctx, const double* recs_dev, int64_t nrec)
{
    SLAB_OK(ctx, "marl_slab_control")
    if (!recs_dev || nrec < 1) return fail(ctx, -1, "marl_slab_control: invalid argument");
    hipLaunchKernelGGL(rk45_control_kernel, dim3(1), dim3(CONTROL_THREADS), 0, ctx->stream, recs_dev, nrec, ctx->dctrl);
    LAUNCH_OK(ctx);
    return 0;
}

int marl_slab_status(marl_ctx* ctx, marl_stats* stats)
{
    SLAB_OK(ctx, "marl_slab_status")
    if (!stats) return fail(ctx, -1, "marl_slab_status: invalid argument");
    HIP_OK(ctx, hipMemcpyAsync(ctx->hctrl, ctx->dctrl, sizeof(Rk45Ctrl), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    ctrl_to_stats(*ctx->hctrl, stats);
    return 0;
}

}  // extern "C"
// ---- domain decomposition with the exchange inside the library -------------------------------------------------------------
// The RCCL API is taken from the librccl the process already has (PyTorch-ROCm bundles one; `rccl_path` names it) through
// dlopen - this library does not link RCCL, single-GPU users never load it.
namespace {
constexpr int kNcclDouble = 8;   // ncclFloat64 (rccl.h)
struct NcclId { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)

void* open_rccl(const char* path)
{
    void* lib = nullptr;
    if (path && *path) lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return lib;
}

inline int dd_msg(const marl_ctx* ctx) { return NQ + 2 * (2 * NF * ctx->halo); }

int dd_buffers(marl_ctx* ctx)
{
    if (ctx->dd_send) return 0;
    const size_t msg = (size_t)dd_msg(ctx);
    HIP_OK(ctx, hipMalloc((void**)&ctx->dd_send, msg * sizeof(double)));
    HIP_OK(ctx, hipMalloc((void**)&ctx->dd_gathered, msg * sizeof(double) * (size_t)ctx->dd_world));
    HIP_OK(ctx, hipMalloc((void**)&ctx->dd_recs, NQ * sizeof(double) * (size_t)ctx->dd_world));
    HIP_OK(ctx, hipMemsetAsync(ctx->dd_send, 0, msg * sizeof(double), ctx->stream));
    return 0;
}

// all-gather of the ranks' messages (one slab: the message itself is the gathered buffer)
int dd_allgather(marl_ctx* ctx, const double** gathered)
{
    if (!ctx->rccl_comm) { *gathered = ctx->dd_send; return 0; }   // one slab, no communicator
    const int rc = ctx->rccl_allgather(ctx->dd_send, ctx->dd_gathered, (size_t)dd_msg(ctx), kNcclDouble, ctx->rccl_comm, ctx->stream);
    if (rc != 0) return fail(ctx, -3, "ncclAllGather failed (ncclResult %d)", rc);
    *gathered = ctx->dd_gathered;
    return 0;
}
}  // namespace

extern "C" {

int marl_slab_comm_probe(const char* rccl_path)
{
    void* lib = open_rccl(rccl_path);
    if (!lib) return fail(nullptr, -2, "marl_slab_comm_probe: cannot load librccl (%s)", dlerror());
    for (const char* sym : {"ncclGetUniqueId", "ncclCommInitRank", "ncclAllGather", "ncclCommDestroy"})
        if (!dlsym(lib, sym)) return fail(nullptr, -2, "marl_slab_comm_probe: %s not found in librccl", sym);
    return 0;
}

int marl_slab_comm_id(const char* rccl_path, char id_out[128])
{
    if (!id_out) return -1;
    void* lib = open_rccl(rccl_path);
    if (!lib) return fail(nullptr, -2, "marl_slab_comm_id: cannot load librccl (%s)", dlerror());
    auto get_id = (int (*)(NcclId*))dlsym(lib, "ncclGetUniqueId");
    if (!get_id) return fail(nullptr, -2, "marl_slab_comm_id: ncclGetUniqueId not found");
    NcclId id;
    const int rc = get_id(&id);
    if (rc != 0) return fail(nullptr, -3, "ncclGetUniqueId failed (ncclResult %d)", rc);
    memcpy(id_out, id.internal, 128);
    return 0;
}

int marl_slab_comm_init(marl_ctx* ctx, const char* rccl_path, const char id[128], int rank, int world)
{
    SLAB_OK(ctx, "marl_slab_comm_init")
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, -1, "marl_slab_comm_init: invalid rank / world");
    if (ctx->dd_send) return fail(ctx, -1, "marl_slab_comm_init: already initialised");
    ctx->dd_rank = rank;
    ctx->dd_world = world;
    if (world > 1 || id) {   // (world == 1 with an id: a one-rank communicator - exercises the RCCL path on one GPU)
        if (!id) return fail(ctx, -1, "marl_slab_comm_init: a unique id is required for world > 1");
        bool zero = true;
        for (int i = 0; i < 128; i++) zero = zero && id[i] == 0;
        if (zero) return fail(ctx, -1, "marl_slab_comm_init: the unique id is all zero (not made by marl_slab_comm_id / ncclGetUniqueId)");
        ctx->rccl_lib = open_rccl(rccl_path);
        if (!ctx->rccl_lib) return fail(ctx, -2, "marl_slab_comm_init: cannot load librccl (%s)", dlerror());
        auto init_rank = (int (*)(void**, int, NcclId, int))dlsym(ctx->rccl_lib, "ncclCommInitRank");
        ctx->rccl_allgather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(ctx->rccl_lib, "ncclAllGather");
        ctx->rccl_destroy = (int (*)(void*))dlsym(ctx->rccl_lib, "ncclCommDestroy");
        if (!init_rank || !ctx->rccl_allgather || !ctx->rccl_destroy) return fail(ctx, -2, "marl_slab_comm_init: RCCL symbols not found");
        NcclId nid;
        memcpy(nid.internal, id, 128);
        const int rc = init_rank(&ctx->rccl_comm, world, nid, rank);
        if (rc != 0) { ctx->rccl_comm = nullptr; return fail(ctx, -3, "ncclCommInitRank failed (ncclResult %d)", rc); }
    }
    if (world > 1 && !ctx->rccl_comm) return fail(ctx, -1, "marl_slab_comm_init: no communicator for world > 1");
    return dd_buffers(ctx);
}

// pack(which) -> all-gather -> unpack(which); the ranks' records end up in the context (marl_slab_init_control with recs NULL)
int marl_slab_exchange(marl_ctx* ctx, int which)
{
    SLAB_OK(ctx, "marl_slab_exchange")
    if (!ctx->dd_send) return fail(ctx, -1, "marl_slab_exchange: call marl_slab_comm_init first");
    hipLaunchKernelGGL(slab_reduce_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dctrl, which,
                       ctx->slab, ctx->halo, ctx->part, (int64_t)0, ctx->dd_send);
    LAUNCH_OK(ctx);
    const double* gathered;
    if (int rc = dd_allgather(ctx, &gathered)) return rc;
    hipLaunchKernelGGL(slab_unpack_control_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dctrl, which,
                       ctx->slab, ctx->halo, gathered, ctx->dd_rank, ctx->dd_world, dd_msg(ctx), 0);
    LAUNCH_OK(ctx);
    hipLaunchKernelGGL(slab_records_kernel, dim3(1), dim3(64), 0, ctx->stream, gathered, ctx->dd_world, dd_msg(ctx), ctx->dd_recs);
    LAUNCH_OK(ctx);
    return 0;
}

// The whole adaptive loop of a domain-decomposed run: per attempt  attempt kernel -> reduce + pack -> all-gather -> unpack +
// control, enqueued `poll_interval` attempts at a time; the status is read once per batch (every rank reads the same status:
// all decisions are computed from the same gathered records).  Before: load, exchange(0), rhs0, monitors(NULL), exchange(0),
// init_control(NULL, world, ...).  After: store.
int marl_slab_run(marl_ctx* ctx, marl_stats* stats)
{
    SLAB_OK(ctx, "marl_slab_run")
    if (!stats) return fail(ctx, -1, "marl_slab_run: invalid argument");
    if (!ctx->dd_send) return fail(ctx, -1, "marl_slab_run: call marl_slab_comm_init first");
    const int v = default_rk45_variant(ctx);
    const int64_t nb = rk45_blocks(ctx, v);
    if (int rc = ensure_part(ctx, (size_t)std::max<int64_t>(nb + kReduceGroups, 1024))) return rc;
    const int msg = dd_msg(ctx);
    int64_t executed = 0;
    if (ctx->dd_world == 1 && !ctx->rccl_comm) {
        // ONE slab and no communicator: nothing is exchanged - the single-grid integrator's loop (same kernels, same bits): the persistent
        // launch up to ~750 000 cells, attempt + reduction + control launches above (one launch less per attempt than packing a message
        // nobody reads and unpacking it again)
        const bool streamed = rk45_use_stream(ctx, v);
        while (true) {
            if (streamed) {
                if (int rc = launch_rk45_stream(ctx, v, LAYOUT_FIELD_MAJOR, ctx->rk45_stream_attempts)) return rc;
            } else {
                const int64_t batch = attempts_per_batch(ctx->poll, ctx->dd_max_attempts, executed);
                for (int64_t i = 0; i < batch; i++)
                    if (int rc = launch_attempt(ctx, v, LAYOUT_FIELD_MAJOR)) return rc;
            }
            HIP_OK(ctx, hipMemcpyAsync(ctx->hctrl, ctx->dctrl, sizeof(Rk45Ctrl), hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
            const int64_t before = executed;
            executed = ctx->hctrl->attempts - (ctx->hctrl->status == ST_RUNNING ? 1 : 0);
            if (streamed)
                if (int rc = rk45_stream_account(ctx, executed - before)) return rc;
            if (ctx->hctrl->status != ST_RUNNING) break;
        }
        ctrl_to_stats(*ctx->hctrl, stats);
        return 0;
    }
    // the slab's attempt as ONE launch (rk45_stream_kernel, one attempt, its last workgroup packs the message): two launches + the
    // all-gather per attempt instead of four.  A batch of them is enqueued blindly (attempts after the controller has stopped return
    // at once), so the counters are re-armed by the kernel itself; they must be at zero when the first one starts.
    const bool dd_stream = ctx->dd_stream == 2 || (ctx->dd_stream == 1 && rk45_use_stream(ctx, v));
    if (dd_stream) {
        if (int rc = rk45_stream_setup(ctx)) return rc;
        if (ctx->rs_arrive_base || ctx->rs_epoch_base || ctx->rs_grab_base)
            if (int rc = rk45_stream_reset(ctx)) return rc;
    }
    while (true) {
        // (every rank computes the same batch size: the budget and the status are the same everywhere)
        const int64_t batch = attempts_per_batch(ctx->poll, ctx->dd_max_attempts, executed);
        for (int64_t i = 0; i < batch; i++) {
            if (dd_stream) {
                if (int rc = launch_rk45_stream(ctx, v, LAYOUT_FIELD_MAJOR, 1, true)) return rc;
            } else {
                if (ctx->var_dphi) launch_attempt_t<256, 1, true>(ctx, nb, LAYOUT_FIELD_MAJOR); else launch_attempt_t<256, 1>(ctx, nb, LAYOUT_FIELD_MAJOR);
                LAUNCH_OK(ctx);
                int64_t nrec = nb;
                const double* recs = reduce_first_level(ctx, &nrec);
                LAUNCH_OK(ctx);
                hipLaunchKernelGGL(slab_reduce_pack_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dctrl, -1,
                                   ctx->slab, ctx->halo, recs, nrec, ctx->dd_send);
                LAUNCH_OK(ctx);
            }
            const double* gathered;
            if (int rc = dd_allgather(ctx, &gathered)) return rc;
            hipLaunchKernelGGL(slab_unpack_control_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->buf[0], ctx->buf[1], ctx->buf[2], ctx->buf[3], ctx->dctrl,
                               -1, ctx->slab, ctx->halo, gathered, ctx->dd_rank, ctx->dd_world, msg, 1);
            LAUNCH_OK(ctx);
        }
        HIP_OK(ctx, hipMemcpyAsync(ctx->hctrl, ctx->dctrl, sizeof(Rk45Ctrl), hipMemcpyDeviceToHost, ctx->stream));
        HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->hctrl->status != ST_RUNNING) break;
        executed = ctx->hctrl->attempts - 1;
    }
    ctrl_to_stats(*ctx->hctrl, stats);
    return 0;
}

}  // extern "C"

// ==============================================================================================
// Implicit path: scipy's Radau as the reference runs it by default (marlpde/parameters.py:213, jac_sparsity :150-199;
// call site marlpde/Evolve_scenario.py:104-109).  Host = the scalar step logic of scipy/integrate/_ivp/radau.py
// (_step_impl :404-537, solve_collocation_system :47-130, predict_factor :133-173) and of the solve_ivp driver
// (ivp.py:654-723: events with Brent on the dense output, t_eval); device = marl_radau.h.
// ==============================================================================================
namespace {
using radau::cplx;

constexpr int kRadauPartials = 256;   // workgroups of the norm + update kernels on large grids

struct RadauWork {
    double *y, *f, *fnew, *ynew, *err, *yerr, *yold, *scale, *tmp;
    double *Z, *W, *F, *Z0, *Q, *YS;
    double *fac, *h, *yscale, *maxdiff, *scl, *hnew, *Jraw, *YP, *FN;
    double *J, *Dinv_r, *Up_r, *rhs_r, *out, *partial;
    cplx *Dinv_c, *Up_c, *rhs_c;
    int32_t *small, *groups, *flags;
    int ng = 0;
    bool have_factor = false;
    // block parallel cyclic reduction (default linear solver)
    radau::PcrSystem<double> Sr{};
    radau::PcrSystem<cplx> Sc{};
    int nlevels = 0;
    bool pcr = true;
    // cyclic reduction in front of it (marl_radau_cr.h; large grids of single runs): cr_k levels, level l has cr_n[l] rows stored from row
    // cr_off[l]; Sr / Sc / nlevels then describe the COMPACT system of the cr_n[cr_k] rows that are left
    int cr_k = 0;
    int64_t cr_n[radau::CR_MAX_LEVELS + 1] = {}, cr_off[radau::CR_MAX_LEVELS + 2] = {};
    radau::CrSystem<double> Cr{};
    radau::CrSystem<cplx> Cc{};
    radau::CrPlan plan{};   // the same levels for the one-launch solve kernels (plan.k == 0: they run PCR alone)
};

// Zero-copy slots: arm = store the sentinel; wait = poll until the kernel has overwritten it (bounded, then fall back to a synchronise).
constexpr uint64_t kZcSentinel = 0x7ff8dead5eed0001ull;   // a quiet NaN with a payload arithmetic does not produce
inline void zc_arm(marl_ctx* ctx, int slot, int count = 1)
{
    for (int i = 0; i < count; i++) reinterpret_cast<volatile uint64_t*>(ctx->zc_h)[slot + i] = kZcSentinel;
}
inline int zc_wait(marl_ctx* ctx, int slot, int count = 1)
{
    volatile uint64_t* p = reinterpret_cast<volatile uint64_t*>(ctx->zc_h);
    for (int i = 0; i < count; i++) {
        int64_t spins = 0;
        while (p[slot + i] == kZcSentinel) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
            if (++spins > (int64_t)1 << 28) {   // ~seconds: something is wrong with the stream - let the runtime say what
                HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
                if (p[slot + i] == kZcSentinel) return fail(ctx, -3, "implicit driver: a result word was never written");
            }
        }
    }
    return 0;
}

// `instances` > 1: one arena per instance, all with the layout of instance 0 (`w`), `*zstride` bytes apart
int radau_alloc(marl_ctx* ctx, RadauWork& w, const int32_t* groups_host, int64_t instances = 1, int64_t* zstride = nullptr)
{
    const int64_t N = ctx->N, n = NF * N;
    std::vector<int32_t> g(n);
    if (groups_host) g.assign(groups_host, groups_host + n);
    else for (int64_t j = 0; j < n; j++) g[j] = (int32_t)(3 * (j / N) + (j % N) % 3);   // structured colouring: 15 groups
    int ng = 0;
    for (int64_t j = 0; j < n; j++) {
        if (g[j] < 0 || g[j] > 4096) return fail(ctx, -1, "radau: invalid column group %d", (int)g[j]);
        ng = std::max(ng, g[j] + 1);
    }
    // a valid grouping never puts two columns that share a pattern row (same cell neighbourhood) into one group
    for (int64_t j = 0; j < n; j++) {
        const int64_t ip = j % N;
        for (int fp = 0; fp < NF; fp++)
            for (int64_t i2 = std::max<int64_t>(0, ip - 2); i2 <= std::min<int64_t>(N - 1, ip + 2); i2++) {
                const int64_t j2 = fp * N + i2;
                if (j2 != j && g[j2] == g[j]) return fail(ctx, -1, "radau: columns %lld and %lld share rows but are in one group", (long long)j, (long long)j2);
            }
    }
    // cyclic-reduction levels in front of PCR (single runs with the PCR solver only)
    w.cr_k = 0;
    w.cr_n[0] = N; w.cr_off[0] = 0;
    // Small grids whose solves run in one workgroup (5 N <= PCR_FUSED_MAX, single runs and sweeps): radau_cr_small levels (default 3) -
    // the chain of levels in one workgroup is bound by the factor bytes that pass through one compute unit, and cyclic reduction
    // in front of PCR cuts them 2.4 - 2.8 times (marl_radau.h, crpcr_solve_all).  The all-in-workgroup sweep mode keeps plain PCR.
    // By default only from radau_cr_small_min_n = 205 cells (two unknowns per thread of the chain): on the reference's N = 200 grid
    // plain PCR is what takes every decision scipy takes in single runs AND sweeps (with 1 - 3 levels in front, one knife-edge Newton
    // test or another falls the other way in one of the pinned cases - profiles/r03_lab_radau_wg.log); set radau_cr_small_min_n = 32
    // for throughput there (4096 scenarios: 1.58 -> 1.17 s).
    const bool one_wg = n <= radau::PCR_FUSED_MAX && N >= 32 && (instances == 1 || ctx->radau_fused_solve);   // (single runs with radau_fused_solve = 0: the same levels, one launch each)
    const bool small_cr = one_wg && ctx->radau_cr < 0 && ctx->radau_cr_small > 0 && N >= ctx->radau_cr_small_min_n && !(instances > 1 && ctx->radau_sweep_wg == 2);
    const bool large_cr = instances == 1 && ctx->radau_cr != 0 && (ctx->radau_cr > 0 || N >= ctx->radau_cr_min_n);
    if (ctx->radau_solver == 0 && (small_cr || large_cr)) {
        const int64_t want = small_cr ? ctx->radau_cr_small : ctx->radau_cr;
        int k = 0;
        while (k < radau::CR_MAX_LEVELS && w.cr_n[k] / 2 >= 2 && (want > 0 ? k < want : NF * w.cr_n[k] > radau::PCR_FUSED_THREADS)) {
            w.cr_n[k + 1] = w.cr_n[k] / 2;
            w.cr_off[k + 1] = w.cr_off[k] + w.cr_n[k];
            k++;
        }
        w.cr_k = k;
    }
    // the levels as the one-launch solve kernels take them (k <= CR_WG_MAX_LEVELS; else they are not used)
    w.plan = radau::CrPlan{};
    if (w.cr_k > 0 && w.cr_k <= radau::CR_WG_MAX_LEVELS && n <= radau::PCR_FUSED_MAX) {
        w.plan.k = w.cr_k;
        for (int l = 0; l <= w.cr_k; l++) { w.plan.n[l] = w.cr_n[l]; w.plan.off[l] = w.cr_off[l]; }
    }
    const int64_t M = w.cr_n[w.cr_k];                                      // rows of the system PCR works on
    const int64_t cr_rows = w.cr_k ? w.cr_off[w.cr_k] + M : 0;             // rows of all levels together (< 2 N)
    int nlev = 0;
    while (((int64_t)1 << nlev) < M) nlev++;
    const size_t pcr_real = (size_t)M * 25 * (8 + 2 * (size_t)nlev) + 2 * (size_t)NF * M;   // L, D, U, Dinv ping-pong; alpha, gamma per level; b ping-pong
    const size_t cr_real = (size_t)cr_rows * (8 * 25 + NF);                             // L, D, U, Dinv, P, Q, alpha, gamma; b
    const size_t doubles = (size_t)n * (9 + 6 * 3 + 6 + 15 + 2 * (size_t)ng + 15 + 10 + 20 + 1 + 2) + 64 + kRadauPartials + (size_t)n + 3 * pcr_real + 3 * cr_real;
    const size_t per = (doubles + 1) & ~(size_t)1;   // 16-byte multiples: complex members stay aligned in every instance
    if (zstride) *zstride = (int64_t)(per * sizeof(double));
    if (ctx->rd_cap < per * (size_t)instances) {
        if (ctx->rd_arena) HIP_OK(ctx, hipFree(ctx->rd_arena));
        ctx->rd_arena = nullptr; ctx->rd_cap = 0;
        HIP_OK(ctx, hipMalloc((void**)&ctx->rd_arena, per * (size_t)instances * sizeof(double)));
        ctx->rd_cap = per * (size_t)instances;
    }
    if (!ctx->rd_host) HIP_OK(ctx, hipHostMalloc((void**)&ctx->rd_host, 16 * sizeof(double), hipHostMallocDefault));
    double* p = ctx->rd_arena;
    auto take = [&](size_t k) { double* r = p; p += k; return r; };
    w.y = take(n); w.f = take(n); w.fnew = take(n); w.ynew = take(n); w.err = take(n); w.yerr = take(n); w.yold = take(n); w.scale = take(n); w.tmp = take(n);
    w.Z = take(3 * n); w.W = take(3 * n); w.F = take(3 * n); w.Z0 = take(3 * n); w.Q = take(3 * n); w.YS = take(3 * n);
    w.fac = take(n); w.h = take(n); w.yscale = take(n); w.maxdiff = take(n); w.scl = take(n); w.hnew = take(n); w.Jraw = take(15 * n);
    w.YP = take((size_t)ng * n); w.FN = take((size_t)ng * n);
    w.J = take(15 * n); w.Dinv_r = take(5 * n); w.Up_r = take(5 * n);
    w.Dinv_c = (cplx*)take(10 * n); w.Up_c = (cplx*)take(10 * n);
    w.rhs_r = take(n); w.rhs_c = (cplx*)take(2 * n); w.out = take(8); w.partial = take(kRadauPartials);
    w.small = (int32_t*)take((n + 1) / 2 + 1); w.groups = (int32_t*)take((n + 1) / 2 + 1); w.flags = (int32_t*)take(2);
    ctx->zc_on = false;
    if (instances == 1 && ctx->implicit_zero_copy) {
        if (!ctx->zc_h) {
            HIP_OK(ctx, hipHostMalloc((void**)&ctx->zc_h, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
            HIP_OK(ctx, hipHostGetDevicePointer((void**)&ctx->zc_d, ctx->zc_h, 0));
        }
        HIP_OK(ctx, hipStreamSynchronize(ctx->stream));   // nothing of an earlier run may still write the words
        memset(ctx->zc_h, 0, 32 * sizeof(double));
        zc_arm(ctx, 0);
        w.out = ctx->zc_d;
        w.flags = reinterpret_cast<int32_t*>(ctx->zc_d + 1);
        ctx->zc_on = true;
    }
    w.ng = ng;
    w.nlevels = nlev;
    w.pcr = ctx->radau_solver == 0;
    for (int k = 0; k < 2; k++) {
        w.Sr.L[k] = take(25 * M); w.Sr.D[k] = take(25 * M); w.Sr.U[k] = take(25 * M); w.Sr.Dinv[k] = take(25 * M); w.Sr.b[k] = take(NF * M);
        w.Sc.L[k] = (cplx*)take(50 * M); w.Sc.D[k] = (cplx*)take(50 * M); w.Sc.U[k] = (cplx*)take(50 * M); w.Sc.Dinv[k] = (cplx*)take(50 * M);
        w.Sc.b[k] = (cplx*)take(2 * NF * M);
    }
    w.Sr.alpha = take((size_t)nlev * 25 * M); w.Sr.gamma = take((size_t)nlev * 25 * M);
    w.Sc.alpha = (cplx*)take((size_t)nlev * 50 * M); w.Sc.gamma = (cplx*)take((size_t)nlev * 50 * M);
    if (w.cr_k) {
        const size_t R = (size_t)cr_rows;
        double** pr[8] = {&w.Cr.L, &w.Cr.D, &w.Cr.U, &w.Cr.Dinv, &w.Cr.P, &w.Cr.Q, &w.Cr.alpha, &w.Cr.gamma};
        cplx** pc[8] = {&w.Cc.L, &w.Cc.D, &w.Cc.U, &w.Cc.Dinv, &w.Cc.P, &w.Cc.Q, &w.Cc.alpha, &w.Cc.gamma};
        for (int a = 0; a < 8; a++) { *pr[a] = take(25 * R); *pc[a] = (cplx*)take(50 * R); }
        w.Cr.b = take(NF * R); w.Cc.b = (cplx*)take(2 * NF * R);
    }
    if (instances > 1) HIP_OK(ctx, hipMemsetAsync(ctx->rd_arena, 0, per * (size_t)instances * sizeof(double), ctx->stream));
    else HIP_OK(ctx, hipMemsetAsync(w.J, 0, sizeof(double) * 15 * n, ctx->stream));
    HIP_OK(ctx, hipMemcpyAsync(w.groups, g.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, ctx->stream));   // (one copy serves every instance)
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));   // g dies with this frame
    return 0;
}

inline unsigned blocks256(int64_t n) { return (unsigned)((n + 255) / 256); }
// workgroups of the norm kernels: one up to 8192 unknowns, then one per 4096, at most kRadauPartials
inline int radau_norm_blocks(int64_t n) { return n <= 8192 ? 1 : (int)std::min<int64_t>(256, (n + 4095) / 4096); }

inline unsigned pcr_groups(int64_t rows) { return (unsigned)((rows + radau::PCR_CELLS_PER_BLOCK - 1) / radau::PCR_CELLS_PER_BLOCK); }

// factorise  mu I - jscale J  for the real system (systems = 1) or the real and the complex one (2): block PCR over all N rows
// (1 + ceil(log2 N) launches), or - large grids of single runs - cr_k levels of cyclic reduction and PCR over the rows that are left
int pcr_factor_launch(marl_ctx* ctx, RadauWork& w, double mu_r, cplx mu_c, double jscale, int systems)
{
    const int64_t N = ctx->N;
    const ZBatch none{0, nullptr, 0, 0, nullptr};
    if (!w.cr_k) {
        const dim3 grid(pcr_groups(N), systems);
        for (int level = -1; level < w.nlevels; level++) {
            hipLaunchKernelGGL(radau::pcr_factor_kernel, grid, dim3(256), 0, ctx->stream, w.J, N, level, mu_r, mu_c, w.Sr, w.Sc, none, jscale);
            LAUNCH_OK(ctx);
        }
        return 0;
    }
    hipLaunchKernelGGL(radau::cr_init_kernel, dim3(pcr_groups((N + 1) / 2), systems), dim3(256), 0, ctx->stream, w.J, N, mu_r, mu_c, jscale, w.Cr, w.Cc);
    LAUNCH_OK(ctx);
    for (int l = 0; l < w.cr_k; l++) {
        const radau::CrShape sh{w.cr_n[l], w.cr_off[l], w.cr_n[l + 1], w.cr_off[l + 1]};
        hipLaunchKernelGGL(radau::cr_reduce_kernel, dim3(pcr_groups(sh.n_next), systems), dim3(256), 0, ctx->stream, mu_r, mu_c, w.Cr, w.Cc, sh,
                           l + 1 == w.cr_k ? 1 : 0, w.Sr, w.Sc);
        LAUNCH_OK(ctx);
    }
    const int64_t M = w.cr_n[w.cr_k];
    for (int level = 0; level < w.nlevels; level++) {   // (level -1 - blocks from J and their inverses - is what the last reduction left in set 0)
        hipLaunchKernelGGL(radau::pcr_factor_kernel, dim3(pcr_groups(M), systems), dim3(256), 0, ctx->stream, w.J, M, level, mu_r, mu_c, w.Sr, w.Sc, none, jscale);
        LAUNCH_OK(ctx);
    }
    return 0;
}

// PCR solve of M rows, in place in x_r (and, with `both`, x_c)
int pcr_solve_launch(marl_ctx* ctx, RadauWork& w, int64_t M, bool both, double* x_r, cplx* x_c)
{
    const int64_t n = NF * M;
    if (n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve) {   // small systems: every level in one launch
        hipLaunchKernelGGL(radau::pcr_solve_fused_kernel, dim3(1, both ? 2 : 1), dim3(radau::PCR_FUSED_THREADS), 0, ctx->stream, M, w.nlevels, 0, w.Sr, w.Sc,
                           x_r, x_r, x_c, x_c);
        LAUNCH_OK(ctx);
        return 0;
    }
    const dim3 grid(blocks256(n), both ? 2 : 1);
    // ping-pong: x -> b[0] -> b[1] -> ... ; the last launch (x = D^-1 b) writes back into x
    const double* in_r = x_r;
    const cplx* in_c = x_c;
    for (int level = 0; level <= w.nlevels; level++) {
        double* out_r = (level == w.nlevels) ? x_r : w.Sr.b[level & 1];
        cplx* out_c = (level == w.nlevels) ? x_c : w.Sc.b[level & 1];
        hipLaunchKernelGGL(radau::pcr_solve_kernel, grid, dim3(256), 0, ctx->stream, M, level, w.nlevels, 0, w.Sr, w.Sc, in_r, out_r, in_c, out_c);
        LAUNCH_OK(ctx);
        in_r = out_r;
        in_c = out_c;
    }
    return 0;
}

// factorise  mu_r I - J  and  mu_c I - J  (block PCR: 1 + ceil(log2 N) launches; or sequential block Thomas: 1 launch)
int radau_factor(marl_ctx* ctx, RadauWork& w, double mu_r, cplx mu_c)
{
#ifdef MARL_LAB_BLOCK_THOMAS
    const int64_t N = ctx->N;
    if (!w.pcr) {
        hipLaunchKernelGGL(radau::factor_kernel, dim3(2), dim3(64), 0, ctx->stream, w.J, N, mu_r, mu_c, w.Dinv_r, w.Up_r, w.Dinv_c, w.Up_c);
        LAUNCH_OK(ctx);
        return 0;
    }
#endif
    return pcr_factor_launch(ctx, w, mu_r, mu_c, 1.0, 2);
}

// solve in place: w.rhs_r (and, with `both`, w.rhs_c), cell-major
int radau_solve(marl_ctx* ctx, RadauWork& w, bool both)
{
    const int64_t N = ctx->N;
#ifdef MARL_LAB_BLOCK_THOMAS
    if (!w.pcr) {
        hipLaunchKernelGGL(radau::solve_kernel, dim3(both ? 2 : 1), dim3(64), 0, ctx->stream, w.J, N, w.Dinv_r, w.Up_r, w.Dinv_c, w.Up_c, w.rhs_r, w.rhs_c,
                           both ? 3 : 1);
        LAUNCH_OK(ctx);
        return 0;
    }
#endif
    if (!w.cr_k) return pcr_solve_launch(ctx, w, N, both, w.rhs_r, w.rhs_c);
    if (w.plan.k && ctx->radau_fused_solve) {   // small grids: reduction levels, compact PCR and back-substitution in one launch (crpcr_solve_all)
        hipLaunchKernelGGL(radau::pcr_solve_fused_kernel, dim3(1, both ? 2 : 1), dim3(radau::PCR_FUSED_THREADS), 0, ctx->stream, N, w.nlevels, 0, w.Sr, w.Sc,
                           w.rhs_r, w.rhs_r, w.rhs_c, w.rhs_c, ZBatch{0, nullptr, 0, 0, nullptr}, w.plan, w.Cr, w.Cc);
        LAUNCH_OK(ctx);
        return 0;
    }
    // right-hand sides down the reduction levels, the compact system by PCR, solutions back up (in place from level to level); from the
    // first level of at most CR_TAIL_ROWS rows on, all remaining levels in one launch each way (option radau_cr_tail = 0: level by level)
    const unsigned gy = both ? 2 : 1;
    int tail0 = w.cr_k;
    if (ctx->radau_cr_tail) {
        while (tail0 > 0 && w.cr_n[tail0 - 1] <= radau::CR_TAIL_ROWS && w.cr_k - (tail0 - 1) <= radau::CR_TAIL_MAX_LEVELS) tail0--;
    }
    const int tailJ = w.cr_k - tail0;
    radau::CrTail tail{};
    tail.l0 = tail0;
    for (int j = 0; j <= tailJ; j++) { tail.n[j] = w.cr_n[tail0 + j]; tail.off[j] = w.cr_off[tail0 + j]; }
    for (int l = 0; l < tail0; l++) {
        const radau::CrShape sh{w.cr_n[l], w.cr_off[l], w.cr_n[l + 1], w.cr_off[l + 1]};
        hipLaunchKernelGGL(radau::cr_rhs_kernel, dim3(blocks256(NF * sh.n_next), gy), dim3(256), 0, ctx->stream, w.Cr, w.Cc, sh, l, w.rhs_r, w.rhs_c);
        LAUNCH_OK(ctx);
    }
#define MARL_TAIL(KERNEL, BLOCKS)                                                                                                               \
    switch (tailJ) {                                                                                                                            \
        case 1: hipLaunchKernelGGL(radau::KERNEL<1>, dim3(BLOCKS, gy), dim3(radau::CR_TAIL_THREADS), 0, ctx->stream, w.Cr, w.Cc, tail, w.rhs_r, w.rhs_c); break; \
        case 2: hipLaunchKernelGGL(radau::KERNEL<2>, dim3(BLOCKS, gy), dim3(radau::CR_TAIL_THREADS), 0, ctx->stream, w.Cr, w.Cc, tail, w.rhs_r, w.rhs_c); break; \
        case 3: hipLaunchKernelGGL(radau::KERNEL<3>, dim3(BLOCKS, gy), dim3(radau::CR_TAIL_THREADS), 0, ctx->stream, w.Cr, w.Cc, tail, w.rhs_r, w.rhs_c); break; \
        case 4: hipLaunchKernelGGL(radau::KERNEL<4>, dim3(BLOCKS, gy), dim3(radau::CR_TAIL_THREADS), 0, ctx->stream, w.Cr, w.Cc, tail, w.rhs_r, w.rhs_c); break; \
        case 5: hipLaunchKernelGGL(radau::KERNEL<5>, dim3(BLOCKS, gy), dim3(radau::CR_TAIL_THREADS), 0, ctx->stream, w.Cr, w.Cc, tail, w.rhs_r, w.rhs_c); break; \
        default: hipLaunchKernelGGL(radau::KERNEL<6>, dim3(BLOCKS, gy), dim3(radau::CR_TAIL_THREADS), 0, ctx->stream, w.Cr, w.Cc, tail, w.rhs_r, w.rhs_c); break; \
    }                                                                                                                                           \
    LAUNCH_OK(ctx);
    if (tailJ > 0) {
        const unsigned nb = (unsigned)((tail.n[tailJ] + radau::CR_TAIL_TOP - 1) / radau::CR_TAIL_TOP);
        MARL_TAIL(cr_rhs_tail_kernel, nb)
    }
    const int64_t offk = w.cr_off[w.cr_k];
    if (int rc = pcr_solve_launch(ctx, w, w.cr_n[w.cr_k], both, w.Cr.b + offk * NF, w.Cc.b + offk * NF)) return rc;
    if (tailJ > 0) {
        const int64_t rows = (int64_t)radau::CR_TAIL_TOP << tailJ;
        const unsigned nb = (unsigned)((tail.n[0] + rows - 1) / rows);
        MARL_TAIL(cr_back_tail_kernel, nb)
    }
#undef MARL_TAIL
    for (int l = tail0 - 1; l >= 0; l--) {
        const radau::CrShape sh{w.cr_n[l], w.cr_off[l], w.cr_n[l + 1], w.cr_off[l + 1]};
        hipLaunchKernelGGL(radau::cr_back_kernel, dim3((unsigned)((sh.n_cur + radau::CR_ROWS_PER_BLOCK - 1) / radau::CR_ROWS_PER_BLOCK), gy),
                           dim3(radau::CR_ROWS_PER_BLOCK * NF), 0, ctx->stream, w.Cr, w.Cc, sh, l, w.rhs_r, w.rhs_c);
        LAUNCH_OK(ctx);
    }
    return 0;
}

// J = finite-difference Jacobian at (y, f0)  (num_jac; njev is the caller's)
int radau_num_jac(marl_ctx* ctx, RadauWork& w, const double* y, const double* f0, double threshold)
{
    const int64_t N = ctx->N, n = NF * N;
    hipLaunchKernelGGL(radau::fd_prepare_kernel, dim3(blocks256(n)), dim3(256), 0, ctx->stream, y, f0, w.fac, threshold, w.have_factor ? 0 : 1, w.groups,
                       w.ng, n, w.h, w.yscale, w.YP);
    LAUNCH_OK(ctx);
    w.have_factor = true;
    if (int rc = launch_rhs(ctx, w.YP, w.FN, LAYOUT_FIELD_MAJOR, w.ng)) return rc;
    hipLaunchKernelGGL(radau::fd_columns_kernel, dim3(blocks256(n)), dim3(256), 0, ctx->stream, y, f0, w.FN, w.groups, w.ng, N, w.fac, w.yscale, w.Jraw,
                       w.maxdiff, w.scl, w.small, w.hnew, w.YP);
    LAUNCH_OK(ctx);
    // second trial step of the columns whose difference drowned in rounding (all groups in one launch; the columns that
    // were fine are perturbed by 0 - scipy evaluates only the groups that need it, with the same numbers)
    if (int rc = launch_rhs(ctx, w.YP, w.FN, LAYOUT_FIELD_MAJOR, w.ng)) return rc;
    hipLaunchKernelGGL(radau::fd_finish_kernel, dim3(blocks256(n)), dim3(256), 0, ctx->stream, f0, w.FN, w.groups, N, w.fac, w.h, w.maxdiff, w.scl, w.small,
                       w.hnew, w.Jraw, w.J);
    LAUNCH_OK(ctx);
    return 0;
}

// err = the solved right-hand side; sum (err / scale)^2 -> w.out[0]
int radau_error_norm(marl_ctx* ctx, RadauWork& w, double rtol, double atol)
{
    const int64_t N = ctx->N;
    const int nbk = radau_norm_blocks(NF * N);
    hipLaunchKernelGGL(radau::error_norm_kernel, dim3(nbk), dim3(1024), 0, ctx->stream, w.rhs_r, w.y, w.ynew, N, rtol, atol, w.err, w.yerr,
                       nbk > 1 ? w.partial : w.out);
    LAUNCH_OK(ctx);
    if (nbk > 1) {
        hipLaunchKernelGGL(radau::sum_partials_kernel, dim3(1), dim3(1), 0, ctx->stream, w.partial, nbk, w.out);
        LAUNCH_OK(ctx);
    }
    return 0;
}

// read back w.out[0] (a sum of squares) and the non-finite flag; resets the flag
int radau_read(marl_ctx* ctx, RadauWork& w, double* sumsq, int* flag)
{
    if (ctx->zc_on) {
        if (int rc = zc_wait(ctx, 0)) return rc;
        *sumsq = const_cast<const volatile double*>(ctx->zc_h)[0];
        volatile int32_t* fl = reinterpret_cast<volatile int32_t*>(ctx->zc_h + 1);
        if (flag) *flag = *fl;
        if (*fl) *fl = 0;     // (every kernel that could set it has completed: the word that was waited for is written after them)
        zc_arm(ctx, 0);       // for the next producer
        return 0;
    }
    HIP_OK(ctx, hipMemcpyAsync(ctx->rd_host, w.out, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipMemcpyAsync(ctx->rd_host + 1, w.flags, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    *sumsq = ctx->rd_host[0];
    int32_t fl;
    memcpy(&fl, ctx->rd_host + 1, sizeof fl);
    if (flag) *flag = fl;
    if (fl) HIP_OK(ctx, hipMemsetAsync(w.flags, 0, sizeof(int32_t), ctx->stream));
    return 0;
}

int radau_monitors(marl_ctx* ctx, const double* y, double g[7])
{
    if (ctx->zc_on) {
        zc_arm(ctx, 8, NQ);
        if (int rc = launch_monitors(ctx, y, LAYOUT_FIELD_MAJOR, ctx->zc_d + 8)) return rc;
        if (int rc = zc_wait(ctx, 8, NQ)) return rc;
        double r[NQ];
        for (int j = 0; j < NQ; j++) r[j] = const_cast<const volatile double*>(ctx->zc_h)[8 + j];
        record_to_events(r, g);
        return 0;
    }
    if (int rc = launch_monitors(ctx, y, LAYOUT_FIELD_MAJOR)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(ctx->hrec, ctx->rec, sizeof(double) * NQ, hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    record_to_events(ctx->hrec, g);
    return 0;
}

struct RadauDense { double t_old, h; };

// state of the dense output at time t -> out (device)
int radau_dense(marl_ctx* ctx, RadauWork& w, const RadauDense& d, double t, double* out)
{
    radau::X3 X = {{(t - d.t_old) / d.h, 0, 0}};
    const int64_t n = NF * ctx->N;
    hipLaunchKernelGGL(radau::dense_eval_kernel, dim3(blocks256(n)), dim3(256), 0, ctx->stream, w.Q, w.yold, (const double*)nullptr, n, X, 1, out);
    LAUNCH_OK(ctx);
    return 0;
}

// Brent's method for monitor e on [a, b] (solve_event_equation, ivp.py:51-76 -> scipy.optimize.brentq, xtol = rtol = 4 eps)
int radau_brent(marl_ctx* ctx, RadauWork& w, const RadauDense& d, int e, double a, double b, double* root)
{
    const double xtol = 4 * 2.220446049250313e-16, rtol = xtol;
    double g[7];
    auto at = [&](double t, double* v) -> int {
        if (int rc = radau_dense(ctx, w, d, t, w.tmp)) return rc;
        if (int rc = radau_monitors(ctx, w.tmp, g)) return rc;
        *v = g[e];
        return 0;
    };
    double fa, fb;
    if (int rc = at(a, &fa)) return rc;
    if (int rc = at(b, &fb)) return rc;
    if (fa == 0) { *root = a; return 0; }
    if (fb == 0) { *root = b; return 0; }
    double xpre = a, xcur = b, fpre = fa, fcur = fb, xblk = 0, fblk = 0, spre = 0, scur = 0;
    for (int it = 0; it < 100; it++) {
        if (fpre != 0 && fcur != 0 && ((fpre < 0) != (fcur < 0))) { xblk = xpre; fblk = fpre; spre = scur = xcur - xpre; }
        if (std::fabs(fblk) < std::fabs(fcur)) { xpre = xcur; xcur = xblk; xblk = xpre; fpre = fcur; fcur = fblk; fblk = fpre; }
        const double delta = (xtol + rtol * std::fabs(xcur)) / 2, sbis = (xblk - xcur) / 2;
        if (fcur == 0 || std::fabs(sbis) < delta) break;
        if (std::fabs(spre) > delta && std::fabs(fcur) < std::fabs(fpre)) {
            double stry;
            if (xpre == xblk) stry = -fcur * (xcur - xpre) / (fcur - fpre);
            else {
                const double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * std::fabs(stry) < std::fmin(std::fabs(spre), 3 * std::fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else { spre = sbis; scur = sbis; }
        xpre = xcur; fpre = fcur;
        if (std::fabs(scur) > delta) xcur += scur; else xcur += (sbis > 0 ? delta : -delta);
        if (int rc = at(xcur, &fcur)) return rc;
    }
    *root = xcur;
    return 0;
}

double radau_predict_factor(double h_abs, double h_abs_old, double error_norm, double error_norm_old)   // radau.py:133-173; < 0 = None
{
    double multiplier;
    if (error_norm_old < 0 || h_abs_old < 0 || error_norm == 0) multiplier = 1;
    else multiplier = h_abs / h_abs_old * std::pow(error_norm_old / error_norm, 0.25);
    return (multiplier < 1 ? multiplier : 1) * std::pow(error_norm, -0.25);
}

// y0 in w.y (device, field-major).  y_eval_host: n_eval x 5N (host).  On return w.y holds the state at stats->t.
int radau_run(marl_ctx* ctx, RadauWork& w, double t0, double t1, double first_step, double rtol, double atol, const double* t_eval,
              int64_t n_eval, double* y_eval_host, double* t_events, int64_t max_events, int64_t max_attempts, marl_stats* st)
{
    const int64_t N = ctx->N, n = NF * N;
    const double S6 = std::sqrt(6.0);
    const double C3[3] = {(4 - S6) / 10, (4 + S6) / 10, 1};
    const double E3[3] = {(-13 - 7 * S6) / 3, (-13 + 7 * S6) / 3, -1.0 / 3};
    const double MU_REAL = 3 + std::pow(3, 2.0 / 3) - std::pow(3, 1.0 / 3);
    const cplx MU_COMPLEX = {3 + 0.5 * (std::pow(3, 1.0 / 3) - std::pow(3, 2.0 / 3)), -0.5 * (std::pow(3, 5.0 / 6) + std::pow(3, 7.0 / 6))};
    const radau::P33 P = {{{13.0 / 3 + 7 * S6 / 3, -23.0 / 3 - 22 * S6 / 3, 10.0 / 3 + 5 * S6},
                           {13.0 / 3 - 7 * S6 / 3, -23.0 / 3 + 22 * S6 / 3, 10.0 / 3 - 5 * S6},
                           {1.0 / 3, -8.0 / 3, 10.0 / 3}}};
    constexpr int NEWTON_MAXITER = 6;
    constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10;
    memset(st, 0, sizeof *st);
    rtol = clamp_rtol(rtol);
    const dim3 gn(blocks256(n)), b256(256);

    // Radau.__init__ (radau.py:290-343)
    double t = t0;
    if (int rc = launch_rhs(ctx, w.y, w.f, LAYOUT_FIELD_MAJOR)) return rc;
    st->nfev = 1;
    double S_h_abs = first_step, S_h_abs_old = -1, S_err_old = -1;
    const double newton_tol = std::fmax(10 * radau::EPS / rtol, std::fmin(0.03, std::sqrt(rtol)));
    if (!ctx->zc_on) HIP_OK(ctx, hipMemsetAsync(w.flags, 0, sizeof(int32_t) * 2, ctx->stream));
    if (int rc = radau_num_jac(ctx, w, w.y, w.f, atol)) return rc;
    st->njev = 1;
    bool current_jac = true, have_lu = false, have_sol = false;
    RadauDense dense = {t0, 0};
    double g[7], g_new[7];
    if (int rc = radau_monitors(ctx, w.y, g)) return rc;   // ivp.py:645
    int64_t eval_i = 0, attempts = 0;
    int status = 1;

    while (status == 1) {
        if (t == t1) { status = 0; break; }
        const double min_step = 10 * std::fabs(std::nextafter(t, INFINITY) - t);
        double h_abs = S_h_abs, h_abs_o = S_h_abs_old, err_o = S_err_old;
        if (S_h_abs < min_step) { h_abs = min_step; h_abs_o = -1; err_o = -1; }
        bool rejected = false, accepted = false;
        int n_iter = 0;
        double rate = -1, h = 0, t_new = t, error_norm = 0, safety = 0;
        while (!accepted) {
            if (h_abs < min_step) { status = -1; break; }
            if (max_attempts > 0 && attempts >= max_attempts) { status = 2; break; }
            attempts++;
            h = h_abs;
            t_new = t + h;
            if (t_new - t1 > 0) t_new = t1;
            h = t_new - t;
            h_abs = std::fabs(h);
            if (!have_sol) {
                HIP_OK(ctx, hipMemsetAsync(w.Z0, 0, sizeof(double) * 3 * n, ctx->stream));
            } else {   // Z0 = self.sol(t + h * C).T - y
                radau::X3 X;
                for (int s = 0; s < 3; s++) X.x[s] = ((t + h * C3[s]) - dense.t_old) / dense.h;
                hipLaunchKernelGGL(radau::dense_eval_kernel, gn, b256, 0, ctx->stream, w.Q, w.yold, (const double*)w.y, n, X, 3, w.Z0);
                LAUNCH_OK(ctx);
            }
            bool converged = false;
            while (!converged) {
                if (!have_lu) {
                    const cplx muc = {MU_COMPLEX.re / h, MU_COMPLEX.im / h};
                    if (int rc = radau_factor(ctx, w, MU_REAL / h, muc)) return rc;
                    st->nlu += 2;
                    have_lu = true;
                }
                // ---- solve_collocation_system ----
                const double M_real = MU_REAL / h;
                const cplx M_c = {MU_COMPLEX.re / h, MU_COMPLEX.im / h};
                hipLaunchKernelGGL(radau::newton_begin_kernel, gn, b256, 0, ctx->stream, w.y, w.Z0, n, rtol, atol, w.scale, w.Z, w.W, w.YS);
                LAUNCH_OK(ctx);
                double dW_norm_old = -1;
                rate = -1;
                int k;
                // small grids, radau_fused_solve = 3 (default there): two launches per iteration - the solve workgroups assemble their own
                // right-hand sides, the update workgroup evaluates the next iteration's stage derivatives (marl_radau_wg.h)
                const bool two_launch = w.pcr && (!w.cr_k || w.plan.k) && n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve == 3 && ctx->zc_on;
                for (k = 0; k < NEWTON_MAXITER; k++) {
                    if (!two_launch || k == 0)
                        if (int rc = launch_rhs(ctx, w.YS, w.F, LAYOUT_FIELD_MAJOR, 3)) return rc;
                    st->nfev += 3;
                    if (two_launch) {
                        hipLaunchKernelGGL(radau::newton_solve2_kernel, dim3(1, 2), dim3(radau::PCR_FUSED_THREADS), 0, ctx->stream, w.F, w.W, N, M_real, M_c, w.nlevels, w.Sr,
                                           w.Sc, w.rhs_r, w.rhs_c, w.flags, w.plan, w.Cr, w.Cc);
                        LAUNCH_OK(ctx);
                        if (ctx->var_dphi)
                            hipLaunchKernelGGL(radau::newton_update_rhs_kernel<true>, dim3(1), dim3(radau::WG_THREADS), 0, ctx->stream, w.y, w.rhs_r, w.rhs_c, w.scale, N, w.W,
                                               w.Z, w.YS, w.F, ctx->dconsts, w.out);
                        else
                            hipLaunchKernelGGL(radau::newton_update_rhs_kernel<false>, dim3(1), dim3(radau::WG_THREADS), 0, ctx->stream, w.y, w.rhs_r, w.rhs_c, w.scale, N, w.W,
                                               w.Z, w.YS, w.F, ctx->dconsts, w.out);
                        LAUNCH_OK(ctx);
                    } else
                    if (w.pcr && (!w.cr_k || w.plan.k) && n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve == 2) {   // right-hand sides + both solves + update + norm: one launch
                        hipLaunchKernelGGL(radau::newton_fused_kernel, dim3(1), dim3(radau::PCR_FUSED_THREADS), 0, ctx->stream, w.y, w.F, N, M_real, M_c, w.nlevels, w.Sr,
                                           w.Sc, w.scale, w.W, w.Z, w.YS, w.rhs_r, w.rhs_c, w.flags, w.out, w.plan, w.Cr, w.Cc);
                        LAUNCH_OK(ctx);
                    } else {
                    hipLaunchKernelGGL(radau::newton_rhs_kernel, gn, b256, 0, ctx->stream, w.F, w.W, N, M_real, M_c, w.rhs_r, w.rhs_c, w.flags);
                    LAUNCH_OK(ctx);
                    if (int rc = radau_solve(ctx, w, true)) return rc;
                    {
                        const int nbk = radau_norm_blocks(n);
                        hipLaunchKernelGGL(radau::newton_update_kernel, dim3(nbk), dim3(1024), 0, ctx->stream, w.y, w.rhs_r, w.rhs_c, w.scale, N, w.W, w.Z, w.YS,
                                           nbk > 1 ? w.partial : w.out);
                        LAUNCH_OK(ctx);
                        if (nbk > 1) {
                            hipLaunchKernelGGL(radau::sum_partials_kernel, dim3(1), dim3(1), 0, ctx->stream, w.partial, nbk, w.out);
                            LAUNCH_OK(ctx);
                        }
                    }
                    }
                    double ss;
                    int nonfinite;
                    if (int rc = radau_read(ctx, w, &ss, &nonfinite)) return rc;
                    if (nonfinite) break;   // `if not np.all(np.isfinite(F)): break` (the solve above is then unused)
                    const double dW_norm = std::sqrt(ss) / std::sqrt((double)(3 * n));
                    if (dW_norm_old >= 0) rate = dW_norm / dW_norm_old;
                    if (rate >= 0 && (rate >= 1 || std::pow(rate, NEWTON_MAXITER - k) / (1 - rate) * dW_norm > newton_tol)) break;
                    // (W += dW, Z = T W were applied by newton_update_kernel)
                    if (dW_norm == 0 || (rate >= 0 && rate / (1 - rate) * dW_norm < newton_tol)) { converged = true; break; }
                    dW_norm_old = dW_norm;
                }
                n_iter = (k < NEWTON_MAXITER ? k : NEWTON_MAXITER - 1) + 1;
                if (!converged) {
                    if (current_jac) break;
                    if (int rc = radau_num_jac(ctx, w, w.y, w.f, atol)) return rc;
                    st->njev++;
                    current_jac = true;
                    have_lu = false;
                }
            }
            if (!converged) {
                h_abs *= 0.5;
                have_lu = false;
                st->n_rejected++;
                continue;
            }
            // error estimate (radau.py:466-478); small grids: one launch (error_fused_kernel)
            const bool err_fused = w.pcr && (!w.cr_k || w.plan.k) && n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve == 3;
            auto error_estimate = [&](const double* fvec) -> int {
                if (err_fused) {
                    hipLaunchKernelGGL(radau::error_fused_kernel, dim3(1), dim3(radau::PCR_FUSED_THREADS), 0, ctx->stream, fvec, w.Z, w.y, N, E3[0], E3[1], E3[2], h, w.nlevels,
                                       w.Sr, rtol, atol, w.ynew, w.err, w.yerr, w.out, w.plan, w.Cr);
                    LAUNCH_OK(ctx);
                    return 0;
                }
                hipLaunchKernelGGL(radau::error_rhs_kernel, gn, b256, 0, ctx->stream, fvec, w.Z, w.y, N, E3[0], E3[1], E3[2], h, w.rhs_r, w.ynew);
                LAUNCH_OK(ctx);
                if (int rc = radau_solve(ctx, w, false)) return rc;
                return radau_error_norm(ctx, w, rtol, atol);
            };
            if (int rc = error_estimate(w.f)) return rc;
            double ss;
            if (int rc = radau_read(ctx, w, &ss, nullptr)) return rc;
            error_norm = std::sqrt(ss) / std::sqrt((double)n);
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
            if (rejected && error_norm > 1) {
                if (int rc = launch_rhs(ctx, w.yerr, w.tmp, LAYOUT_FIELD_MAJOR)) return rc;   // fun(t, y + error)
                st->nfev++;
                if (int rc = error_estimate(w.tmp)) return rc;
                if (int rc = radau_read(ctx, w, &ss, nullptr)) return rc;
                error_norm = std::sqrt(ss) / std::sqrt((double)n);
            }
            if (error_norm > 1) {
                const double sf = safety * radau_predict_factor(h_abs, h_abs_o, error_norm, err_o);
                h_abs *= (sf > MIN_FACTOR) ? sf : MIN_FACTOR;
                have_lu = false;
                rejected = true;
                st->n_rejected++;
            } else {
                accepted = true;
            }
        }
        if (status != 1) break;
        const bool recompute_jac = n_iter > 2 && rate > 1e-3;
        double factor = radau_predict_factor(h_abs, h_abs_o, error_norm, err_o);
        { const double sf = safety * factor; factor = (sf < MAX_FACTOR) ? sf : MAX_FACTOR; }
        if (!recompute_jac && factor < 1.2) factor = 1;
        else have_lu = false;
        // small grids: f(y_new), the dense-output coefficients and the monitors of y_new in one launch (accept_fused_kernel)
        const bool acc_fused = w.pcr && n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve == 3 && ctx->zc_on;
        if (acc_fused) {
            if (ctx->var_dphi)
                hipLaunchKernelGGL(radau::accept_fused_kernel<true>, dim3(1), dim3(radau::WG_THREADS), 0, ctx->stream, w.ynew, w.fnew, w.Z, N, P, w.Q, ctx->dconsts, ctx->zc_d);
            else
                hipLaunchKernelGGL(radau::accept_fused_kernel<false>, dim3(1), dim3(radau::WG_THREADS), 0, ctx->stream, w.ynew, w.fnew, w.Z, N, P, w.Q, ctx->dconsts, ctx->zc_d);
            LAUNCH_OK(ctx);
        } else if (int rc = launch_rhs(ctx, w.ynew, w.fnew, LAYOUT_FIELD_MAJOR)) return rc;
        st->nfev++;
        if (recompute_jac) {
            if (int rc = radau_num_jac(ctx, w, w.ynew, w.fnew, atol)) return rc;
            st->njev++;
            current_jac = true;
        } else {
            current_jac = false;
        }
        S_h_abs_old = S_h_abs;      // radau.py:512: the value self.h_abs had when this step started
        S_err_old = error_norm;
        S_h_abs = h_abs * factor;
        st->n_accepted++;
        std::swap(w.yold, w.y);      // y_old = y
        std::swap(w.y, w.ynew);      // y = y_new   (w.ynew now holds the state before last: scratch)
        std::swap(w.f, w.fnew);
        const double t_old = t;
        t = t_new;
        if (!acc_fused) {
            hipLaunchKernelGGL(radau::dense_q_kernel, gn, b256, 0, ctx->stream, w.Z, n, P, w.Q);
            LAUNCH_OK(ctx);
        }
        have_sol = true;
        dense = {t_old, t - t_old};
        if (t - t1 >= 0) status = 0;

        // events (ivp.py:673-694) and t_eval (ivp.py:706-723)
        if (acc_fused) {
            double marker;
            if (int rc = radau_read(ctx, w, &marker, nullptr)) return rc;
            for (int e = 0; e < 7; e++) g_new[e] = const_cast<const volatile double*>(ctx->zc_h)[16 + e];
        } else if (int rc = radau_monitors(ctx, w.y, g_new)) return rc;
        for (int e = 0; e < 7; e++) {
            const bool up = g[e] <= 0 && g_new[e] >= 0, down = g[e] >= 0 && g_new[e] <= 0;
            if (up || down) {
                if (t_events && st->n_events[e] < max_events) {
                    double root;
                    if (int rc = radau_brent(ctx, w, dense, e, t_old, t, &root)) return rc;
                    t_events[e * max_events + st->n_events[e]] = root;
                }
                st->n_events[e]++;
            }
            g[e] = g_new[e];
        }
        while (t_eval && eval_i < n_eval && t_eval[eval_i] <= t) {
            if (int rc = radau_dense(ctx, w, dense, t_eval[eval_i], w.tmp)) return rc;
            HIP_OK(ctx, hipMemcpyAsync(y_eval_host + eval_i * n, w.tmp, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
            eval_i++;
        }
    }
    st->status = status;
    st->t = t;
    st->h_next = S_h_abs;
    if (int rc = radau_monitors(ctx, w.y, st->event_value)) return rc;
    return 0;
}

// ---- BDF (scipy/integrate/_ivp/bdf.py), the other implicit method the reference's Solver names --------------------------------------
// compute_R / change_D (bdf.py:18-33): D[:order + 1] = (R U)^T D[:order + 1]
void bdf_compute_R(int order, double factor, double R[6][6])
{
    for (int j = 0; j <= order; j++) R[0][j] = 1;
    for (int i = 1; i <= order; i++) {
        R[i][0] = 0;
        for (int j = 1; j <= order; j++) R[i][j] = R[i - 1][j] * ((i - 1 - factor * j) / i);
    }
}

int bdf_change_D(marl_ctx* ctx, double* D, int64_t n, int order, double factor)
{
    double R[6][6], U[6][6];
    bdf::Mat6 RU = {};
    bdf_compute_R(order, factor, R);
    bdf_compute_R(order, 1.0, U);
    for (int i = 0; i <= order; i++)
        for (int j = 0; j <= order; j++) {
            double a = 0;
            for (int k = 0; k <= order; k++) a += R[i][k] * U[k][j];
            RU.m[i][j] = a;
        }
    hipLaunchKernelGGL(bdf::change_D_kernel, dim3(blocks256(n)), dim3(256), 0, ctx->stream, D, n, order, RU);
    LAUNCH_OK(ctx);
    return 0;
}

// sum (coef v / (atol + rtol |yref|))^2 -> host
int bdf_scaled_sumsq(marl_ctx* ctx, RadauWork& w, const double* v, double coef, const double* yref, double rtol, double atol, double* ss)
{
    const int64_t n = NF * ctx->N;
    const int nbk = radau_norm_blocks(n);
    hipLaunchKernelGGL(bdf::scaled_norm_kernel, dim3(nbk), dim3(1024), 0, ctx->stream, v, coef, yref, rtol, atol, n, nbk > 1 ? w.partial : w.out);
    LAUNCH_OK(ctx);
    if (nbk > 1) {
        hipLaunchKernelGGL(radau::sum_partials_kernel, dim3(1), dim3(1), 0, ctx->stream, w.partial, nbk, w.out);
        LAUNCH_OK(ctx);
    }
    return radau_read(ctx, w, ss, nullptr);
}

struct BdfDense { double t, h; int order; const double* D; };

int bdf_dense(marl_ctx* ctx, const BdfDense& d, double tq, double* out)
{
    bdf::Vec6 p = {};
    double acc = 1;
    for (int j = 0; j < d.order; j++) {   // x_j = (t - (t_end - h j)) / (h (1 + j)); p = cumprod(x)
        acc *= (tq - (d.t - d.h * j)) / (d.h * (1 + j));
        p.v[j] = acc;
    }
    const int64_t n = NF * ctx->N;
    hipLaunchKernelGGL(bdf::dense_kernel, dim3(blocks256(n)), dim3(256), 0, ctx->stream, d.D, n, d.order, p, out);
    LAUNCH_OK(ctx);
    return 0;
}

// Brent on monitor e over the BDF dense output (as radau_brent)
int bdf_brent(marl_ctx* ctx, RadauWork& w, const BdfDense& d, int e, double a, double b, double* root)
{
    const double xtol = 4 * 2.220446049250313e-16, rtol = xtol;
    double g[7];
    auto at = [&](double t, double* v) -> int {
        if (int rc = bdf_dense(ctx, d, t, w.tmp)) return rc;
        if (int rc = radau_monitors(ctx, w.tmp, g)) return rc;
        *v = g[e];
        return 0;
    };
    double fa, fb;
    if (int rc = at(a, &fa)) return rc;
    if (int rc = at(b, &fb)) return rc;
    if (fa == 0) { *root = a; return 0; }
    if (fb == 0) { *root = b; return 0; }
    double xpre = a, xcur = b, fpre = fa, fcur = fb, xblk = 0, fblk = 0, spre = 0, scur = 0;
    for (int it = 0; it < 100; it++) {
        if (fpre != 0 && fcur != 0 && ((fpre < 0) != (fcur < 0))) { xblk = xpre; fblk = fpre; spre = scur = xcur - xpre; }
        if (std::fabs(fblk) < std::fabs(fcur)) { xpre = xcur; xcur = xblk; xblk = xpre; fpre = fcur; fcur = fblk; fblk = fpre; }
        const double delta = (xtol + rtol * std::fabs(xcur)) / 2, sbis = (xblk - xcur) / 2;
        if (fcur == 0 || std::fabs(sbis) < delta) break;
        if (std::fabs(spre) > delta && std::fabs(fcur) < std::fabs(fpre)) {
            double stry;
            if (xpre == xblk) stry = -fcur * (xcur - xpre) / (fcur - fpre);
            else {
                const double dpre = (fpre - fcur) / (xpre - xcur), dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk * dblk - fpre * dpre) / (dblk * dpre * (fblk - fpre));
            }
            if (2 * std::fabs(stry) < std::fmin(std::fabs(spre), 3 * std::fabs(sbis) - delta)) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else { spre = sbis; scur = sbis; }
        xpre = xcur; fpre = fcur;
        if (std::fabs(scur) > delta) xcur += scur; else xcur += (sbis > 0 ? delta : -delta);
        if (int rc = at(xcur, &fcur)) return rc;
    }
    *root = xcur;
    return 0;
}

// BDF.__init__ + the solve_ivp loop (ivp.py:654-723) around BDF._step_impl (bdf.py:310-450).  Buffers of the Radau arena are reused:
// D = [Z | W | F] (8 n of 9 n), f = Z0, y_predict / psi / d = Q.
int bdf_run(marl_ctx* ctx, RadauWork& w, double t0, double t1, double first_step, double rtol, double atol, const double* t_eval,
            int64_t n_eval, double* y_eval_host, double* t_events, int64_t max_events, int64_t max_attempts, marl_stats* st)
{
    const int64_t N = ctx->N, n = NF * N;
    constexpr int NEWTON_MAXITER = 4, MAX_ORDER = 5;
    constexpr double MIN_FACTOR = 0.2, MAX_FACTOR = 10;
    memset(st, 0, sizeof *st);
    rtol = clamp_rtol(rtol);
    const dim3 gn(blocks256(n)), b256(256);
    const double kappa[6] = {0, -0.1850, -1.0 / 9, -0.0823, -0.0415, 0};
    bdf::Vec6 gamma = {};
    double alpha[6], error_const[6];
    for (int k = 1; k <= MAX_ORDER; k++) gamma.v[k] = gamma.v[k - 1] + 1.0 / k;
    for (int k = 0; k <= MAX_ORDER; k++) { alpha[k] = (1 - kappa[k]) * gamma.v[k]; error_const[k] = kappa[k] * gamma.v[k] + 1.0 / (k + 1); }
    double* D = w.Z;            // [MAX_ORDER + 3][n]
    double* f = w.Z0;           // f(t_new, y) of the Newton iteration
    double* ypred = w.Q, *psi = w.Q + n, *d = w.Q + 2 * n;
    w.pcr = true;               // (the sequential block-Thomas option serves the Radau path only)

    double t = t0;
    if (int rc = launch_rhs(ctx, w.y, w.f, LAYOUT_FIELD_MAJOR)) return rc;
    st->nfev = 1;
    double S_h_abs = first_step;
    const double newton_tol = std::fmax(10 * radau::EPS / rtol, std::fmin(0.03, std::sqrt(rtol)));
    if (!ctx->zc_on) HIP_OK(ctx, hipMemsetAsync(w.flags, 0, sizeof(int32_t) * 2, ctx->stream));
    // jac_wrapped(t0, y0): f = fun_single(t, y) (not counted; the same values as w.f), J by finite differences
    if (int rc = radau_num_jac(ctx, w, w.y, w.f, atol)) return rc;
    st->njev = 1;
    hipLaunchKernelGGL(bdf::init_D_kernel, gn, b256, 0, ctx->stream, w.y, w.f, S_h_abs, n, D);
    LAUNCH_OK(ctx);
    int order = 1, n_equal_steps = 0;
    bool have_lu = false;
    double g[7], g_new[7], g_spec[7];
    bool have_g_spec = false;   // g_spec: the monitors of the state the last converged solve left in w.ynew (solve_wg)
    const bool solve_wg = ctx->bdf_solve_wg && ctx->zc_on && (!w.cr_k || w.plan.k) && n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve;
    if (int rc = radau_monitors(ctx, w.y, g)) return rc;
    int64_t eval_i = 0, attempts = 0;
    int status = 1;

    while (status == 1) {
        if (t == t1) { status = 0; break; }
        const double min_step = 10 * std::fabs(std::nextafter(t, INFINITY) - t);
        double h_abs;
        if (S_h_abs < min_step) {
            h_abs = min_step;
            if (int rc = bdf_change_D(ctx, D, n, order, min_step / S_h_abs)) return rc;
            n_equal_steps = 0;
        } else {
            h_abs = S_h_abs;
        }
        bool current_jac = false, accepted = false;
        int n_iter = 0;
        double h = 0, t_new = t, error_norm = 0, safety = 0;
        while (!accepted) {
            if (h_abs < min_step) { status = -1; break; }
            if (max_attempts > 0 && attempts >= max_attempts) { status = 2; break; }
            attempts++;
            h = h_abs;
            t_new = t + h;
            if (t_new - t1 > 0) {
                t_new = t1;
                if (int rc = bdf_change_D(ctx, D, n, order, std::fabs(t_new - t) / h_abs)) return rc;
                n_equal_steps = 0;
                have_lu = false;
            }
            h = t_new - t;
            h_abs = std::fabs(h);
            if (!solve_wg) {
                hipLaunchKernelGGL(bdf::predict_kernel, gn, b256, 0, ctx->stream, D, n, order, gamma, alpha[order], rtol, atol, ypred, w.scale, psi, w.ynew, d);
                LAUNCH_OK(ctx);
            }
            bool converged = false;
            bool fused_err = false;   // the converged iteration's launch also left the local error sum (small systems, zero-copy words)
            const double c = h / alpha[order];
            bool first_pass = true;
            while (!converged) {
                if (!have_lu) {   // LU = self.lu(self.I - c * J)
                    if (int rc = pcr_factor_launch(ctx, w, 1.0, cplx{0, 0}, c, 1)) return rc;
                    st->nlu++;
                    have_lu = true;
                }
                if (!first_pass && !solve_wg) {
                    hipLaunchKernelGGL(bdf::newton_restart_kernel, gn, b256, 0, ctx->stream, ypred, n, w.ynew, d);
                    LAUNCH_OK(ctx);
                }
                const int wg_mode = first_pass ? 0 : 1;
                first_pass = false;
                // ---- solve_bdf_system (bdf.py:36-68) ----
                double dy_norm_old = -1;
                int k;
                if (solve_wg) {   // predictor / restart, every Newton iteration with its tests, error sum and monitors: one launch, one wait
                    if (ctx->var_dphi)
                        hipLaunchKernelGGL(bdf::solve_wg_kernel<true>, dim3(1), dim3(radau::WG_THREADS), 0, ctx->stream, wg_mode, D, order, gamma, alpha[order], ypred,
                                           psi, w.scale, w.ynew, d, f, N, c, w.nlevels, w.Sr, ctx->dconsts, newton_tol, error_const[order], rtol, atol, ctx->zc_d, w.plan, w.Cr);
                    else
                        hipLaunchKernelGGL(bdf::solve_wg_kernel<false>, dim3(1), dim3(radau::WG_THREADS), 0, ctx->stream, wg_mode, D, order, gamma, alpha[order], ypred,
                                           psi, w.scale, w.ynew, d, f, N, c, w.nlevels, w.Sr, ctx->dconsts, newton_tol, error_const[order], rtol, atol, ctx->zc_d, w.plan, w.Cr);
                    LAUNCH_OK(ctx);
                    double ss;
                    int nonfinite;
                    if (int rc = radau_read(ctx, w, &ss, &nonfinite)) return rc;
                    const volatile double* zw = const_cast<const volatile double*>(ctx->zc_h);
                    const int iters = (int)zw[bdf::SW_ITERS];
                    st->nfev += iters;
                    converged = zw[bdf::SW_CONVERGED] != 0;
                    fused_err = converged;
                    if (converged) {
                        for (int e = 0; e < 7; e++) g_spec[e] = zw[bdf::SW_G + e];
                        have_g_spec = true;
                    }
                    k = iters - 1;
                } else
                for (k = 0; k < NEWTON_MAXITER; k++) {
                    if (int rc = launch_rhs(ctx, w.ynew, f, LAYOUT_FIELD_MAJOR)) return rc;
                    st->nfev++;
                    if ((!w.cr_k || w.plan.k) && n <= radau::PCR_FUSED_MAX && ctx->radau_fused_solve) {   // right-hand side + every level + update + norm in one launch
                        // (with the zero-copy result words: the step's local error norm rides along - bdf.py:398-400 - one launch and
                        //  one wait less per step)
                        fused_err = ctx->zc_on;
                        hipLaunchKernelGGL(bdf::newton_fused_kernel, dim3(1), dim3(radau::PCR_FUSED_THREADS), 0, ctx->stream, f, psi, N, c, w.nlevels, w.Sr, w.scale,
                                           w.ynew, d, w.flags, w.out, error_const[order], rtol, atol, fused_err ? ctx->zc_d + 2 : (double*)nullptr, w.plan, w.Cr);
                        LAUNCH_OK(ctx);
                    } else {
                    hipLaunchKernelGGL(bdf::newton_rhs_kernel, gn, b256, 0, ctx->stream, f, psi, d, N, c, w.rhs_r, w.flags);
                    LAUNCH_OK(ctx);
                    if (int rc = radau_solve(ctx, w, false)) return rc;
                    {
                        const int nbk = radau_norm_blocks(n);
                        hipLaunchKernelGGL(bdf::newton_update_kernel, dim3(nbk), dim3(1024), 0, ctx->stream, w.rhs_r, w.scale, N, w.ynew, d, nbk > 1 ? w.partial : w.out);
                        LAUNCH_OK(ctx);
                        if (nbk > 1) {
                            hipLaunchKernelGGL(radau::sum_partials_kernel, dim3(1), dim3(1), 0, ctx->stream, w.partial, nbk, w.out);
                            LAUNCH_OK(ctx);
                        }
                    }
                    }
                    double ss;
                    int nonfinite;
                    if (int rc = radau_read(ctx, w, &ss, &nonfinite)) return rc;
                    if (nonfinite) break;
                    const double dy_norm = std::sqrt(ss) / std::sqrt((double)n);
                    double rate = -1;
                    if (dy_norm_old >= 0) rate = dy_norm / dy_norm_old;
                    if (rate >= 0 && (rate >= 1 || std::pow(rate, NEWTON_MAXITER - k) / (1 - rate) * dy_norm > newton_tol)) break;
                    if (dy_norm == 0 || (rate >= 0 && rate / (1 - rate) * dy_norm < newton_tol)) { converged = true; break; }
                    dy_norm_old = dy_norm;
                }
                n_iter = (k < NEWTON_MAXITER ? k : NEWTON_MAXITER - 1) + 1;
                if (!converged) {
                    if (current_jac) break;
                    // J = self.jac(t_new, y_predict): f = fun_single(t_new, y_predict) - not counted
                    if (int rc = launch_rhs(ctx, ypred, w.fnew, LAYOUT_FIELD_MAJOR)) return rc;
                    if (int rc = radau_num_jac(ctx, w, ypred, w.fnew, atol)) return rc;
                    st->njev++;
                    have_lu = false;
                    current_jac = true;
                }
            }
            if (!converged) {
                h_abs *= 0.5;
                if (int rc = bdf_change_D(ctx, D, n, order, 0.5)) return rc;
                n_equal_steps = 0;
                have_lu = false;
                st->n_rejected++;
                continue;
            }
            safety = 0.9 * (2 * NEWTON_MAXITER + 1) / (2 * NEWTON_MAXITER + n_iter);
            double ss;
            if (fused_err) ss = const_cast<const volatile double*>(ctx->zc_h)[2];   // written before the word radau_read waited for
            else if (int rc = bdf_scaled_sumsq(ctx, w, d, error_const[order], w.ynew, rtol, atol, &ss)) return rc;
            error_norm = std::sqrt(ss) / std::sqrt((double)n);
            if (error_norm > 1) {
                const double sf = safety * std::pow(error_norm, -1.0 / (order + 1));
                const double factor = (sf > MIN_FACTOR) ? sf : MIN_FACTOR;
                h_abs *= factor;
                if (int rc = bdf_change_D(ctx, D, n, order, factor)) return rc;
                n_equal_steps = 0;
                st->n_rejected++;   // (bdf.py:405-406: the LU is kept)
            } else {
                accepted = true;
            }
        }
        if (status != 1) break;
        n_equal_steps++;
        const double t_old = t;
        t = t_new;
        S_h_abs = h_abs;
        st->n_accepted++;
        hipLaunchKernelGGL(bdf::accept_kernel, gn, b256, 0, ctx->stream, D, d, w.ynew, n, order, w.y);
        LAUNCH_OK(ctx);
        if (n_equal_steps >= order + 1) {   // order / step-size selection (bdf.py:427-448); scale = atol + rtol |y_new| = |y| now
            double em = INFINITY, ep = INFINITY, ss;
            if (order > 1 && order < MAX_ORDER && ctx->zc_on && radau_norm_blocks(n) == 1) {   // both norms: one launch, one wait
                hipLaunchKernelGGL(bdf::scaled_norm2_kernel, dim3(1), dim3(1024), 0, ctx->stream, D + (int64_t)order * n, error_const[order - 1],
                                   D + (int64_t)(order + 2) * n, error_const[order + 1], w.y, rtol, atol, n, w.out, ctx->zc_d + 3);
                LAUNCH_OK(ctx);
                if (int rc = radau_read(ctx, w, &ss, nullptr)) return rc;
                em = std::sqrt(ss) / std::sqrt((double)n);
                ep = std::sqrt(const_cast<const volatile double*>(ctx->zc_h)[3]) / std::sqrt((double)n);
            } else {
            if (order > 1) {
                if (int rc = bdf_scaled_sumsq(ctx, w, D + (int64_t)order * n, error_const[order - 1], w.y, rtol, atol, &ss)) return rc;
                em = std::sqrt(ss) / std::sqrt((double)n);
            }
            if (order < MAX_ORDER) {
                if (int rc = bdf_scaled_sumsq(ctx, w, D + (int64_t)(order + 2) * n, error_const[order + 1], w.y, rtol, atol, &ss)) return rc;
                ep = std::sqrt(ss) / std::sqrt((double)n);
            }
            }
            const double en[3] = {em, error_norm, ep};
            double factors[3];
            int best = 0;
            for (int i = 0; i < 3; i++) {
                factors[i] = std::pow(en[i], -1.0 / (order + i));
                if (factors[i] > factors[best]) best = i;
            }
            for (int i = 0; i < 3; i++)
                if (factors[i] != factors[i]) { best = i; break; }   // np.argmax: the first NaN
            order += best - 1;
            const double sf = safety * factors[best];
            const double factor = (sf < MAX_FACTOR) ? sf : MAX_FACTOR;   // python's min(MAX_FACTOR, x): a NaN x yields MAX_FACTOR (bdf.py:442)
            S_h_abs *= factor;
            if (int rc = bdf_change_D(ctx, D, n, order, factor)) return rc;
            n_equal_steps = 0;
            have_lu = false;
        }
        if (t - t1 >= 0) status = 0;

        // events and t_eval on the dense output built AFTER the order / step-size update (bdf.py:452-454)
        const BdfDense dense = {t, S_h_abs, order, D};
        if (solve_wg && have_g_spec) {   // y = the state of the last converged solve: its monitors came with the solve
            for (int e = 0; e < 7; e++) g_new[e] = g_spec[e];
        } else if (int rc = radau_monitors(ctx, w.y, g_new)) return rc;
        have_g_spec = false;
        for (int e = 0; e < 7; e++) {
            const bool up = g[e] <= 0 && g_new[e] >= 0, down = g[e] >= 0 && g_new[e] <= 0;
            if (up || down) {
                if (t_events && st->n_events[e] < max_events) {
                    double root;
                    if (int rc = bdf_brent(ctx, w, dense, e, t_old, t, &root)) return rc;
                    t_events[e * max_events + st->n_events[e]] = root;
                }
                st->n_events[e]++;
            }
            g[e] = g_new[e];
        }
        while (t_eval && eval_i < n_eval && t_eval[eval_i] <= t) {
            if (int rc = bdf_dense(ctx, dense, t_eval[eval_i], w.tmp)) return rc;
            HIP_OK(ctx, hipMemcpyAsync(y_eval_host + eval_i * n, w.tmp, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
            HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
            eval_i++;
        }
    }
    st->status = status;
    st->t = t;
    st->h_next = S_h_abs;
    if (int rc = radau_monitors(ctx, w.y, st->event_value)) return rc;
    return 0;
}
}  // namespace

extern "C" int marl_integrate_radau(marl_ctx* ctx, double* y, double t0, double t1, double first_step, double rtol, double atol,
                                    const int32_t* groups, const double* t_eval, int64_t n_eval, double* y_eval, double* t_events,
                                    int64_t max_events, int64_t max_attempts, marl_stats* stats)
{
    if (!ctx || !y || !stats || (n_eval > 0 && (!t_eval || !y_eval))) return ctx ? fail(ctx, -1, "marl_integrate_radau: invalid argument") : -1;
    if (ctx->batch != 1 || ctx->halo > 0) return fail(ctx, -1, "marl_integrate_radau: single-instance, whole-grid context required");
    if (!(first_step > 0) || !(t1 >= t0)) return fail(ctx, -1, "radau: need first_step > 0 and t1 >= t0 (forward integration)");
    if (t1 > t0 && first_step > t1 - t0) return fail(ctx, -1, "radau: `first_step` exceeds bounds");   // common.py:10-16
    if (!(rtol > 0) || !(atol >= 0)) return fail(ctx, -1, "radau: tolerances must be positive");
    for (int64_t i = 0; i < n_eval; i++)
        if (t_eval[i] < t0 || t_eval[i] > t1 || (i > 0 && t_eval[i] <= t_eval[i - 1]))
            return fail(ctx, -1, "radau: `t_eval` must be sorted and within t_span");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    RadauWork w;
    if (int rc = radau_alloc(ctx, w, groups)) return rc;
    const size_t n = (size_t)NF * ctx->N;
    HIP_OK(ctx, hipMemcpyAsync(w.y, y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = radau_run(ctx, w, t0, t1, first_step, rtol, atol, t_eval, n_eval, y_eval, t_events, max_events, max_attempts, stats)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(y, w.y, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int marl_integrate_bdf(marl_ctx* ctx, double* y, double t0, double t1, double first_step, double rtol, double atol, const int32_t* groups,
                                  const double* t_eval, int64_t n_eval, double* y_eval, double* t_events, int64_t max_events, int64_t max_attempts,
                                  marl_stats* stats)
{
    if (!ctx || !y || !stats || (n_eval > 0 && (!t_eval || !y_eval))) return ctx ? fail(ctx, -1, "marl_integrate_bdf: invalid argument") : -1;
    if (ctx->batch != 1 || ctx->halo > 0) return fail(ctx, -1, "marl_integrate_bdf: single-instance, whole-grid context required");
    if (!(first_step > 0) || !(t1 >= t0)) return fail(ctx, -1, "bdf: need first_step > 0 and t1 >= t0 (forward integration)");
    if (t1 > t0 && first_step > t1 - t0) return fail(ctx, -1, "bdf: `first_step` exceeds bounds");   // common.py:10-16
    if (!(rtol > 0) || !(atol >= 0)) return fail(ctx, -1, "bdf: tolerances must be positive");
    for (int64_t i = 0; i < n_eval; i++)
        if (t_eval[i] < t0 || t_eval[i] > t1 || (i > 0 && t_eval[i] <= t_eval[i - 1]))
            return fail(ctx, -1, "bdf: `t_eval` must be sorted and within t_span");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    RadauWork w;
    if (int rc = radau_alloc(ctx, w, groups)) return rc;
    const size_t n = (size_t)NF * ctx->N;
    HIP_OK(ctx, hipMemcpyAsync(w.y, y, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (int rc = bdf_run(ctx, w, t0, t1, first_step, rtol, atol, t_eval, n_eval, y_eval, t_events, max_events, max_attempts, stats)) return rc;
    HIP_OK(ctx, hipMemcpyAsync(y, w.y, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_OK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- a sweep of Radau instances (marl_radau_batch.h) -------------------------------------------------------------------------
extern "C" int marl_sweep_radau_events_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol, double atol,
                                           const int32_t* groups, int64_t max_attempts, double* t_events, int64_t max_events, marl_stats* stats);

extern "C" int marl_sweep_radau_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol, double atol,
                                    const int32_t* groups, int64_t max_attempts, marl_stats* stats)
{
    return marl_sweep_radau_events_dev(ctx, y_dev, t0, t1, first_step, rtol, atol, groups, max_attempts, nullptr, 0, stats);
}

// t_events (host, may be NULL): [instance][7][max_events] root times of the seven monitors, located inside the sweep as the single run
// locates them (dense output of the accepted step + Brent, one A_DENSE action per function evaluation); entries beyond n_events are NaN.
extern "C" int marl_sweep_radau_events_dev(marl_ctx* ctx, double* y_dev, double t0, double t1, double first_step, double rtol, double atol,
                                           const int32_t* groups, int64_t max_attempts, double* t_events, int64_t max_events, marl_stats* stats)
{
    if (!ctx || !y_dev || !stats) return ctx ? fail(ctx, -1, "marl_sweep_radau_dev: invalid argument") : -1;
    const bool locate = t_events != nullptr && max_events > 0;
    if (ctx->halo > 0) return fail(ctx, -1, "marl_sweep_radau_dev: whole-grid context required");
    if (!(first_step > 0) || !(t1 >= t0)) return fail(ctx, -1, "radau: need first_step > 0 and t1 >= t0 (forward integration)");
    if (t1 > t0 && first_step > t1 - t0) return fail(ctx, -1, "radau: `first_step` exceeds bounds");
    if (!(rtol > 0) || !(atol >= 0)) return fail(ctx, -1, "radau: tolerances must be positive");
    if (ctx->batch > 65535) return fail(ctx, -1, "marl_sweep_radau_dev: at most 65535 instances per call");
    const int64_t N = ctx->N, n = NF * N, B = ctx->batch;
    if (n > 8192) return fail(ctx, -1, "marl_sweep_radau_dev: sweeps are for small grids (N <= 1638); use marl_integrate_radau for one large grid");
    HIP_OK(ctx, hipSetDevice(ctx->device));
    rtol = clamp_rtol(rtol);
    RadauWork w;
    int64_t zs = 0;
    if (int rc = radau_alloc(ctx, w, groups, B, &zs)) return rc;
    using radau::RadauCtl;
    // controllers, monitors records, work lists + their counts
    RadauCtl* dctl = nullptr;
    double* drec = nullptr;
    int32_t* dcounts = nullptr;   // [L_COUNT] then lists [L_COUNT][B]
    HIP_OK(ctx, hipMalloc((void**)&dctl, sizeof(RadauCtl) * B));
    HIP_OK(ctx, hipMalloc((void**)&drec, sizeof(double) * NQ * B));
    HIP_OK(ctx, hipMalloc((void**)&dcounts, sizeof(int32_t) * (size_t)(radau::L_COUNT * (B + 1))));
    int32_t* dlists = dcounts + radau::L_COUNT;
    double *drec_dense = nullptr, *dtev = nullptr;   // event root finding: monitors of the dense-output states, the root times
    if (locate) {
        HIP_OK(ctx, hipMalloc((void**)&drec_dense, sizeof(double) * NQ * B));
        HIP_OK(ctx, hipMalloc((void**)&dtev, sizeof(double) * (size_t)(7 * max_events) * B));
        HIP_OK(ctx, hipMemsetAsync(drec_dense, 0, sizeof(double) * NQ * B, ctx->stream));
        HIP_OK(ctx, hipMemsetAsync(dtev, 0xff, sizeof(double) * (size_t)(7 * max_events) * B, ctx->stream));   // (all bits set: NaN)
    }
    unsigned* wg_next = nullptr;   // one-workgroup-per-instance paths: the instance queue, then the list of instances a pass visits [2 B]
    auto cleanup = [&]() {
        (void)hipFree(dctl); (void)hipFree(drec); (void)hipFree(dcounts);
        if (drec_dense) (void)hipFree(drec_dense);
        if (dtev) (void)hipFree(dtev);
        if (wg_next) (void)hipFree(wg_next);
    };
    std::vector<RadauCtl> hctl((size_t)B);
    for (auto& c : hctl) {
        memset(&c, 0, sizeof c);
        c.t_bound = t1; c.rtol = rtol; c.atol = atol; c.max_attempts = max_attempts;
        c.newton_tol = std::fmax(10 * radau::EPS / rtol, std::fmin(0.03, std::sqrt(rtol)));
        c.t = t0; c.S_h_abs = first_step; c.S_h_abs_old = -1; c.S_err_old = -1;
        c.pc = radau::PC_INIT; c.status = 1;
        c.locate_events = locate ? 1 : 0; c.max_events = max_events;
    }
    if (hipMemcpyAsync(dctl, hctl.data(), sizeof(RadauCtl) * B, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { cleanup(); return fail(ctx, -3, "copy failed"); }
    if (hipMemcpy2DAsync(w.y, (size_t)zs, y_dev, n * sizeof(double), n * sizeof(double), (size_t)B, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        cleanup();
        return fail(ctx, -3, "state copy failed");
    }
    (void)hipMemsetAsync(drec, 0, sizeof(double) * NQ * B, ctx->stream);
    const double S6 = std::sqrt(6.0);
    const double E3[3] = {(-13 - 7 * S6) / 3, (-13 + 7 * S6) / 3, -1.0 / 3};
    const radau::P33 P = {{{13.0 / 3 + 7 * S6 / 3, -23.0 / 3 - 22 * S6 / 3, 10.0 / 3 + 5 * S6},
                           {13.0 / 3 - 7 * S6 / 3, -23.0 / 3 + 22 * S6 / 3, 10.0 / 3 - 5 * S6},
                           {1.0 / 3, -8.0 / 3, 10.0 / 3}}};
    const int32_t* act0 = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(dctl) + offsetof(RadauCtl, action));
    // a launch over the instances of work list `which` (no action mask needed: the list IS the mask)
    auto Z = [&](int which) { return ZBatch{zs, act0, (int64_t)sizeof(RadauCtl), ~0, dlists + (int64_t)which * B}; };
    const dim3 b256(256);
    const unsigned gx = blocks256(n), gc = blocks256(N);
    const cplx c0 = {0, 0};
    const int64_t nbm = std::min<int64_t>((N + 255) / 256, 1024);
    if (int rc = ensure_part(ctx, (size_t)(nbm * B))) { cleanup(); return rc; }
    int rc_out = 0;
    int32_t* hcounts = reinterpret_cast<int32_t*>(ctx->rd_host);
    volatile int32_t* zc_words = nullptr;
    if (ctx->implicit_zero_copy) {
        if (!ctx->zc_h) {
            if (hipHostMalloc((void**)&ctx->zc_h, 32 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                hipHostGetDevicePointer((void**)&ctx->zc_d, ctx->zc_h, 0) != hipSuccess) { cleanup(); return fail(ctx, -3, "host allocation failed"); }
        }
        zc_words = reinterpret_cast<volatile int32_t*>(ctx->zc_h + 16);
        zc_words[radau::L_COUNT] = 0;
    }
    using namespace radau;
#define RB_OK()                                                                                       \
    do {                                                                                              \
        hipError_t e_ = hipGetLastError();                                                            \
        if (e_ != hipSuccess) { cleanup(); return fail(ctx, -100 - (int)e_, "kernel launch failed: %s", hipGetErrorString(e_)); } \
    } while (0)
    // Sweeps of small grids (5 N <= PCR_FUSED_MAX: the reference's N = 200), option radau_sweep_wg:
    //   1            HYBRID: one persistent workgroup per instance runs everything of the instance that is sequential and small (step
    //                logic, Newton iterations, error estimates, accepted steps, event roots - marl_radau_wg.h) and hands it back for
    //                Jacobians and factorisations, which the launch kernels do over work lists on the whole chip: one host cycle per
    //                Jacobian / factorisation instead of one per action;
    //   3 (default)  HYBRID with the finite-difference Jacobian in the workgroup too (60 us there against a host cycle of ~100 us): only
    //                factorisations come back to the host cycle - bit-identical to 1;
    //   2            the workgroup does those as well (no host in the loop; measured slower: cyclic reduction on ONE compute unit);
    //   0            the launch-per-action cycle (larger grids always take it).
    const int wg_mode = (n <= PCR_FUSED_MAX) ? (int)ctx->radau_sweep_wg : 0;
    const bool use_wg = wg_mode == 2 && !locate;
    WgWork ww{};
    if (wg_mode) {
        if (!ctx->cus) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) { cleanup(); return fail(ctx, -3, "device properties unavailable"); }
            ctx->cus = prop.multiProcessorCount;
        }
        ww.y = w.y; ww.f = w.f; ww.fnew = w.fnew; ww.ynew = w.ynew; ww.err = w.err; ww.yerr = w.yerr; ww.yold = w.yold; ww.scale = w.scale; ww.tmp = w.tmp;
        ww.Z = w.Z; ww.W = w.W; ww.F = w.F; ww.Q = w.Q; ww.YS = w.YS; ww.fac = w.fac; ww.h = w.h; ww.yscale = w.yscale; ww.maxdiff = w.maxdiff; ww.scl = w.scl;
        ww.hnew = w.hnew; ww.Jraw = w.Jraw; ww.YP = w.YP; ww.FN = w.FN; ww.J = w.J; ww.rhs_r = w.rhs_r; ww.rhs_c = w.rhs_c; ww.small = w.small; ww.groups = w.groups;
        ww.Sr = w.Sr; ww.Sc = w.Sc; ww.ng = w.ng; ww.nlevels = w.nlevels; ww.zs = zs;
        ww.plan = w.plan; ww.Cr = w.Cr; ww.Cc = w.Cc;
        if (hipMalloc((void**)&wg_next, sizeof(unsigned) * (size_t)(2 + 2 * B)) != hipSuccess) { wg_next = nullptr; cleanup(); return fail(ctx, -3, "allocation failed"); }
    }
    // nrun < 0: the pass visits every instance; else the instances run_list[0 .. nrun) (those the last pass handed back)
    auto launch_wg = [&](int hybrid, int64_t nrun) {
        (void)hipMemsetAsync(wg_next, 0, sizeof(unsigned), ctx->stream);
        const int64_t todo = nrun < 0 ? B : nrun;
        const dim3 grid((unsigned)std::max<int64_t>(1, std::min<int64_t>(todo, (int64_t)ctx->cus)));
        const int32_t* run_list = nrun < 0 ? nullptr : reinterpret_cast<const int32_t*>(wg_next + 2);
#define MARL_WG_LAUNCH(VD_, H_)                                                                                                             \
        hipLaunchKernelGGL((radau_wg_kernel<VD_, H_>), grid, dim3(WG_THREADS), 0, ctx->stream, dctl, B, N, ww, ctx->dconsts, atol, P, E3[0], E3[1], E3[2], \
                           wg_next, dcounts, dlists, dtev, run_list, todo)
        if (ctx->var_dphi) {
            if (hybrid == 0) MARL_WG_LAUNCH(true, 0); else if (hybrid == 1) MARL_WG_LAUNCH(true, 1); else MARL_WG_LAUNCH(true, 2);
        } else {
            if (hybrid == 0) MARL_WG_LAUNCH(false, 0); else if (hybrid == 1) MARL_WG_LAUNCH(false, 1); else MARL_WG_LAUNCH(false, 2);
        }
#undef MARL_WG_LAUNCH
    };
    if (use_wg) {
        launch_wg(0, -1);
        RB_OK();
    }
    int64_t wg_nrun = -1;
    const bool hybrid = wg_mode == 1 || wg_mode == 3 || (wg_mode == 2 && locate);
    const int hybrid_kind = wg_mode == 3 ? 2 : 1;   // 2: Jacobians stay in the workgroup, only factorisations come back to the host cycle
    for (int64_t cycle = 0; !use_wg; cycle++) {
        // the controllers advance every instance to its next piece of work and sort the instances into work lists; the host
        // reads the list lengths (one small copy + synchronisation per cycle) and launches each kind of work over its list only
        if (!zc_words || cycle == 0) (void)hipMemsetAsync(dcounts, 0, sizeof(int32_t) * L_COUNT, ctx->stream);   // (publish_counts_kernel zeroes them afterwards)
        if (hybrid) launch_wg(hybrid_kind, wg_nrun);   // every running instance up to its next Jacobian / factorisation (or its end)
        else hipLaunchKernelGGL(radau_control_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, ctx->stream, dctl, drec, B, n, dcounts, dlists, drec_dense, dtev);
        RB_OK();
        if (zc_words) {   // the list lengths through polled host memory
            hipLaunchKernelGGL(publish_counts_kernel, dim3(1), dim3(1), 0, ctx->stream, dcounts, reinterpret_cast<int32_t*>(ctx->zc_d + 16), (int32_t)(cycle + 1));
            RB_OK();
            int64_t spins = 0;
            while (zc_words[L_COUNT] != (int32_t)(cycle + 1)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                if (++spins > (int64_t)1 << 28) {
                    if (hipStreamSynchronize(ctx->stream) != hipSuccess || zc_words[L_COUNT] != (int32_t)(cycle + 1)) { cleanup(); return fail(ctx, -3, "sweep: the work-list lengths never arrived"); }
                }
            }
            for (int i = 0; i < L_COUNT; i++) hcounts[i] = zc_words[i];
        } else {
            (void)hipMemcpyAsync(hcounts, dcounts, sizeof(int32_t) * L_COUNT, hipMemcpyDeviceToHost, ctx->stream);
            if (hipStreamSynchronize(ctx->stream) != hipSuccess) { cleanup(); return fail(ctx, -3, "synchronize failed"); }
        }
        if (hcounts[L_RUNNING] == 0) break;
        if (hybrid) {   // the instances handed back in this pass are the ones the next pass visits (after their Jacobian / factorisation below)
            const int64_t nj = hcounts[L_JAC], nl = hcounts[L_LU];
            int32_t* run_list = reinterpret_cast<int32_t*>(wg_next + 2);
            if (nj) (void)hipMemcpyAsync(run_list, dlists + (int64_t)L_JAC * B, sizeof(int32_t) * (size_t)nj, hipMemcpyDeviceToDevice, ctx->stream);
            if (nl) (void)hipMemcpyAsync(run_list + nj, dlists + (int64_t)L_LU * B, sizeof(int32_t) * (size_t)nl, hipMemcpyDeviceToDevice, ctx->stream);
            wg_nrun = nj + nl;
        }
        const unsigned nR = (unsigned)hcounts[L_RHS1], nA = (unsigned)hcounts[L_ACCEPT], nJ = (unsigned)hcounts[L_JAC], nL = (unsigned)hcounts[L_LU],
                       nN = (unsigned)hcounts[L_NEWTON], nE = (unsigned)hcounts[L_ERR], nD = (unsigned)hcounts[L_DENSE];
        if (nD) {   // event root finding: the dense output of the accepted step at the abscissa Brent asks for, and that state's monitors
            hipLaunchKernelGGL(dense_eval_batch_kernel, dim3(gx, 1, nD), b256, 0, ctx->stream, w.Q, w.yold, n, w.tmp, Z(L_DENSE));
            RB_OK();
            hipLaunchKernelGGL(monitors_kernel<LAYOUT_FIELD_MAJOR>, dim3((unsigned)nbm, (unsigned)B), b256, 0, ctx->stream, w.tmp, ctx->dconsts, ctx->slab, zs / 8, ctx->part);
            RB_OK();
            hipLaunchKernelGGL(reduce_records_kernel, dim3((unsigned)B), b256, 0, ctx->stream, ctx->part, nbm, drec_dense);
            RB_OK();
        }
        if (nR) {   // a single state -> its derivative: y -> f (start), y + err -> tmp (second error estimate), y_new -> f_new (accepted step)
            if (ctx->var_dphi)
                hipLaunchKernelGGL((rhs_pick_kernel<LAYOUT_FIELD_MAJOR, true>), dim3(gc, 1, nR), b256, 0, ctx->stream, w.y, w.f, w.yerr, w.tmp, w.ynew, w.fnew, ctx->dconsts,
                                   ctx->slab, Z(L_RHS1));
            else
                hipLaunchKernelGGL((rhs_pick_kernel<LAYOUT_FIELD_MAJOR, false>), dim3(gc, 1, nR), b256, 0, ctx->stream, w.y, w.f, w.yerr, w.tmp, w.ynew, w.fnew, ctx->dconsts,
                                   ctx->slab, Z(L_RHS1));
            RB_OK();
        }
        if (nA) {
            hipLaunchKernelGGL(accept_kernel, dim3(gx, 1, nA), b256, 0, ctx->stream, w.Z, w.Q, w.y, w.yold, w.ynew, w.f, w.fnew, n, P, Z(L_ACCEPT));
            RB_OK();
        }
        if ((nA || cycle == 0) && !hybrid) {   // monitors of every instance's y (read by the controllers after the start and after each accepted step)
            hipLaunchKernelGGL(monitors_kernel<LAYOUT_FIELD_MAJOR>, dim3((unsigned)nbm, (unsigned)B), b256, 0, ctx->stream, w.y, ctx->dconsts, ctx->slab, zs / 8, ctx->part);
            RB_OK();
            hipLaunchKernelGGL(reduce_records_kernel, dim3((unsigned)B), b256, 0, ctx->stream, ctx->part, nbm, drec);
            RB_OK();
        }
        if (nJ) {   // finite-difference Jacobian at (y, f)
            hipLaunchKernelGGL(fd_prepare_kernel, dim3(gx, 1, nJ), b256, 0, ctx->stream, w.y, w.f, w.fac, atol, 0, w.groups, w.ng, n, w.h, w.yscale, w.YP, Z(L_JAC));
            RB_OK();
            for (int pass = 0; pass < 2; pass++) {
                if (ctx->var_dphi)
                    hipLaunchKernelGGL((rhs_kernel<LAYOUT_FIELD_MAJOR, true>), dim3(gc, (unsigned)w.ng, nJ), b256, 0, ctx->stream, w.YP, w.FN, ctx->dconsts, ctx->slab, n, 0, Z(L_JAC));
                else
                    hipLaunchKernelGGL(rhs_kernel<LAYOUT_FIELD_MAJOR>, dim3(gc, (unsigned)w.ng, nJ), b256, 0, ctx->stream, w.YP, w.FN, ctx->dconsts, ctx->slab, n, 0, Z(L_JAC));
                RB_OK();
                if (pass == 0)
                    hipLaunchKernelGGL(fd_columns_kernel, dim3(gx, 1, nJ), b256, 0, ctx->stream, w.y, w.f, w.FN, w.groups, w.ng, N, w.fac, w.yscale, w.Jraw, w.maxdiff, w.scl,
                                       w.small, w.hnew, w.YP, Z(L_JAC));
                else
                    hipLaunchKernelGGL(fd_finish_kernel, dim3(gx, 1, nJ), b256, 0, ctx->stream, w.f, w.FN, w.groups, N, w.fac, w.h, w.maxdiff, w.scl, w.small, w.hnew, w.Jraw,
                                       w.J, Z(L_JAC));
                RB_OK();
            }
        }
        if (nL && w.cr_k) {   // cyclic reduction in front of PCR (small grids; radau_alloc): levels 0 .. cr_k - 1, then PCR on the rows that are left
            hipLaunchKernelGGL(cr_init_kernel, dim3(pcr_groups((N + 1) / 2), 2, nL), b256, 0, ctx->stream, w.J, N, 0.0, c0, 1.0, w.Cr, w.Cc, Z(L_LU));
            RB_OK();
            for (int l = 0; l < w.cr_k; l++) {
                const CrShape sh{w.cr_n[l], w.cr_off[l], w.cr_n[l + 1], w.cr_off[l + 1]};
                hipLaunchKernelGGL(cr_reduce_kernel, dim3(pcr_groups(sh.n_next), 2, nL), b256, 0, ctx->stream, 0.0, c0, w.Cr, w.Cc, sh, l + 1 == w.cr_k ? 1 : 0, w.Sr, w.Sc,
                                   Z(L_LU));
                RB_OK();
            }
            const int64_t M = w.cr_n[w.cr_k];
            for (int level = 0; level < w.nlevels; level++) {
                hipLaunchKernelGGL(pcr_factor_kernel, dim3(pcr_groups(M), 2, nL), b256, 0, ctx->stream, w.J, M, level, 0.0, c0, w.Sr, w.Sc, Z(L_LU));
                RB_OK();
            }
        } else
        if (nL)
            for (int level = -1; level < w.nlevels; level++) {
                hipLaunchKernelGGL(pcr_factor_kernel, dim3((unsigned)((N + PCR_CELLS_PER_BLOCK - 1) / PCR_CELLS_PER_BLOCK), 2, nL), b256, 0, ctx->stream, w.J, N, level, 0.0, c0,
                                   w.Sr, w.Sc, Z(L_LU));
                RB_OK();
            }
        if (nN) {   // one Newton iteration
            hipLaunchKernelGGL(newton_begin_batch_kernel, dim3(gx, 1, nN), b256, 0, ctx->stream, w.y, w.Q, w.yold, n, w.scale, w.Z, w.W, w.YS, Z(L_NEWTON));
            RB_OK();
            if (ctx->var_dphi)
                hipLaunchKernelGGL((rhs_kernel<LAYOUT_FIELD_MAJOR, true>), dim3(gc, 3, nN), b256, 0, ctx->stream, w.YS, w.F, ctx->dconsts, ctx->slab, n, 0, Z(L_NEWTON));
            else
                hipLaunchKernelGGL(rhs_kernel<LAYOUT_FIELD_MAJOR>, dim3(gc, 3, nN), b256, 0, ctx->stream, w.YS, w.F, ctx->dconsts, ctx->slab, n, 0, Z(L_NEWTON));
            RB_OK();
            hipLaunchKernelGGL(newton_rhs_batch_kernel, dim3(gx, 1, nN), b256, 0, ctx->stream, w.F, w.W, N, w.rhs_r, w.rhs_c, dctl, Z(L_NEWTON));
            RB_OK();
        }
        for (int which = 0; which < 2; which++) {   // 0: both systems of the Newton iteration; 1: the real system of the error estimate
            const unsigned cnt = which == 0 ? nN : nE;
            if (!cnt) continue;
            const ZBatch zb = Z(which == 0 ? L_NEWTON : L_ERR);
            if (which == 1) {
                hipLaunchKernelGGL(error_rhs_batch_kernel, dim3(gx, 1, cnt), b256, 0, ctx->stream, w.f, w.tmp, w.Z, w.y, N, E3[0], E3[1], E3[2], w.rhs_r, w.ynew, zb);
                RB_OK();
            }
            if (n <= PCR_FUSED_MAX && ctx->radau_fused_solve) {
                hipLaunchKernelGGL(pcr_solve_fused_kernel, dim3(1, which == 0 ? 2 : 1, cnt), dim3(PCR_FUSED_THREADS), 0, ctx->stream, N, w.nlevels, 0, w.Sr, w.Sc, w.rhs_r,
                                   w.rhs_r, w.rhs_c, w.rhs_c, zb, w.plan, w.Cr, w.Cc);
                RB_OK();
            } else {
                const double* in_r = w.rhs_r;
                const cplx* in_c = w.rhs_c;
                for (int level = 0; level <= w.nlevels; level++) {
                    double* out_r = (level == w.nlevels) ? w.rhs_r : w.Sr.b[level & 1];
                    cplx* out_c = (level == w.nlevels) ? w.rhs_c : w.Sc.b[level & 1];
                    hipLaunchKernelGGL(pcr_solve_kernel, dim3(gx, which == 0 ? 2 : 1, cnt), b256, 0, ctx->stream, N, level, w.nlevels, 0, w.Sr, w.Sc, in_r, out_r, in_c, out_c, zb);
                    RB_OK();
                    in_r = out_r;
                    in_c = out_c;
                }
            }
            if (which == 0)
                hipLaunchKernelGGL(newton_update_batch_kernel, dim3(1, 1, cnt), dim3(1024), 0, ctx->stream, w.y, w.rhs_r, w.rhs_c, w.scale, N, w.W, w.Z, w.YS, dctl, zb);
            else
                hipLaunchKernelGGL(error_norm_batch_kernel, dim3(1, 1, cnt), dim3(1024), 0, ctx->stream, w.rhs_r, w.y, w.ynew, N, w.err, w.yerr, dctl, zb);
            RB_OK();
        }
    }
#undef RB_OK
    // results
    if (hipMemcpy2DAsync(y_dev, n * sizeof(double), w.y, (size_t)zs, n * sizeof(double), (size_t)B, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess)
        rc_out = fail(ctx, -3, "state copy failed");
    if (rc_out == 0 && hipMemcpyAsync(hctl.data(), dctl, sizeof(RadauCtl) * B, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc_out = fail(ctx, -3, "copy failed");
    if (rc_out == 0 && locate && hipMemcpyAsync(t_events, dtev, sizeof(double) * (size_t)(7 * max_events) * B, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess)
        rc_out = fail(ctx, -3, "copy failed");
    if (rc_out == 0 && hipStreamSynchronize(ctx->stream) != hipSuccess) rc_out = fail(ctx, -3, "synchronize failed");
    if (rc_out == 0)
        for (int64_t b = 0; b < B; b++) {
            const RadauCtl& c = hctl[(size_t)b];
            marl_stats& st = stats[b];
            memset(&st, 0, sizeof st);
            st.nfev = c.nfev; st.njev = c.njev; st.nlu = c.nlu; st.n_accepted = c.n_acc; st.n_rejected = c.n_rej;
            st.status = c.status; st.t = c.t; st.h_next = c.S_h_abs;
            for (int e = 0; e < 7; e++) { st.event_value[e] = c.g[e]; st.n_events[e] = c.n_events[e]; }
        }
    cleanup();
    return rc_out;
}

#endif  // MARL_LAB
