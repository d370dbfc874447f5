// marl_kernels.h - HIP kernels for the five-field RHS and the fused explicit RK time loops (gfx950).
//
// Work decomposition (all kernels): a workgroup of BLK threads owns a WINDOW of BLK*CPT
// consecutive depth cells, CPT consecutive cells per thread, all five fields of a cell in that
// thread's registers.  A Runge-Kutta stage needs u[i-1], u[i], u[i+1] of the current STAGE state:
// cells inside a thread come from registers, the two thread-edge cells are exchanged through a
// double-buffered LDS edge array (one barrier per stage).  Stage vectors K1..Ks never leave
// registers.  A window is wider than the cells it finally writes by H cells per side; every RHS
// evaluation invalidates one more cell at each window edge (the halo is recomputed, never
// communicated), so one launch can advance several stages - a whole RK4 step, NSTEPS RK4 steps, or a
// whole Dormand-Prince attempt - with ONE read and ONE write of the state: 80 algorithmic bytes per
// grid-point-step.  Physical boundaries (global cell 0 / N-1) use the ghost-cell rules instead of
// a halo and are therefore always valid.
//
// State layouts (`LAYOUT`):
//   FIELD_MAJOR  y[f*ld + i]                         the reference's layout (Evolve_scenario.py:64-65)
//   TILED        y[(i>>6)*320 + f*64 + (i&63)]       five 512-byte field rows interleaved per 64-cell
//                                                    tile: one window = one contiguous 40*WIN-byte run
#pragma once
#include <type_traits>

#include "marl_math.h"

namespace marl {

enum : int { LAYOUT_FIELD_MAJOR = 0, LAYOUT_TILED = 1 };

template <int LAYOUT>
__device__ __forceinline__ int64_t at(int f, int64_t i, int64_t ld)
{
    if constexpr (LAYOUT == LAYOUT_FIELD_MAJOR)
        return (int64_t)f * ld + i;
    else
        return (i >> 6) * (int64_t)(NF * 64) + f * 64 + (i & 63);
}

// A contiguous piece of the global grid held in one buffer: local cell l is global cell l + goff.
// Single-GPU runs: n_buf = N, goff = 0.  Domain decomposition: the buffer carries exchanged halo
// cells on interior sides.
struct Slab {
    int64_t n_buf;   // cells in the buffer
    int64_t goff;    // global index of local cell 0
    int64_t ld;      // field stride of FIELD_MAJOR buffers (>= n_buf)
    int64_t out_lo;  // local cells [out_lo, out_hi) are written by fused kernels
    int64_t out_hi;
};

__device__ __forceinline__ double nanmin(double a, double b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ double nanmax(double a, double b) { return (a > b || a != a) ? a : b; }
// FINITE: the caller only uses the result when every input is finite (the monitors of an ACCEPTED Dormand-Prince attempt: a
// non-finite y_new, f(y_new), U or W makes the error norm NaN and the attempt is rejected; its record is never read) - plain
// v_min_f64 / v_max_f64 then (one instruction instead of the four of a NaN-propagating select; np.min / np.max agree with them
// on finite values).
template <bool FINITE = false>
__device__ __forceinline__ double mon_min(double a, double b) { return FINITE ? __builtin_fmin(a, b) : nanmin(a, b); }
template <bool FINITE = false>
__device__ __forceinline__ double mon_max(double a, double b) { return FINITE ? __builtin_fmax(a, b) : nanmax(a, b); }

// ---------------------------------------------------------------------------------------------
// Per-block stencil engine
// ---------------------------------------------------------------------------------------------
// REUSE = false: no transcendental cache (one-workgroup sweeps on coarse grids, where the stage displacements
// are far outside the expansions' range).  The cache (marl_math.h, PointCache) also needs its LDS slots to fit
// the 64 KB a workgroup may declare: wide variants (512 threads, several cells per thread) run without it.
// VD: the time-varying porosity diffusion coefficient (marl_params.dPhi_variable), see marl_math.h.
// A lane's neighbour values straight from the neighbouring LANES of its wave (gfx9 DPP wave shifts: two v_mov_b32_dpp per double,
// VALU speed, no LDS, no barrier).  from_left: lane i receives lane i - 1's value; from_right: lane i + 1's.  Lane 0 / lane 63 receive
// their own value back - WAVE_TILE windows make those two lanes halo cells whose results are never written.
__device__ __forceinline__ double wave_from_left(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false));   // wave_shr:1
}
__device__ __forceinline__ double wave_from_right(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(__builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false), __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false));   // wave_shl:1
}

#ifdef MARL_LAB_PHASE_CLOCK   // kernel-lab build only (tools/rk4_lab.hip): shader-clock cycles per phase of an RHS evaluation, summed over every
// evaluation of wave 0 of every workgroup: [0] edge writes, [1] own-cell phase (point_local), [2] wait at the exchange barrier, [3] neighbour
// reads + boundary branch + stencil phase (point_rates), [4] between evaluations (RK combination, loads, stores), [5] evaluations counted
__device__ unsigned long long marl_lab_phase[8];
#define MARL_PHASE_MARK(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_acc[k] += t_ - ph_last; ph_last = t_; } while (0)
#else
#define MARL_PHASE_MARK(k) do { } while (0)
#endif

template <int BLK, int CPT, bool REUSE = true, bool VD = false>
struct StencilBlock {
    static constexpr int WIN = BLK * CPT;
    static constexpr int NSIDE = (CPT == 1) ? 1 : 2;
    // WAVE_TILE (kernel lab only - tools/rk4_lab.hip -DLAB_BLK=64; no shipped kernel instantiates it): the window is ONE wave (64
    // cells, one per lane): neighbours come from the wave's own lanes by DPP, the two edge lanes are halo - a stage has no LDS
    // exchange and no barrier.  Built for BASELINE configs[1] (N = 65 536: one or two waves per SIMD), measured, NOT faster: 3.3 -
    // 3.8 us per additional step at 1.3 - 2 waves per SIMD against 3.64 us for the shipped 256-thread / 16-step kernel - a lone wave
    // of this instruction stream issues one VALU instruction per ~12 cycles with or without the exchange (profiles/r03_lab_n65536.log)
    static constexpr bool WAVE_TILE = (BLK == 64 && CPT == 1);
    static constexpr int EDGE_DOUBLES = WAVE_TILE ? 0 : 2 * NSIDE * NF * BLK;       // edges [parity][side][field][thread]
    static constexpr bool CACHE_LDS = PC_LDS_SLOTS > 0;
    static constexpr bool CACHE = REUSE && (EDGE_DOUBLES + TABLE_DOUBLES + PC_LDS_SLOTS * WIN) * 8 <= 60 * 1024;
    static constexpr int CACHE_DOUBLES = CACHE ? PC_LDS_SLOTS * WIN : 0;
    static constexpr int LDS_DOUBLES = EDGE_DOUBLES + TABLE_DOUBLES + CACHE_DOUBLES;  // edges, log/exp tables, cache slots [slot][cell][thread]

    double* lds;
    Tables T;
    HotConsts K;                 // hot constants, in registers for the whole kernel
    const DevConsts* C;          // full constant block (cold members are read on rare paths only)
    int tid;
    int parity;
    // the thread's c-th cell is global cell 0 (bit c) / global cell N-1 (bit 8 + c) / in the dissolution zone (bit 16 + c): ONE register
    // instead of three (it lives across every stage, and the fused integrators sit at their register cap)
    static_assert(CPT <= 8, "three 8-bit fields");
    unsigned masks;
    __device__ __forceinline__ unsigned is_first(int c) const { return (masks >> c) & 1u; }
    __device__ __forceinline__ unsigned is_last(int c) const { return (masks >> (8 + c)) & 1u; }
    __device__ __forceinline__ unsigned in_zone(int c) const { return (masks >> (16 + c)) & 1u; }
    __device__ __forceinline__ unsigned is_edge(int c) const { return (masks >> c) & 0x101u; }   // first or last: non-zero
    PointCache<(CACHE && CACHE_LDS) ? WIN : 0> cache[CACHE ? CPT : 1];  // centre of the transcendental expansions (TR_FILL / TR_REUSE / TR_AUTO)
    bool reuse_live[CPT] = {};   // wave-uniform, per cell: the centre is filled and no evaluation since has fallen out of range
#ifdef MARL_LAB_PHASE_CLOCK
    unsigned long long ph_acc[6] = {0, 0, 0, 0, 0, 0}, ph_last = __builtin_amdgcn_s_memtime();
    __device__ __forceinline__ void phase_flush()
    {
        if (threadIdx.x == 0)
            for (int k = 0; k < 6; k++) atomicAdd(&marl_lab_phase[k], ph_acc[k]);
    }
#endif

    // lds: LDS_DOUBLES doubles = edge exchange buffers followed by the log/exp tables (copied here; barrier inside).
    // g0: global index of this thread's first cell.
    __device__ __forceinline__ StencilBlock(double* lds_, int64_t g0, const DevConsts* __restrict__ c)
        : lds(lds_), T(load_tables(lds_ + EDGE_DOUBLES, BLK)), K(load_hot(c)), C(c), tid(threadIdx.x), parity(0)
    {
        set_window(g0);
        if constexpr (CACHE && CACHE_LDS) {
#pragma unroll
            for (int c = 0; c < CPT; c++) cache[c].s = lds_ + EDGE_DOUBLES + TABLE_DOUBLES + c * BLK + tid;
        }
    }

    // Persistent kernels: the per-thread members again, from an index the caller has made opaque inside its work loop - everything
    // derived from the thread index is then recomputed per work item (a few instructions) instead of being hoisted out of the loop and
    // kept in (or spilled from) vector registers across it.
    __device__ __forceinline__ void rebind(int t)
    {
        tid = t;
        if constexpr (CACHE && CACHE_LDS) {
#pragma unroll
            for (int c = 0; c < CPT; c++) cache[c].s = lds + EDGE_DOUBLES + TABLE_DOUBLES + c * BLK + t;
        }
    }

    // (Re)position the window.
    __device__ __forceinline__ void set_window(int64_t g0)
    {
        const int64_t N = C->N, mlo = C->mask_lo, mhi = C->mask_hi;
        masks = 0;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int64_t g = g0 + i;
            masks |= ((g == 0) ? (1u << i) : 0u) | ((g == N - 1) ? (0x100u << i) : 0u) | ((g >= mlo && g < mhi) ? (0x10000u << i) : 0u);
        }
    }

    // k[c] = RHS(stage state ys) for the thread's CPT cells.  Contains exactly one __syncthreads(): the own-cell phase
    // of the evaluation (marl_math.h, point_local) runs between the edge writes and the barrier, the stencil phase
    // after it - the neighbour values are not live during the transcendental part.
    // MODE: see marl_math.h (TR_FILL / TR_AUTO for a first stage, TR_REUSE for the following ones, TR_PLAIN otherwise).
    template <int MODE = TR_PLAIN>
    __device__ __forceinline__ void eval(const double (&ys)[CPT][NF], double (&k)[CPT][NF], PointAux (&aux)[CPT])
    {
        if constexpr (WAVE_TILE) {
            PointLocal pl;
            point_local<CACHE ? MODE : TR_PLAIN, (CACHE && CACHE_LDS) ? WIN : 0, VD>(ys[0], in_zone(0), K, C, T, pl, aux[0], cache[0], reuse_live[0]);
            double um[NF], up[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) um[f] = wave_from_left(ys[0][f]);
            const bool need_right_solids = __builtin_amdgcn_ballot_w64(!pl.upw) != 0;
#pragma unroll
            for (int f = 0; f < NF; f++) up[f] = (f >= 2 || need_right_solids) ? wave_from_right(ys[0][f]) : 0.0;
            if (__builtin_amdgcn_ballot_w64(is_edge(0) != 0) != 0) {   // physical boundaries: two cells of the whole grid
                if (is_last(0)) {
#pragma unroll
                    for (int f = 0; f < NF; f++) up[f] = ghost_upper(f, ys[0][f], um[f]);
                }
                if (is_first(0)) {
#pragma unroll
                    for (int f = 0; f < NF; f++) um[f] = ghost_lower(C->bc[f], ys[0][f]);
                }
            }
            point_rates<VD, true, !CACHE>(ys[0], um, up, K, T, pl, k[0], need_right_solids);
            return;
        }
        MARL_PHASE_MARK(4);
        double* e = lds + parity * (NSIDE * NF * BLK);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            e[f * BLK + tid] = ys[0][f];
            if constexpr (CPT > 1) e[(NF + f) * BLK + tid] = ys[CPT - 1][f];
        }
        MARL_PHASE_MARK(0);
        PointLocal pl[CPT];
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            point_local<CACHE ? MODE : TR_PLAIN, (CACHE && CACHE_LDS) ? WIN : 0, VD>(ys[c], in_zone(c), K, C, T, pl[c], aux[c], cache[CACHE ? c : 0], reuse_live[c]);
            // one cell at a time: interleaving the cells' evaluations doubles the live temporaries (spills at CPT >= 2)
#ifndef MARL_LAB_INTERLEAVE_CELLS   // (kernel-lab switch: let the scheduler interleave the cells of a thread - ILP instead of registers)
            if constexpr (CPT > 1) __builtin_amdgcn_sched_barrier(0);
#endif
        }
        MARL_PHASE_MARK(1);
        __syncthreads();
        MARL_PHASE_MARK(2);
        const int tl = tid > 0 ? tid - 1 : 0;
        const int tr = tid < BLK - 1 ? tid + 1 : BLK - 1;
        double left[NF], right[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) left[f] = e[((NSIDE - 1) * NF + f) * BLK + tl];  // left neighbour's LAST cell
        // the solids are differenced against the upwind neighbour only: with U > 0 in every lane (burial - the normal
        // case) the right-hand values of CA and CC are never used and their two LDS reads are skipped
        const bool need_right_solids = __builtin_amdgcn_ballot_w64(!pl[CPT - 1].upw) != 0;
#pragma unroll
        for (int f = 0; f < NF; f++) right[f] = (f >= 2 || need_right_solids) ? e[f * BLK + tr] : 0.0;  // right neighbour's FIRST cell
        parity ^= 1;
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            double um[NF], up[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) {
                um[f] = (c == 0) ? left[f] : ys[c > 0 ? c - 1 : 0][f];
                up[f] = (c == CPT - 1) ? right[f] : ys[c < CPT - 1 ? c + 1 : c][f];
            }
            // physical boundaries: two cells of the whole grid - a wave-uniform branch, skipped by every other wave
            if (__builtin_amdgcn_ballot_w64(is_edge(c) != 0) != 0) {
                if (is_last(c)) {
#pragma unroll
                    for (int f = 0; f < NF; f++) up[f] = ghost_upper(f, ys[c][f], um[f]);
                }
                if (is_first(c)) {
#pragma unroll
                    for (int f = 0; f < NF; f++) um[f] = ghost_lower(C->bc[f], ys[c][f]);
                }
            }
            point_rates<VD, true, !CACHE>(ys[c], um, up, K, T, pl[c], k[c], CPT == 1 ? need_right_solids : true);
        }
#ifdef MARL_LAB_PHASE_CLOCK
        asm volatile("" :: "v"(k[0][0]), "v"(k[0][4]));   // (the rates are complete before the mark)
        MARL_PHASE_MARK(3);
        ph_acc[5]++;
#endif
    }
};

// Block-wide reduction of NQ quantities: q[0] summed, q[1..NMIN] min-reduced, the rest max-reduced, in a fixed
// order (deterministic).  Result valid in thread 0.  `scratch`: NQ * BLK doubles of LDS that no thread still reads
// for another purpose once it has arrived here (the stencil kernels pass their edge-exchange buffers).
// The quantities are transposed through LDS so that each wave reduces whole quantities: NQ/waves butterfly
// reductions per wave instead of NQ (a 64-lane butterfly of one double costs 12 ds_bpermute + the combines).
// FINITE: the extrema without NaN propagation (mon_min); the sum q[0] is unaffected - a NaN error norm still rejects the attempt.
template <int BLK, int NQ, int NMIN, bool FINITE = false>
__device__ __forceinline__ void block_reduce(double (&q)[NQ], double* scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = BLK / 64;
    auto combine = [](int j, double a, double o) { return (j == 0) ? a + o : (j <= NMIN ? mon_min<FINITE>(a, o) : mon_max<FINITE>(a, o)); };
    if constexpr (NW > 1) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NQ; j++) scratch[j * BLK + threadIdx.x] = q[j];
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < (NQ + NW - 1) / NW; jj++) {
            const int j = wave + jj * NW;   // wave-uniform
            if (j < NQ) {
                double a = scratch[j * BLK + lane];
#pragma unroll
                for (int i = 1; i < NW; i++) a = combine(j, a, scratch[j * BLK + lane + 64 * i]);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) a = combine(j, a, __shfl_xor(a, off, 64));
                if (lane == 0) scratch[j * BLK] = a;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < NQ; j++) q[j] = scratch[j * BLK];
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int j = 0; j < NQ; j++) q[j] = combine(j, q[j], __shfl_xor(q[j], off, 64));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone RHS: dydt = f(y).  The drop-in for the reference's fun / fun_numba callable
// (marlpde/LHeureux_model.py:162, :290).  One thread per cell, neighbours straight from L1/L2.
// ---------------------------------------------------------------------------------------------
// Batches of instances that each follow their own control flow (the batched implicit path): blockIdx.z = instance, whose data
// sit `zstride` BYTES behind instance 0's, and whose 32-bit action word (at act + z * act_stride bytes) says what it needs in this
// cycle: a kernel launched for the actions `want` (a bit mask) returns at once for every other instance.  act == NULL: no masking.
// `list` (may be NULL): the launch covers only the instances list[0 .. gridDim.z) - a compacted index list the controller wrote for
// this kind of work - instead of all of them.
struct ZBatch {
    int64_t zstride;
    const int32_t* act;
    int64_t act_stride;
    int want;
    const int32_t* list;
};
// (readfirstlane: the index is the same in every lane, but a load through a struct member is not provably so - without it every
// shifted pointer becomes a per-lane 64-bit address and the PCR kernels of a single large grid ran 3x slower)
__device__ __forceinline__ int64_t z_inst(const ZBatch& B)
{
    return B.list ? (int64_t)__builtin_amdgcn_readfirstlane(B.list[blockIdx.z]) : (int64_t)blockIdx.z;
}
__device__ __forceinline__ bool z_masked_out(const ZBatch& B)
{
    return B.act && !(*reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(B.act) + z_inst(B) * B.act_stride) & B.want);
}
template <class P>
__device__ __forceinline__ P* z_shift(P* p, const ZBatch& B)
{
    return reinterpret_cast<P*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<P>::type*>(p)) + z_inst(B) * B.zstride);
}

// blockIdx.y: instance (state at y + blockIdx.y * inst_stride, constants consts[blockIdx.y * const_stride]; const_stride = 0:
// several states of ONE model - the stage states / finite-difference columns of the implicit path); blockIdx.z: ZBatch instance
// (constants consts[blockIdx.z] then)
template <int LAYOUT, bool VD = false>
__global__ void __launch_bounds__(256) rhs_kernel(const double* __restrict__ y, double* __restrict__ dydt,
                                                  const DevConsts* __restrict__ consts, Slab S, int64_t inst_stride, int const_stride,
                                                  ZBatch B = ZBatch{0, nullptr, 0, 0, nullptr})
{
    if (z_masked_out(B)) return;
    __shared__ double tabs[TABLE_DOUBLES];
    const Tables T = load_tables(tabs, 256);
    const DevConsts& C = consts[blockIdx.y * const_stride + z_inst(B)];
    y = z_shift(y, B) + blockIdx.y * inst_stride;
    dydt = z_shift(dydt, B) + blockIdx.y * inst_stride;
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l < S.out_lo || l >= S.out_hi) return;
    const int64_t g = l + S.goff;
    double uc[NF], um[NF], up[NF], r[NF];
    PointAux aux;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        uc[f] = y[at<LAYOUT>(f, l, S.ld)];
        um[f] = (g > 0) ? y[at<LAYOUT>(f, l - 1, S.ld)] : ghost_lower(C.bc[f], uc[f]);
    }
#pragma unroll
    for (int f = 0; f < NF; f++) up[f] = (g < C.N - 1) ? y[at<LAYOUT>(f, l + 1, S.ld)] : ghost_upper(f, uc[f], um[f]);
    const HotConsts K = load_hot(&C);
    PointCache<0> pc;
    bool live = false;
    rhs_point<TR_PLAIN, 0, VD>(uc, um, up, g >= C.mask_lo && g < C.mask_hi, K, &C, T, r, aux, pc, live);
#pragma unroll
    for (int f = 0; f < NF; f++) dydt[at<LAYOUT>(f, l, S.ld)] = r[f];
}

// Element-wise evaluation of the device math primitives (accuracy tests only; marl_debug_math).
__global__ void __launch_bounds__(256) math_probe_kernel(int op, const double* __restrict__ x, double* __restrict__ y, int64_t n, double e)
{
    __shared__ double tabs[TABLE_DOUBLES];
    const Tables T = load_tables(tabs, 256);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r;
    switch (op) {
        case 0: r = fast_log(v, T); break;
        case 1: r = fast_exp(v, T); break;
        case 2: r = pow_sat(v, e, T); break;
        case 3: r = rcp_nr(v); break;
        case 4: r = fv_sigma<false>(v, e, T); break;
        default: r = fv_sigma<true>(v, e, T); break;
    }
    y[i] = r;
}

// Layout conversion FIELD_MAJOR <-> TILED (entry / exit of the fused integrators only).
template <int SRC, int DST>
__global__ void __launch_bounds__(256) convert_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n,
                                                      int64_t ld_src, int64_t ld_dst, int64_t stride_src, int64_t stride_dst)
{
    src += blockIdx.y * stride_src;
    dst += blockIdx.y * stride_dst;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int f = 0; f < NF; f++) dst[at<DST>(f, i, ld_dst)] = src[at<SRC>(f, i, ld_src)];
}

// ---------------------------------------------------------------------------------------------
// The seven monitors (marlpde/LHeureux_model.py:524-593) as per-block partials:
//   part[b][0] unused (sum slot), [1] min(y) [2] min CA [3] min CC [4] min U [5] max(CA+CC) [6] max Phi [7] max W
// ---------------------------------------------------------------------------------------------
constexpr int NQ = 8;    // reduction record: sum-of-squares + 7 monitor extrema
constexpr int NQMIN = 4; // slots 1..4 are minima, 5..7 maxima

__device__ __forceinline__ void monitors_init(double (&q)[NQ])
{
    q[0] = 0.0;
    q[1] = q[2] = q[3] = q[4] = __builtin_inf();
    q[5] = q[6] = q[7] = -__builtin_inf();
}

template <bool FINITE = false>   // (see mon_min)
__device__ __forceinline__ void monitors_accumulate(double (&q)[NQ], const double (&u)[NF], double U, double W)
{
    q[1] = mon_min<FINITE>(mon_min<FINITE>(mon_min<FINITE>(mon_min<FINITE>(mon_min<FINITE>(q[1], u[0]), u[1]), u[2]), u[3]), u[4]);
    q[2] = mon_min<FINITE>(q[2], u[0]);
    q[3] = mon_min<FINITE>(q[3], u[1]);
    q[4] = mon_min<FINITE>(q[4], U);
    q[5] = mon_max<FINITE>(q[5], u[0] + u[1]);
    q[6] = mon_max<FINITE>(q[6], u[4]);
    q[7] = mon_max<FINITE>(q[7], W);
}

template <int LAYOUT>
__global__ void __launch_bounds__(256) monitors_kernel(const double* __restrict__ y, const DevConsts* __restrict__ consts,
                                                       Slab S, int64_t inst_stride, double* __restrict__ part)
{
    __shared__ double scratch[NQ * 256];
    __shared__ double tabs[TABLE_DOUBLES];
    const Tables T = load_tables(tabs, 256);
    const DevConsts& C = consts[blockIdx.y];
    y += blockIdx.y * inst_stride;
    double q[NQ];
    monitors_init(q);
    for (int64_t l = S.out_lo + (int64_t)blockIdx.x * 256 + threadIdx.x; l < S.out_hi; l += (int64_t)gridDim.x * 256) {
        double u[NF];
#pragma unroll
        for (int f = 0; f < NF; f++) u[f] = y[at<LAYOUT>(f, l, S.ld)];
        const double Phi = u[4];
        const double F = 1.0 - fast_exp(10.0 - 10.0 * rcp_nr(Phi), T);
        const double rF = C.hot.rhorat * F;
        monitors_accumulate(q, u, C.hot.presum + rF * (Phi * Phi * Phi) * rcp_nr(1.0 - Phi), C.hot.presum - rF * Phi * Phi);
    }
    block_reduce<256, NQ, NQMIN>(q, scratch);
    if (threadIdx.x == 0) {
        double* p = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * NQ;
#pragma unroll
        for (int j = 0; j < NQ; j++) p[j] = q[j];
    }
}

// Deterministic second-level reduction: `nblocks` records -> one record per instance.
__global__ void __launch_bounds__(256) reduce_records_kernel(const double* __restrict__ part, int64_t nblocks, double* __restrict__ out)
{
    __shared__ double scratch[NQ * 256];
    part += (int64_t)blockIdx.x * nblocks * NQ;
    double q[NQ];
    monitors_init(q);
    for (int64_t b = threadIdx.x; b < nblocks; b += 256) {
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const double o = part[b * NQ + j];
            q[j] = (j == 0) ? q[j] + o : (j <= NQMIN ? nanmin(q[j], o) : nanmax(q[j], o));
        }
    }
    block_reduce<256, NQ, NQMIN>(q, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NQ; j++) out[(int64_t)blockIdx.x * NQ + j] = q[j];
    }
}

// First level of a two-level reduction of MANY records (one per workgroup of a large grid): workgroup b reduces the records
// [b * chunk, min(nrec, (b + 1) * chunk)) into out[b] - a fixed partition, so the result is reproducible.  One workgroup reading
// thousands of records alone is latency-bound (17 000 records at N = 2^22: ~60 us); 64 workgroups take ~3 us.
__global__ void __launch_bounds__(256) reduce_chunks_kernel(const double* __restrict__ part, int64_t nrec, int64_t chunk, double* __restrict__ out)
{
    __shared__ double scratch[NQ * 256];
    const int64_t lo = (int64_t)blockIdx.x * chunk, hi = (lo + chunk < nrec) ? lo + chunk : nrec;
    double q[NQ];
    monitors_init(q);
    for (int64_t b = lo + threadIdx.x; b < hi; b += 256) {
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const double o = part[b * NQ + j];
            q[j] = (j == 0) ? q[j] + o : (j <= NQMIN ? nanmin(q[j], o) : nanmax(q[j], o));
        }
    }
    block_reduce<256, NQ, NQMIN>(q, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NQ; j++) out[(int64_t)blockIdx.x * NQ + j] = q[j];
    }
}

// ---------------------------------------------------------------------------------------------
// Shared load / store of a thread's cells
// ---------------------------------------------------------------------------------------------
template <int CPT, int LAYOUT>
__device__ __forceinline__ void load_cells(const double* __restrict__ y, int64_t l0, const Slab& S, const DevConsts& C,
                                           double (&u)[CPT][NF])
{
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t l = l0 + c;
        const bool in = l >= 0 && l < S.n_buf;
#pragma unroll
        for (int f = 0; f < NF; f++) u[c][f] = in ? y[at<LAYOUT>(f, l, S.ld)] : C.bc[f];
    }
}

// ---------------------------------------------------------------------------------------------
// Fused classical RK4: NSTEPS whole steps per launch (BASELINE config 2 / headline).
//   k1 = f(y); k2 = f(y + dt/2 k1); k3 = f(y + dt/2 k2); k4 = f(y + dt k3)
//   y <- y + dt/6 (((k1 + 2 k2) + 2 k3) + k4)                 (the order the CPU checker uses too)
// ---------------------------------------------------------------------------------------------
#ifdef MARL_LAB_CLOCK  // kernel-lab diagnostic build only: in-kernel shader clock (s_memtime) vs 100 MHz s_memrealtime
__device__ unsigned long long marl_lab_clock[3 * 16384];
#endif
// NSTEPS classical RK4 steps on the thread's cells (every evaluation invalidates one more cell per window edge).
template <int NSTEPS, class SB, int CPT>
__device__ __forceinline__ void rk4_advance(SB& sb, double (&y)[CPT][NF], double dt)
{
    double ys[CPT][NF], k[CPT][NF], acc[CPT][NF];
    PointAux aux[CPT];
    const double h2 = 0.5 * dt, h6 = dt / 6.0;
#pragma unroll 1
    for (int step = 0; step < NSTEPS; step++) {
        sb.template eval<TR_AUTO>(y, k, aux);   // the centre of the expansions survives from step to step
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int f = 0; f < NF; f++) { acc[c][f] = k[c][f]; ys[c][f] = y[c][f] + h2 * k[c][f]; }
        sb.template eval<TR_REUSE>(ys, k, aux);
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int f = 0; f < NF; f++) { acc[c][f] = acc[c][f] + 2.0 * k[c][f]; ys[c][f] = y[c][f] + h2 * k[c][f]; }
        sb.template eval<TR_REUSE>(ys, k, aux);
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int f = 0; f < NF; f++) { acc[c][f] = acc[c][f] + 2.0 * k[c][f]; ys[c][f] = y[c][f] + dt * k[c][f]; }
        sb.template eval<TR_REUSE>(ys, k, aux);
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int f = 0; f < NF; f++) y[c][f] = y[c][f] + h6 * (acc[c][f] + k[c][f]);
    }
}

// One cell per thread: 4 waves per SIMD (<= 128 VGPRs; 4 workgroups of 256 share the CU's 160 KB of LDS).
// Variants with more cells per thread keep the compiler's own choice.
#ifndef MARL_LAB_RK4_WAVES_MIN   // kernel-lab switch: occupancy window the register allocator aims at
#define MARL_LAB_RK4_WAVES_MIN (CPT == 1 ? 4 : 1)
#define MARL_LAB_RK4_WAVES_MAX 8
#endif
template <int BLK, int CPT, int LAYOUT, int NSTEPS, bool VD = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(MARL_LAB_RK4_WAVES_MIN, MARL_LAB_RK4_WAVES_MAX))) rk4_fused_kernel(const double* __restrict__ yin, double* __restrict__ yout,
                                                        const DevConsts* __restrict__ consts, Slab S, double dt)
{
#ifdef MARL_LAB_CLOCK
    const unsigned long long lab_t0 = __builtin_amdgcn_s_memtime(), lab_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    constexpr int H = 4 * NSTEPS;
    constexpr int WIN = BLK * CPT;
    constexpr int V = WIN - 2 * H;
    static_assert(V > 0, "window too small for the fused halo");
    using SB = StencilBlock<BLK, CPT, true, VD>;
    __shared__ double lds[SB::LDS_DOUBLES];
    const DevConsts& C = consts[0];

    const int64_t w0 = S.out_lo + (int64_t)blockIdx.x * V - H;  // window start, local index
    const int64_t l0 = w0 + (int64_t)threadIdx.x * CPT;
    double y[CPT][NF];
    load_cells<CPT, LAYOUT>(yin, l0, S, C, y);              // in flight while the tables are staged
    SB sb(lds, l0 + S.goff, consts);   // (table copy + barrier inside)
    rk4_advance<NSTEPS>(sb, y, dt);

#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int wi = threadIdx.x * CPT + c;
        const int64_t l = l0 + c;
        if (wi >= H && wi < WIN - H && l >= S.out_lo && l < S.out_hi) {
#pragma unroll
            for (int f = 0; f < NF; f++) yout[at<LAYOUT>(f, l, S.ld)] = y[c][f];
        }
    }
#ifdef MARL_LAB_PHASE_CLOCK
    sb.phase_flush();
#endif
#ifdef MARL_LAB_CLOCK
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        marl_lab_clock[3 * blockIdx.x] = __builtin_amdgcn_s_memtime() - lab_t0;
        marl_lab_clock[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - lab_r0;
        marl_lab_clock[3 * blockIdx.x + 2] = lab_r0;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// The whole fixed-step loop of ONE large grid in ONE launch: a dataflow schedule over (level, tile) work items.
//
// A launch of rk4_fused_kernel is a global barrier: with waves that run ~20 us and only ~4.6 "rounds" of them per launch
// (N = 2^20: 18 728 waves on 4096 slots), the partially filled last round and the ramp-up leave the SIMDs with 3.3 instead
// of 4 resident waves on average (SQ counters) - and tile t of the next level only needs tiles t-1, t, t+1 of
// this one.  So: `levels` x `tiles` work items (a tile = the V cells one window writes, a level = NSTEPS steps), handed out in
// level-major order by an atomic counter to a grid of resident workgroups; an item waits until the three tiles it reads have
// published the previous level, computes exactly what the workgroup of the per-level launch computes (bit-identical results),
// and publishes its tile.  Ping-pong between two buffers is safe: neighbours can never be more than one level apart.
//   * Progress: an item only waits for items with smaller numbers, which were handed out earlier to workgroups that are
//     running - the smallest unfinished item never waits.  Every wait is bounded (STREAM_SPIN_LIMIT polls, ~1 s): on expiry,
//     or when any workgroup has raised the sticky flag, the workgroup raises it and leaves; the host reports the error.
//   * Coherence: tiles cross XCDs (one L2 each).  A release fence at agent scope writes back the XCD's whole L2 (measured 4x
//     slower, profiles/r02_lab_rk45_fused_control.log); instead the state itself moves with agent-scope (sc1: write-through /
//     L2-bypassing) stores and loads, and a tile is published - done[tile] = level_base + level + 1, also sc1 - only after every thread's
//     stores have been acknowledged (s_waitcnt vmcnt(0) + barrier).  The path is bound by the fp64 VALU, not by memory:
//     80 MB per level through the Infinity Fabric instead of the L2s costs nothing measurable.
// Nothing is reset between launches (a call is ONE launch, no memset in front of it): queue[0], the item counter, and done[t], the
// levels tile t has published, only ever grow; a launch is told where they stand (item_base: every earlier launch took its
// items plus one failing grab per workgroup; level_base: the levels of all earlier launches) and compares wrap-safely.  A stale
// value can only be too SMALL - it makes a reader wait, never pass.  sticky: raised by a workgroup that gives up; checked by
// waiting workgroups, never cleared by the device: the host looks at it at its next synchronisation point (marl_synchronize)
// and then resets counters and bases.
// ---------------------------------------------------------------------------------------------
constexpr unsigned STREAM_SPIN_LIMIT = 1u << 19;

template <int BLK, int LAYOUT, int NSTEPS, bool VD = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(4, 8)))
rk4_stream_kernel(double* bufA, double* bufB, const DevConsts* __restrict__ consts, Slab S, double dt, unsigned levels, unsigned tiles,
                  unsigned* queue, unsigned* done, unsigned* sticky, unsigned item_base, unsigned level_base, double* bufC = nullptr)
{
    // bufC != NULL (an ODD number of levels >= 3 in this launch): A -> B, then B <-> C, the last level back into A - the result lands in the
    // caller's buffer and the whole-state copy that an odd level count used to need afterwards (42 MB at N = 2^20: 12 us of a 20-step
    // call) goes away.  The write-after-read distance is that of the two-buffer scheme (a level writes what the level before reads, and
    // done[tile +- 1] orders it); the last level writes A, which only level 0 reads - five or more tiles of dependency away.
    constexpr int CPT = 1;
    constexpr int H = 4 * NSTEPS;
    constexpr int V = BLK - 2 * H;
    using SB = StencilBlock<BLK, CPT, true, VD>;
    __shared__ double lds[SB::LDS_DOUBLES];
    __shared__ unsigned s_item, s_abort;
    const DevConsts& C = consts[0];
    SB sb(lds, 0, consts);   // tables once per workgroup (barrier inside)
    const unsigned total = levels * tiles;   // (host: < 2^31)

#ifdef MARL_LAB_BROKEN_STREAM_LATCH   // tests/test_stream_isa.py only: the round-2 loop shape whose code the invariant checker must REJECT
    if (threadIdx.x == 0) {
        s_item = __hip_atomic_fetch_add(&queue[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - item_base;
        s_abort = 0;
    }
    while (true) {
        __syncthreads();
#else
    while (true) {
        // (barrier 1: every wave has left the previous item - s_item / s_abort may be rewritten.  It also separates the
        // thread-0 block below from the one at the end of the loop body: without it the two are fused across the back-edge
        // into a private loop of lane 0, whose wave then arrives at the barriers twice per item - seen with NSTEPS = 1: hang.
        // Taking the next item at the END of the body instead puts the LDS write of s_item into the loop latch, and the
        // compiler then emits the barrier at the loop header without the s_waitcnt lgkmcnt(0) in front of it: waves read the
        // previous s_item - seen as partly stale tiles.  tools/lab_src, profiles/r02_lab_rk4_stream.log.)
        __syncthreads();
        if (threadIdx.x == 0) {
            s_item = __hip_atomic_fetch_add(&queue[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - item_base;
            s_abort = 0;
        }
        __syncthreads();
#endif
        const unsigned item = s_item;
        if (item >= total) break;
        const unsigned level = item / tiles, tile = item - level * tiles;
        if (level > 0) {
            if (threadIdx.x < 3) {
                const int64_t t = (int64_t)tile - 1 + threadIdx.x;
                if (t >= 0 && t < (int64_t)tiles) {
                    unsigned spins = 0;
                    while ((int)(__hip_atomic_load(&done[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (level_base + level)) < 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > STREAM_SPIN_LIMIT ||
                            ((spins & 255u) == 0 && __hip_atomic_load(sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                            __hip_atomic_store(sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_abort = 1;
                            break;
                        }
                    }
                }
            }
            __syncthreads();
            if (s_abort) break;
        }
        const double* src = bufC ? (level == 0 ? bufA : ((level & 1u) ? bufB : bufC)) : ((level & 1u) ? bufB : bufA);
        double* dst = bufC ? (level + 1 == levels ? bufA : ((level & 1u) ? bufC : bufB)) : ((level & 1u) ? bufA : bufB);
        const int64_t l = S.out_lo + (int64_t)tile * V - H + threadIdx.x;
        const bool in = l >= 0 && l < S.n_buf;
        double y[CPT][NF];
#pragma unroll
        for (int f = 0; f < NF; f++)
            y[0][f] = in ? __hip_atomic_load(src + at<LAYOUT>(f, l, S.ld), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : C.bc[f];
        sb.set_window(l + S.goff);
        sb.reuse_live[0] = false;
        rk4_advance<NSTEPS>(sb, y, dt);
        if ((int)threadIdx.x >= H && (int)threadIdx.x < BLK - H && l >= S.out_lo && l < S.out_hi) {
#pragma unroll
            for (int f = 0; f < NF; f++) __hip_atomic_store(dst + at<LAYOUT>(f, l, S.ld), y[0][f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);   // this thread's stores have been acknowledged (vmcnt = expcnt = lgkmcnt = 0)
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __syncthreads();                 // ... and everybody else's (and everybody has read s_item, s_abort)
        if (threadIdx.x == 0) {
            __hip_atomic_store(&done[tile], level_base + level + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef MARL_LAB_BROKEN_STREAM_LATCH
            s_item = __hip_atomic_fetch_add(&queue[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - item_base;   // the item grab in the loop latch
            s_abort = 0;
#endif
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Dormand-Prince 5(4) coefficients: scipy/integrate/_ivp/rk.py:377-391
// ---------------------------------------------------------------------------------------------
namespace dp {
constexpr double A21 = 1.0 / 5;
constexpr double A31 = 3.0 / 40, A32 = 9.0 / 40;
constexpr double A41 = 44.0 / 45, A42 = -56.0 / 15, A43 = 32.0 / 9;
constexpr double A51 = 19372.0 / 6561, A52 = -25360.0 / 2187, A53 = 64448.0 / 6561, A54 = -212.0 / 729;
constexpr double A61 = 9017.0 / 3168, A62 = -355.0 / 33, A63 = 46732.0 / 5247, A64 = 49.0 / 176, A65 = -5103.0 / 18656;
constexpr double B1 = 35.0 / 384, B3 = 500.0 / 1113, B4 = 125.0 / 192, B5 = -2187.0 / 6784, B6 = 11.0 / 84;
constexpr double E1 = -71.0 / 57600, E3 = 71.0 / 16695, E4 = -71.0 / 1920, E5 = 17253.0 / 339200, E6 = -22.0 / 525, E7 = 1.0 / 40;
constexpr double SAFETY = 0.9, MIN_FACTOR = 0.2, MAX_FACTOR = 10.0;
}  // namespace dp


enum : int { ST_DONE = 0, ST_RUNNING = 1, ST_BUDGET = 2, ST_PAUSED = 3, ST_TOO_SMALL = -1 };

// Device-resident step controller: the scalar state of scipy's RungeKutta._step_impl
// (rk.py:111-176) plus the driver's event bookkeeping (ivp.py:673-694).  One per instance.
// Plain data; mirrored field for field by the ctypes structure in _abi.py.
struct Rk45Ctrl {
    double t, h_abs, t_bound, rtol, atol;
    double h_try, t_new;     // the attempt in flight (h_try = t_new - t, clipped at t_bound)
    double t_old, h_prev;    // last accepted step [t_old, t_old + h_prev] (dense output)
    double err_norm;         // error norm of the last attempt
    double pause_t;          // pause (ST_PAUSED) after the accepted step that reaches this time; +inf = never
    double g[7];             // monitors at the last accepted state (ivp.py:645, :694)
    double ev_first[7];      // first / latest sign change per monitor, located by linear interpolation
    double ev_last[7];       //   inside the bracketing step (the host refines with dense output + Brent)
    int64_t n_events[7];
    int64_t nfev, n_acc, n_rej, attempts, max_attempts;
    int64_t n_total;         // 5 * N of the global grid: size of the RMS norm (common.py:63-65)
    int32_t status;          // ST_*
    int32_t cur;             // which of the two state / FSAL buffers holds (y, f) at time t
    int32_t rejected;        // a rejection happened in the step in progress (rk.py:128,163-165)
    int32_t accepted_last;   // the last attempt was accepted
    int32_t pause_on_event;  // pause after an accepted step in which a monitor changed sign
    int32_t event_fired;     // set with ST_PAUSED when that was the reason
};

// Top of RungeKutta._step_impl: choose the step of the next attempt (rk.py:119-142).
__device__ __forceinline__ void rk45_prepare_attempt(Rk45Ctrl& c)
{
    const double min_step = 10.0 * fabs(nextafter(c.t, __builtin_inf()) - c.t);
    if (!c.rejected) {
        if (c.h_abs < min_step) c.h_abs = min_step;
    } else if (c.h_abs < min_step) {
        c.status = ST_TOO_SMALL;
        return;
    }
    if (c.max_attempts > 0 && c.attempts >= c.max_attempts) {
        c.status = ST_BUDGET;
        return;
    }
    double t_new = c.t + c.h_abs;
    if (t_new - c.t_bound > 0.0) t_new = c.t_bound;
    c.t_new = t_new;
    c.h_try = t_new - c.t;
    c.h_abs = fabs(c.h_try);
    c.attempts++;
}

// Bottom of _step_impl + the driver's per-step bookkeeping, in three pieces (rk45_finish_attempt = all three in order):
//   rk45_decide         error norm -> accept / reject and the step-size factor (rk.py:146-165)
//   rk45_events         the seven monitors of an accepted step against those of the step before (ivp.py:149-151, :673-694)
//   rk45_advance_status end of interval / pause, and the next attempt's step (rk.py:119-142)
// The one-workgroup sweep kernel runs decide + advance_status in EVERY wave (replicated scalar work, no publish barrier) and the
// event bookkeeping one barrier later, off the critical path.
// T: the log / exp tables when the caller has them in LDS: err^-0.2 is then exp(-0.2 log err) at a fifth of OCML pow's
// instructions, ~3 ulp instead of 1.
__device__ __forceinline__ bool rk45_decide(Rk45Ctrl& c, double sumsq, const Tables* T = nullptr)
{
    const double err = sqrt(sumsq) / sqrt((double)c.n_total);  // common.py:63-65
    c.err_norm = err;
    c.nfev += 6;
    double f;
    if (T && err > 1e-300 && err < 1e300) f = dp::SAFETY * fast_exp(-0.2 * fast_log(err, *T), *T);
    else f = dp::SAFETY * pow(err, -0.2);
    if (err < 1.0) {  // rk.py:149-161
        double factor = (err == 0.0) ? dp::MAX_FACTOR : ((f < dp::MAX_FACTOR) ? f : dp::MAX_FACTOR);
        if (c.rejected && !(factor < 1.0)) factor = 1.0;
        c.h_abs *= factor;
        c.accepted_last = 1;
        c.rejected = 0;
        c.n_acc++;
        c.t_old = c.t;
        c.h_prev = c.h_try;
        c.t = c.t_new;
        c.cur ^= 1;
        return true;
    }
    // rk.py:162-165; a NaN norm lands here with factor 0.2
    c.h_abs *= (f > dp::MIN_FACTOR) ? f : dp::MIN_FACTOR;
    c.accepted_last = 0;
    c.rejected = 1;
    c.n_rej++;
    return false;
}

// The same decision from the MEAN square error (sum / n) for the replicated controller of the one-workgroup sweep kernel, where
// every wave pays for these instructions: err < 1 <=> err^2 < 1, and 0.9 err^-0.2 = 0.9 exp(-0.1 log err^2) - no square roots, no
// division (inv_n = 1 / n_total, formed once), always through the LDS tables.  c.err_norm holds err^2 until the kernel leaves
// (rk45_lean_finish).  Against rk45_decide the factor differs in the last bits (as the table path does from OCML pow already).
__device__ __forceinline__ bool rk45_decide_lean(Rk45Ctrl& c, double sumsq, double inv_n, const Tables& T)
{
    const double e2 = sumsq * inv_n;
    c.err_norm = e2;
    c.nfev += 6;
    double f;
    if (e2 > 1e-300 && e2 < 1e300) f = dp::SAFETY * fast_exp(-0.1 * fast_log(e2, T), T);
    else f = dp::SAFETY * pow(sqrt(e2), -0.2);
    if (e2 < 1.0) {
        double factor = (e2 == 0.0) ? dp::MAX_FACTOR : ((f < dp::MAX_FACTOR) ? f : dp::MAX_FACTOR);
        if (c.rejected && !(factor < 1.0)) factor = 1.0;
        c.h_abs *= factor;
        c.accepted_last = 1;
        c.rejected = 0;
        c.n_acc++;
        c.t_old = c.t;
        c.h_prev = c.h_try;
        c.t = c.t_new;
        c.cur ^= 1;
        return true;
    }
    c.h_abs *= (f > dp::MIN_FACTOR) ? f : dp::MIN_FACTOR;
    c.accepted_last = 0;
    c.rejected = 1;
    c.n_rej++;
    return false;
}

// rk45_prepare_attempt with the minimum-step test (10 ulp(t)) behind a cheap sufficient condition: h_abs > |t| 2^-48 > 10 ulp(t)
// holds on every attempt of a healthy run, and nextafter costs ~20 instructions in every wave
__device__ __forceinline__ void rk45_prepare_attempt_lean(Rk45Ctrl& c)
{
    if (!(c.h_abs > fabs(c.t) * 0x1p-48) || c.t == 0.0) {   // (t = 0: nextafter gives the smallest subnormal)
        rk45_prepare_attempt(c);
        return;
    }
    if (c.max_attempts > 0 && c.attempts >= c.max_attempts) {
        c.status = ST_BUDGET;
        return;
    }
    double t_new = c.t + c.h_abs;
    if (t_new - c.t_bound > 0.0) t_new = c.t_bound;
    c.t_new = t_new;
    c.h_try = t_new - c.t;
    c.h_abs = fabs(c.h_try);
    c.attempts++;
}

// Monitor e of the step just accepted ([c.t_old, c.t], size c.h_prev): gn against the stored c.g[e]; returns 1 on a sign change.
__device__ __forceinline__ int rk45_event_one(Rk45Ctrl& c, int e, double gn)
{
    const double go = c.g[e];
    int fired = 0;
    const bool up = go <= 0.0 && gn >= 0.0, down = go >= 0.0 && gn <= 0.0;  // ivp.py:149-151
    if (up || down) {
        const double d = go - gn;
        const double tc = (d != 0.0) ? c.t_old + c.h_prev * (go / d) : c.t;
        if (c.n_events[e] == 0) c.ev_first[e] = tc;
        c.ev_last[e] = tc;
        c.n_events[e]++;
        fired = 1;
    }
    c.g[e] = gn;
    return fired;
}

// monitor e (the reference's order, Evolve_scenario.py:107-109) from a reduction record {sum, min y, min CA, min CC, min U, max CA+CC, max Phi, max W}
__device__ __forceinline__ double rk45_monitor_of_record(const double* rec, int e)
{
    const int slot = (e < 3) ? e + 1 : (e == 3 ? 5 : (e == 4 ? 6 : (e == 5 ? 4 : 7)));
    return (e == 3 || e == 4) ? rec[slot] - 1.0 : rec[slot];
}

__device__ __forceinline__ int rk45_events(Rk45Ctrl& c, const double (&rec)[NQ])
{
    int fired = 0;
#pragma unroll
    for (int e = 0; e < 7; e++) fired |= rk45_event_one(c, e, rk45_monitor_of_record(rec, e));
    return fired;
}

__device__ __forceinline__ void rk45_advance_status(Rk45Ctrl& c, bool accepted, int fired)
{
    if (accepted) {
        if (c.t - c.t_bound >= 0.0) {  // base.py:203-204
            c.status = ST_DONE;
        } else if (c.t >= c.pause_t || (fired && c.pause_on_event)) {
            c.status = ST_PAUSED;
            c.event_fired = fired;
        }
    }
    if (c.status == ST_RUNNING) rk45_prepare_attempt(c);
}

// rk45_advance_status after an ACCEPTED step of a run that does not pause on monitor sign changes (the events are then pure
// bookkeeping and may follow later)
__device__ __forceinline__ void rk45_advance_status_no_events(Rk45Ctrl& c)
{
    if (c.t - c.t_bound >= 0.0) c.status = ST_DONE;
    else if (c.t >= c.pause_t) c.status = ST_PAUSED;
    if (c.status == ST_RUNNING) rk45_prepare_attempt(c);
}

// rec = {sum (err/scale)^2, monitors of y_new}.
__device__ __forceinline__ void rk45_finish_attempt(Rk45Ctrl& c, const double (&rec)[NQ], const Tables* T = nullptr)
{
    const bool accepted = rk45_decide(c, rec[0], T);
    const int fired = accepted ? rk45_events(c, rec) : 0;
    rk45_advance_status(c, accepted, fired);
}

// rec0: monitors record of y(t0).  RungeKutta.__init__ (rk.py:94-102) + ivp.py:645.
__global__ void rk45_init_kernel(Rk45Ctrl* ctrl, const double* __restrict__ rec0, double t0, double t1, double first_step,
                                 double rtol, double atol, int64_t n_total, int64_t max_attempts, int32_t cur)
{
    Rk45Ctrl c = {};
    c.t = t0; c.t_bound = t1; c.h_abs = first_step; c.rtol = rtol; c.atol = atol;
    c.t_old = t0; c.pause_t = __builtin_inf();
    c.n_total = n_total; c.max_attempts = max_attempts; c.cur = cur;
    c.nfev = 1;
    c.status = (t0 == t1) ? ST_DONE : ST_RUNNING;  // base.py:189-194
    const double* r = rec0 + (int64_t)blockIdx.x * NQ;
    const double g0[7] = {r[1], r[2], r[3], r[5] - 1.0, r[6] - 1.0, r[4], r[7]};
    for (int e = 0; e < 7; e++) c.g[e] = g0[e];
    if (c.status == ST_RUNNING) rk45_prepare_attempt(c);
    ctrl[blockIdx.x] = c;
}

// Resume a paused / budget-stopped controller (host sets new pause_t / max_attempts first).
__global__ void rk45_resume_kernel(Rk45Ctrl* ctrl, double pause_t, int64_t max_attempts)
{
    Rk45Ctrl& c = ctrl[blockIdx.x];
    c.pause_t = pause_t;
    c.max_attempts = max_attempts;
    if (c.status == ST_PAUSED || c.status == ST_BUDGET) {
        c.status = ST_RUNNING;
        c.event_fired = 0;
        rk45_prepare_attempt(c);
    }
}

// Combine `nrec` reduction records (per-block partials on one GPU, or one record per rank after the
// all-gather of a domain-decomposed run) in index order and finish the attempt.
// One workgroup of 512 threads: with thousands of block records the reduction is latency-bound, so it is spread
// over many lanes (4 300 records at N = 2^20: ~8 per thread).
constexpr int CONTROL_THREADS = 512;
__global__ void __launch_bounds__(CONTROL_THREADS) rk45_control_kernel(const double* __restrict__ recs, int64_t nrec, Rk45Ctrl* ctrl)
{
    __shared__ double scratch[NQ * CONTROL_THREADS];
#ifdef MARL_CTL_CLOCK
    const unsigned long long c0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int32_t status = ctrl->status;
    double q[NQ];
    monitors_init(q);
    for (int64_t b = threadIdx.x; b < nrec; b += CONTROL_THREADS) {
        double o[NQ];
#pragma unroll
        for (int j = 0; j < NQ; j++) o[j] = recs[b * NQ + j];
#pragma unroll
        for (int j = 0; j < NQ; j++) q[j] = (j == 0) ? q[j] + o[j] : (j <= NQMIN ? nanmin(q[j], o[j]) : nanmax(q[j], o[j]));
    }
    if (status != ST_RUNNING) return;   // (read first, tested last: the record loads do not wait for it)
#ifdef MARL_CTL_CLOCK
    const unsigned long long c1 = __builtin_amdgcn_s_memrealtime();
#endif
    block_reduce<CONTROL_THREADS, NQ, NQMIN>(q, scratch);
#ifdef MARL_CTL_CLOCK
    const unsigned long long c2 = __builtin_amdgcn_s_memrealtime();
#endif
    if (threadIdx.x == 0) {
        Rk45Ctrl c = *ctrl;
        rk45_finish_attempt(c, q);
        *ctrl = c;
#ifdef MARL_CTL_CLOCK
        const unsigned long long c3 = __builtin_amdgcn_s_memrealtime();
        if (c.attempts == 300) printf("CTLCLOCK load %llu reduce %llu finish %llu (x10 ns)\n", c1 - c0, c2 - c1, c3 - c2);
#endif
    }
}

// One Dormand-Prince attempt on the per-thread cells: k1 given; returns y_new in `yn`, f(y_new) in `k7`
// and the error-estimate numerator  sum_j E_j K_j  in `esum` (rk_step, rk.py:61-69; _estimate_error :103-104).
// With DENSE, `esum` instead receives  sum_j w_j K_j  for the dense-output weights w_j = sum_m P[j][m] x^(m+1)
// (RkDenseOutput, rk.py:560-574; w_2 = 0 because row 2 of P is zero).
struct DenseWeights { double w[7]; };

// PARK = number of fields (0..5) of the step's first state y that are NOT held in registers but in this thread's LDS column
// `pk` (pk[(c*NF + f) * BLK], f < PARK) and re-read where a stage state is formed - up to five live doubles less across every
// evaluation (kernels that sit just above a register cap: 128 VGPRs = 4 waves per SIMD).
template <int BLK, int CPT, bool DENSE = false, class SB = StencilBlock<BLK, CPT>, int PARK = 0>
__device__ __forceinline__ void dp45_attempt(SB& sb, double h,
                                             const double (&y)[CPT][NF], const double (&k1)[CPT][NF],
                                             double (&yn)[CPT][NF], double (&k7)[CPT][NF], double (&esum)[CPT][NF],
                                             PointAux (&aux)[CPT], const DenseWeights& dw = DenseWeights{}, const double* pk = nullptr)
{
    const double e1 = DENSE ? dw.w[0] : dp::E1, e3 = DENSE ? dw.w[2] : dp::E3, e4 = DENSE ? dw.w[3] : dp::E4;
    const double e5 = DENSE ? dw.w[4] : dp::E5, e6 = DENSE ? dw.w[5] : dp::E6, e7 = DENSE ? dw.w[6] : dp::E7;
    // K1..K4 are folded into the partial sums of everything that still needs them as soon as K4 exists, so that at
    // most five vectors (instead of seven) are live across an evaluation.  The order of every
    // sum is the left-to-right order of np.dot(K[:s].T, a[:s]) (rk.py:61-69).
    double ys[CPT][NF], k2[CPT][NF], k3[CPT][NF], k4[CPT][NF], kk[CPT][NF], s6[CPT][NF], bn[CPT][NF];
#define MARL_CELLS _Pragma("unroll") for (int c = 0; c < CPT; c++) _Pragma("unroll") for (int f = 0; f < NF; f++)
#define MARL_Y(c, f) ((f) < PARK ? pk[((c) * NF + (f)) * BLK] : y[c][f])
    MARL_CELLS ys[c][f] = MARL_Y(c, f) + (k1[c][f] * dp::A21) * h;
    sb.template eval<TR_FILL>(ys, k2, aux);
    MARL_CELLS ys[c][f] = MARL_Y(c, f) + (k1[c][f] * dp::A31 + k2[c][f] * dp::A32) * h;
    sb.template eval<TR_REUSE>(ys, k3, aux);
    MARL_CELLS ys[c][f] = MARL_Y(c, f) + (k1[c][f] * dp::A41 + k2[c][f] * dp::A42 + k3[c][f] * dp::A43) * h;
    sb.template eval<TR_REUSE>(ys, k4, aux);
    MARL_CELLS {
        ys[c][f] = MARL_Y(c, f) + (k1[c][f] * dp::A51 + k2[c][f] * dp::A52 + k3[c][f] * dp::A53 + k4[c][f] * dp::A54) * h;
        s6[c][f] = k1[c][f] * dp::A61 + k2[c][f] * dp::A62 + k3[c][f] * dp::A63 + k4[c][f] * dp::A64;
        bn[c][f] = k1[c][f] * dp::B1 + k3[c][f] * dp::B3 + k4[c][f] * dp::B4;
        esum[c][f] = k1[c][f] * e1 + k3[c][f] * e3 + k4[c][f] * e4;
    }
    sb.template eval<TR_REUSE>(ys, kk, aux);   // K5
    MARL_CELLS {
        ys[c][f] = MARL_Y(c, f) + (s6[c][f] + kk[c][f] * dp::A65) * h;
        bn[c][f] = bn[c][f] + kk[c][f] * dp::B5;
        esum[c][f] = esum[c][f] + kk[c][f] * e5;
    }
    sb.template eval<TR_REUSE>(ys, kk, aux);   // K6
    MARL_CELLS {
        yn[c][f] = MARL_Y(c, f) + h * (bn[c][f] + kk[c][f] * dp::B6);
        esum[c][f] = esum[c][f] + kk[c][f] * e6;
    }
    sb.template eval<TR_REUSE>(yn, k7, aux);
    MARL_CELLS esum[c][f] = esum[c][f] + k7[c][f] * e7;
#undef MARL_Y
#undef MARL_CELLS
}

// (err/scale)^2 of one component; scale = atol + max(|y|, |y_new|) * rtol  (rk.py:146-147)
__device__ __forceinline__ double dp45_err2(double esum, double h, double y, double yn, double rtol, double atol)
{
    const double ay = fabs(y), an = fabs(yn);
    const double scale = atol + ((ay > an || ay != ay) ? ay : an) * rtol;
    const double e = esum * h * rcp_nr(scale);
    return e * e;
}

// ---------------------------------------------------------------------------------------------
// Reduction of a tile's eight quantities over its 256 threads with ONE barrier (round 4; the round-1 block_reduce takes four).
// Each thread writes its q[j] into column j of an LDS area [8][BLK]; after the barrier HALF-WAVE j (32 lanes) reduces column j:
// lane i combines rows i, i + 32, ..., i + 224 in that order, then a 5-step butterfly over the 32 lanes - a fixed order.
// Quantity 0 is a sum (NaN-propagating), 1..NQMIN minima, the rest maxima, carried NEGATED through the columns so that every
// half-wave but the first runs the same v_min_f64 (finite values: see monitors_accumulate<true>).  Result in lane 0 of half-wave j
// (threads 0, 32, ..., 224), maxima still negated; other lanes hold garbage.
// Where the columns live when a Dormand-Prince attempt has just finished (no barrier needed in FRONT of the writes): columns 0..6 =
// the four cache-slot columns + the three park columns (contiguous; both hold thread-PRIVATE data that the thread has finished
// with), column 7 = field 0 of the parity-0 edge buffer (last read in the fifth evaluation; every wave has passed the sixth's
// barrier).  The caller must put a barrier between this function's reads and the next writes to those areas.
// ---------------------------------------------------------------------------------------------
template <int BLK>
__device__ __forceinline__ double* attempt_reduce_column(double* lds, int edge_doubles, int j)
{
    return j < 7 ? lds + edge_doubles + TABLE_DOUBLES + j * BLK : lds;
}
template <int BLK, class SB>
__device__ __forceinline__ double tile_reduce_halfwaves(const double (&q)[NQ], double* lds, int tid)
{
    static_assert(BLK == 256 && NQ == 8, "eight quantities, eight half-waves");
    static_assert(PC_LDS_SLOTS >= 4 && SB::CACHE, "columns 0..6 = the cache-slot columns (four at least) + three park columns");
#pragma unroll
    for (int j = 0; j < NQ; j++) attempt_reduce_column<BLK>(lds, SB::EDGE_DOUBLES, j)[tid] = (j > NQMIN) ? -q[j] : q[j];
    __syncthreads();
    const int j = tid >> 5, i = tid & 31;
    const bool is_sum = j == 0;
    const double* col = (j < 7 ? lds + SB::EDGE_DOUBLES + TABLE_DOUBLES + j * BLK : lds) + i;
    double a = col[0];
#pragma unroll
    for (int m = 1; m < BLK / 32; m++) {
        const double o = col[32 * m];
        a = is_sum ? a + o : __builtin_fmin(a, o);
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {
        const double o = __shfl_xor(a, off, 64);
        a = is_sum ? a + o : __builtin_fmin(a, o);
    }
    return a;
}

// ---------------------------------------------------------------------------------------------
// Fused adaptive attempt for ONE large grid (or one slab of a domain-decomposed grid): reads
// (y, f) from buffer `cur`, writes (y_new, f_new) into the other buffer and one reduction record per
// block; rk45_control_kernel then accepts (flips `cur`) or rejects.  FSAL: f_new becomes K1.
// ---------------------------------------------------------------------------------------------
// One cell per thread in 256-thread blocks: 4 waves per SIMD (<= 128 VGPRs) with ATTEMPT_PARK fields of y parked in LDS
// (4 blocks per CU leave 9 KB of LDS per block beside the edge buffers, tables and cache slots; three parked doubles = 6 KB suffice: 128 VGPRs, no scratch).
#ifndef MARL_ATTEMPT_PARK
#define MARL_ATTEMPT_PARK 3
#endif
template <int BLK, int CPT>
constexpr int ATTEMPT_PARK = (BLK == 256 && CPT == 1) ? MARL_ATTEMPT_PARK : 0;

template <int BLK, int CPT, int LAYOUT, bool VD = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu((BLK == 256 && CPT == 1 && MARL_ATTEMPT_PARK > 0) ? 4 : 1, 8)))
rk45_attempt_kernel(double* __restrict__ Y0, double* __restrict__ Y1,
                                                           double* __restrict__ F0, double* __restrict__ F1,
                                                           const DevConsts* __restrict__ consts, Slab S,
                                                           const Rk45Ctrl* __restrict__ ctrl, double* __restrict__ part)
{
    constexpr int H = 6;
    constexpr int WIN = BLK * CPT;
    constexpr int V = WIN - 2 * H;
    constexpr int PARK = ATTEMPT_PARK<BLK, CPT>;
    using SB = StencilBlock<BLK, CPT, true, VD>;
    __shared__ double lds[SB::LDS_DOUBLES + PARK * CPT * BLK];
    if (ctrl->status != ST_RUNNING) return;
    const DevConsts& C = consts[0];
    const int cur = ctrl->cur;
    const double h = ctrl->h_try, rtol = ctrl->rtol, atol = ctrl->atol;
    const double* yin = cur ? Y1 : Y0;
    const double* fin = cur ? F1 : F0;
    double* yout = cur ? Y0 : Y1;
    double* fout = cur ? F0 : F1;

    const int64_t w0 = S.out_lo + (int64_t)blockIdx.x * V - H;
    const int64_t l0 = w0 + (int64_t)threadIdx.x * CPT;

    double y[CPT][NF], k1[CPT][NF], yn[CPT][NF], k7[CPT][NF], esum[CPT][NF];
    PointAux aux[CPT];
    load_cells<CPT, LAYOUT>(yin, l0, S, C, y);              // in flight while the tables are staged
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t l = l0 + c;
        const bool in = l >= 0 && l < S.n_buf;
#pragma unroll
        for (int f = 0; f < NF; f++) k1[c][f] = in ? fin[at<LAYOUT>(f, l, S.ld)] : 0.0;
    }
    SB sb(lds, l0 + S.goff, consts);   // (table copy + barrier inside)
    double* pk = lds + SB::LDS_DOUBLES + threadIdx.x;
    if constexpr (PARK > 0) {
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int f = 0; f < PARK; f++) pk[(c * NF + f) * BLK] = y[c][f];
    }
    dp45_attempt<BLK, CPT, false, SB, PARK>(sb, h, y, k1, yn, k7, esum, aux, DenseWeights{}, pk);

    double q[NQ];
    monitors_init(q);
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int wi = threadIdx.x * CPT + c;
        const int64_t l = l0 + c;
        if (wi >= H && wi < WIN - H && l >= S.out_lo && l < S.out_hi) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                yout[at<LAYOUT>(f, l, S.ld)] = yn[c][f];
                fout[at<LAYOUT>(f, l, S.ld)] = k7[c][f];
                q[0] += dp45_err2(esum[c][f], h, f < PARK ? pk[(c * NF + f) * BLK] : y[c][f], yn[c][f], rtol, atol);
            }
            monitors_accumulate<true>(q, yn[c], aux[c].U, aux[c].W);   // (the controller reads the extrema of ACCEPTED attempts only: all finite)
        }
    }
#ifndef MARL_LAB_ATTEMPT_BLOCK_REDUCE   // (kernel-lab switch: the round-1 four-barrier reduction)
    if constexpr (BLK == 256 && CPT == 1 && PARK >= 3 && SB::CACHE && PC_LDS_SLOTS >= 4) {
        // one barrier instead of four: columns in LDS areas this thread / every wave has finished with (tile_reduce_halfwaves)
        const double r = tile_reduce_halfwaves<BLK, SB>(q, lds, threadIdx.x);
        if ((threadIdx.x & 31) == 0) {
            const int j = threadIdx.x >> 5;
            part[(int64_t)blockIdx.x * NQ + j] = (j > NQMIN) ? -r : r;
        }
    } else
#endif
    {
        block_reduce<BLK, NQ, NQMIN, true>(q, lds);   // the edge-exchange buffers are free now
        if (threadIdx.x == 0) {
#pragma unroll
            for (int j = 0; j < NQ; j++) part[(int64_t)blockIdx.x * NQ + j] = q[j];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The whole adaptive loop of ONE large grid in ONE launch (round 4): resident workgroups, one device-wide barrier per attempt.
//
// What a launch per attempt costs (rk45_attempt_kernel, N = 2^20: 63 us + two small launches): every workgroup pays a prologue
// (kernel arguments -> the controller's cur / h -> the state loads that depend on cur -> tables into LDS -> barrier), a four-barrier
// block reduction, and the dispatcher's turnover of its slot; 4 297 tiles on 1 024 slots leave a fifth, 20 % filled round.  Here a
// grid of G resident workgroups (4 per CU) stages the tables once, keeps (h, cur, status) in scalar registers and walks a STATIC
// set of tiles per attempt (tile = g, g + G, ...; the remainder round is SPREAD over the grid, see extra_tile) with the loads of the
// next tile in flight while the present one is reduced; the attempt ends in ONE barrier across the grid:
//   * each workgroup folds its tiles' records (tile order) into one group record: the sum of squares into gsum[g], the seven
//     extrema into gmon[g][7] (sc1 stores), waits for all its stores to be acknowledged, and takes a ticket (one agent-scope atomic);
//   * the workgroup that takes the LAST ticket of the attempt adds the G sums in a fixed order (thread t: groups t, t + BLK, ...;
//     wave butterflies; waves in order - deterministic, independent of who is last), takes scipy's accept / reject and step-size
//     decision (rk45_decide, the function the control kernel runs) and PUBLISHES what the next attempt needs - h, cur, status, the
//     barrier's number - as one 16-byte record replicated over PUB_COPIES cache lines; only then, off the critical path, it reduces
//     the extrema of an accepted step, does the event bookkeeping and writes the controller back (pause_on_event: the events
//     decide the status, so they come first);
//   * everybody polls its copy of the record (bounded; a time-out raises the sticky flag and every workgroup leaves).
// Coherence across the 8 XCDs as in rk4_stream_kernel: state, records and the controller move with agent-scope (sc1) loads /
// stores; a ticket is taken only after s_waitcnt vmcnt(0) + barrier.  Needs all G workgroups resident at once (the host sizes G by
// hipOccupancyMaxActiveBlocksPerMultiprocessor); counters are never reset between launches (arrive_base / epoch_base say where
// they stand).  The published record carries a 24-bit hash of h beside the barrier number, so a torn read cannot pass.
// ---------------------------------------------------------------------------------------------
// a value that is the same in every lane, moved into scalar registers (readfirstlane)
__device__ __forceinline__ double to_sgpr(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ int32_t to_sgpr(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t to_sgpr(int64_t v)
{
    return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int32_t)((uint64_t)v >> 32)) << 32) |
                     (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v));
}
constexpr int MON_STRIDE = 16;          // doubles between the grid-wide extrema (one 128-byte line each)
constexpr int MON_OFFSET = 4096;        // grec: [0, MON_OFFSET) the workgroups' sums, then [2][NQ - 1][MON_STRIDE] extrema
constexpr int GREC_DOUBLES = MON_OFFSET + 2 * (NQ - 1) * MON_STRIDE;
constexpr int PUB_COPIES = 32;          // replicas of the published record, PUB_STRIDE words apart (pollers spread over channels)
constexpr int PUB_STRIDE = 32;          // 64-bit words (256 B)
struct Rk45Stream {
    unsigned arrive;                    // tickets taken since the words were zeroed
    unsigned sticky;                    // raised by a workgroup that gave up waiting
    unsigned pad0[30];
    unsigned grab;                      // remainder-round tiles handed out (own cache line)
    unsigned pad1[31];
    unsigned long long pub[PUB_COPIES * PUB_STRIDE];   // copy k: pub[k * PUB_STRIDE + 0] = bits of h_try, [+ 1] = hash24(h) << 40 | epoch << 8 | (status + 1) << 1 | cur
};
__device__ __forceinline__ unsigned long long pub_word(unsigned long long hbits, unsigned epoch, int status, int cur)
{
    const unsigned long long hash = (hbits ^ (hbits >> 24) ^ (hbits >> 48)) & 0xffffffull;
    return (hash << 40) | ((unsigned long long)epoch << 8) | (unsigned long long)(((status + 1) & 7) << 1) | (unsigned long long)(cur & 1);
}
// The tiles of an attempt: `full` = tiles / G rounds are STATIC (workgroup g: tiles g, g + G, ...; the next tile's loads in flight
// while the present one is reduced); the R = tiles mod G tiles of the remainder round are handed out by an atomic counter to whoever
// has finished its static tiles.  The SIMDs arbitrate oldest-first, so the four workgroups of a CU finish their static tiles one
// after the other (measured at N = 2^20: after 31 ... 65 us) and the remainder goes to the early finishers, about one per CU - a
// static owner (every ~G/R-th workgroup) left some CUs with 18 tiles and others with 16 (profiles/r04_lab_rk45_stream.log).
// Determinism: a static tile's sum of squares goes into its workgroup's sum (tile order), a remainder tile's into its own slot
// gextra[e]; the barrier's last workgroup adds G + R numbers in index order - the same numbers whoever computed them; extrema are
// exact under any order.  The counter is never reset: every attempt takes exactly R successful and G failing grabs.
#ifdef MARL_LAB_CLOCK45  // kernel-lab diagnostic build only: s_memrealtime (100 MHz) stamps per (attempt < 64, workgroup < 2048): attempt start, tiles done, ticket taken, decision seen
__device__ unsigned long long marl_lab_clock45[64 * 2048 * 4 + 64 * 8];   // (+ the last arriver's phases: [64][8])
#define MARL_STAMP45(k) do { if (threadIdx.x == 0 && a < 64 && g < 2048) marl_lab_clock45[(a * 2048 + g) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MARL_STAMP45L(k) do { if (threadIdx.x == 0 && a < 64) marl_lab_clock45[64 * 2048 * 4 + a * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
__device__ unsigned long long marl_lab_clock45t[2048 * 8 * 8];   // tiles of attempt 20: [g][tile index < 8][phase < 8]
#define MARL_STAMP45T(k) do { if (threadIdx.x == 0 && a == 20 && g < 2048 && ti < 8 && tile < tiles) marl_lab_clock45t[(g * 8 + ti) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MARL_STAMP45(k) do { } while (0)
#define MARL_STAMP45L(k) do { } while (0)
#define MARL_STAMP45T(k) do { } while (0)
#endif
#ifdef MARL_LAB_STREAM_PLAIN   // kernel-lab cost probe ONLY (results are wrong across XCDs): the state through the non-coherent L2s
#define MARL_STREAM_LOAD(p) (*(p))
#define MARL_STREAM_STORE(p, v) (*(p) = (v))
#else
#define MARL_STREAM_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define MARL_STREAM_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#ifndef MARL_STREAM_PARK   // fields of y parked in LDS (dp45_attempt): 4 leave LDS for exactly four workgroups per CU (40 448 of 40 960 bytes)
#define MARL_STREAM_PARK 4
#endif
template <int BLK, int LAYOUT, bool VD = false, bool DD = false>
__global__ void __launch_bounds__(BLK) __attribute__((amdgpu_waves_per_eu(4, 8)))
rk45_stream_kernel(double* Y0, double* Y1, double* F0, double* F1, const DevConsts* __restrict__ consts, Slab S, Rk45Ctrl* ctrl,
                   double* grec, Rk45Stream* sync, unsigned tiles, unsigned arrive_base, unsigned epoch_base, unsigned grab_base, unsigned max_attempts,
                   double* send = nullptr, int halo = 0)
{
    // DD (one slab of a domain-decomposed grid; max_attempts = 1, all three bases 0; option dd_stream): ONE attempt per launch, and the
    // barrier's last workgroup, instead of deciding, writes the rank's message [record (8) | lower strip | upper strip] (what
    // reduce_chunks_kernel + slab_reduce_pack_kernel did) and re-arms the counters; the all-gather and slab_unpack_control_kernel
    // follow, and the next launch reads the common decision out of *ctrl.
    constexpr int CPT = 1;
    constexpr int H = 6;
    constexpr int V = BLK - 2 * H;
    constexpr int PARK = MARL_STREAM_PARK;
    constexpr int NW = BLK / 64;
    constexpr int CTRL_WORDS = (int)(sizeof(Rk45Ctrl) / 8);
    static_assert(sizeof(Rk45Ctrl) % 8 == 0 && CTRL_WORDS <= BLK, "the controller moves as 64-bit words, one per thread");
    static_assert(PARK >= 3, "the reduction columns of a tile use three park columns");
    using SB = StencilBlock<BLK, CPT, true, VD>;
    __shared__ double lds[SB::LDS_DOUBLES + PARK * CPT * BLK];
    __shared__ double s_acc[NQ];         // this workgroup's record of the attempt (maxima negated), slot j touched by thread 32 j only
    __shared__ double s_wsum[NW];
    __shared__ double s_rec[NQ];         // last workgroup of a barrier: the grid's record {-, seven extrema of y_new}
    __shared__ Rk45Ctrl sc;
    __shared__ unsigned long long s_pub[2];
    __shared__ int s_last, s_abort;
    __shared__ unsigned s_next;
    SB sb(lds, 0, consts);   // tables once per workgroup (barrier inside)
    const unsigned G = gridDim.x, g = blockIdx.x;
    const unsigned n_full = tiles / G, R = tiles - n_full * G;         // n_full >= 1: the host launches G <= tiles workgroups
    double* const gsum = grec;                      // [G + R]: the workgroups' sums of squares over their static tiles, then the remainder tiles' sums
    double* const gmon = grec + MON_OFFSET;         // [2][NQ - 1][MON_STRIDE]: the grid's seven extrema (maxima negated) by attempt parity, +inf when idle
    // the controller as the previous launch (or rk45_init / rk45_resume) left it
    int cur = ctrl->cur, status = ctrl->status;
    double h = ctrl->h_try;
    const double rtol = ctrl->rtol, atol = ctrl->atol;
    const int pause_on_event = ctrl->pause_on_event;
    unsigned long long* const my_pub = sync->pub + (g % PUB_COPIES) * PUB_STRIDE;


    for (unsigned a = 0; a < max_attempts && status == ST_RUNNING; a++) {
        const double* yin = cur ? Y1 : Y0;
        const double* fin = cur ? F1 : F0;
        double* yout = cur ? Y0 : Y1;
        double* fout = cur ? F0 : F1;
        MARL_STAMP45(0);
#ifdef MARL_LAB_CLOCK45
        if (threadIdx.x == 0 && g == 0 && (a == 8 || a == 56)) {   // shader clock (s_memtime) against the 100 MHz reference: the clock this kernel runs at
            marl_lab_clock45t[2048 * 8 * 8 - 4 + (a == 8 ? 0 : 2)] = __builtin_amdgcn_s_memtime();
            marl_lab_clock45t[2048 * 8 * 8 - 4 + (a == 8 ? 1 : 3)] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        double y[CPT][NF], k1[CPT][NF];
        // (a thread's cell of tile t: local index l = out_lo + t V - H + tid; lanes outside the buffer - beyond a physical boundary, where the
        // ghost-cell rules apply and nothing reads them - load a copy of the edge cell: no predicate, no branch, finite values)
#define MARL_LOAD_TILE(t, tid_)                                                                                                        \
        do {                                                                                                                           \
            int64_t l_ = S.out_lo + (int64_t)(t) * V - H + (tid_);                                                                     \
            l_ = l_ < 0 ? 0 : (l_ < S.n_buf ? l_ : S.n_buf - 1);                                                                       \
            _Pragma("unroll") for (int f = 0; f < NF; f++)                                                                             \
                y[0][f] = MARL_STREAM_LOAD(yin + at<LAYOUT>(f, l_, S.ld));                                                             \
            _Pragma("unroll") for (int f = 0; f < NF; f++)                                                                             \
                k1[0][f] = MARL_STREAM_LOAD(fin + at<LAYOUT>(f, l_, S.ld));                                                            \
        } while (0)
        {
            int tid0 = threadIdx.x;
            asm volatile("" : "+v"(tid0));
            MARL_LOAD_TILE(g, tid0);
        }
        if ((threadIdx.x & 31) == 0) {   // thread 32 j owns s_acc[j]
            const int j = threadIdx.x >> 5;
            s_acc[j] = (j == 0) ? 0.0 : __builtin_inf();
        }
        const unsigned grab0 = grab_base + a * (R + G);
        for (unsigned ti = 0;; ti++) {
            unsigned tile = g + ti * G, extra = 0;
            MARL_STAMP45T(0);
            int tid = threadIdx.x;
            asm volatile("" : "+v"(tid));   // opaque: what derives from the thread index is recomputed per tile, not carried across the loop
            if (ti >= n_full) {             // the remainder round: first come, first served
                if (R == 0) break;
                if (threadIdx.x == 0) s_next = __hip_atomic_fetch_add(&sync->grab, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - grab0;
                __syncthreads();            // (also: every wave has finished the reduction of the previous tile)
                extra = (unsigned)to_sgpr((int32_t)s_next);
                if (extra >= R) break;
                tile = n_full * G + extra;
                MARL_LOAD_TILE(tile, tid);  // (no prefetch here: whoever works on the remainder has time)
            } else {
                // every wave has finished the reduction of the previous tile (whose columns this tile's park / cache / edge writes
                // reuse) - or, first tile, has read the decision out of s_pub; the loads issued before it are in flight meanwhile
                __syncthreads();
            }
            sb.rebind(tid);
            double* pk = lds + SB::LDS_DOUBLES + tid;
            double yn[CPT][NF], k7[CPT][NF], esum[CPT][NF];
            PointAux aux[CPT];
            {
                const int64_t l = S.out_lo + (int64_t)tile * V - H + tid;
                sb.set_window(l + S.goff);
            }
            sb.reuse_live[0] = false;
#pragma unroll
            for (int f = 0; f < PARK; f++) pk[f * BLK] = y[0][f];
            MARL_STAMP45T(1);
            dp45_attempt<BLK, CPT, false, SB, PARK>(sb, h, y, k1, yn, k7, esum, aux, DenseWeights{}, pk);
            MARL_STAMP45T(2);
            // everything below is recomputed from the thread index again (nothing but the results is carried across the stages)
            int tid2 = threadIdx.x;
            asm volatile("" : "+v"(tid2));
            const int64_t l2 = S.out_lo + (int64_t)tile * V - H + tid2;
            const bool owned = tid2 >= H && tid2 < BLK - H && l2 >= S.out_lo && l2 < S.out_hi;
            double q[NQ];
            monitors_init(q);
            if (owned) {
                const double* pk2 = lds + SB::LDS_DOUBLES + tid2;
#pragma unroll
                for (int f = 0; f < NF; f++) q[0] += dp45_err2(esum[0][f], h, f < PARK ? pk2[f * BLK] : y[0][f], yn[0][f], rtol, atol);
                monitors_accumulate<true>(q, yn[0], aux[0].U, aux[0].W);
            }
            // the next tile's loads first, the stores behind them: vector memory operations return in order, so waiting for
            // those loads does not wait for the stores' acknowledgements (write-through to memory: ~2 us)
            // (unconditional - after the last tile it re-reads that tile: a conditional prefetch makes y, k1 of THIS tile the other
            // operand of the loop's phi, i.e. twenty more registers live across the six evaluations)
            MARL_LOAD_TILE(ti + 1 < n_full ? tile + G : tile, tid2);
            if (owned) {
#pragma unroll
                for (int f = 0; f < NF; f++) {
                    MARL_STREAM_STORE(yout + at<LAYOUT>(f, l2, S.ld), yn[0][f]);
                    MARL_STREAM_STORE(fout + at<LAYOUT>(f, l2, S.ld), k7[0][f]);
                }
            }
            MARL_STAMP45T(3);
            const double r = tile_reduce_halfwaves<BLK, SB>(q, lds, tid2);   // (one barrier inside)
            if ((tid2 & 31) == 0) {
                const int j = tid2 >> 5;
                if (j > 0) s_acc[j] = __builtin_fmin(s_acc[j], r);
                else if (ti < n_full) s_acc[0] += r;
                else __hip_atomic_store(gsum + G + extra, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            MARL_STAMP45T(4);
        }
#undef MARL_LOAD_TILE
        // ---- the barrier of the attempt ----
        MARL_STAMP45(1);
        if ((threadIdx.x & 31) == 0) {   // the sum into this workgroup's slot; the extrema by fp64 atomic minimum (exact whatever the order)
            const int j = threadIdx.x >> 5;
            if (j == 0) __hip_atomic_store(gsum + g, s_acc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_min(gmon + (((epoch_base + a) & 1u) * (NQ - 1) + (j - 1)) * MON_STRIDE, s_acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_s_waitcnt(0);   // this thread's stores (state, record) have been acknowledged
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __syncthreads();                 // ... and everybody else's
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(&sync->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old + 1u - arrive_base) == G * (a + 1u);
            s_abort = 0;
        }
        MARL_STAMP45(2);
        __syncthreads();
        const unsigned epoch = epoch_base + a + 1u;
        if (s_last) {   // (the whole workgroup) every workgroup's record of this attempt is in memory
            MARL_STAMP45L(0);
            // one round trip: the controller (one word per thread), the grid's extrema (threads 64..70), the workgroups' sums
            // (thread t: groups t, t + BLK, ... - four per pass, issued together)
            double* const mon = gmon + ((epoch_base + a) & 1u) * (NQ - 1) * MON_STRIDE;
            if ((int)threadIdx.x < CTRL_WORDS)
                reinterpret_cast<unsigned long long*>(&sc)[threadIdx.x] =
                    __hip_atomic_load(reinterpret_cast<unsigned long long*>(ctrl) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x >= 64 && threadIdx.x < 64 + NQ - 1) {
                const int j = threadIdx.x - 64 + 1;
                const double v = __hip_atomic_load(mon + (j - 1) * MON_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_rec[j] = (j > NQMIN) ? -v : v;
            }
            double e2 = 0.0;
            const unsigned GS = G + R;
            for (unsigned base = 0; base < GS; base += 4 * BLK) {
                double v[4];
                const unsigned r0 = base + threadIdx.x, last = GS - 1;
                const double* p0 = gsum + (r0 < GS ? r0 : last);
                const double* p1 = gsum + (r0 + BLK < GS ? r0 + BLK : last);
                const double* p2 = gsum + (r0 + 2 * BLK < GS ? r0 + 2 * BLK : last);
                const double* p3 = gsum + (r0 + 3 * BLK < GS ? r0 + 3 * BLK : last);
                asm volatile("global_load_dwordx2 %0, %4, off sc1\n\tglobal_load_dwordx2 %1, %5, off sc1\n\tglobal_load_dwordx2 %2, %6, off sc1\n\t"
                             "global_load_dwordx2 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
                e2 += (r0 < GS) ? v[0] : 0.0;
                e2 += (r0 + BLK < GS) ? v[1] : 0.0;
                e2 += (r0 + 2 * BLK < GS) ? v[2] : 0.0;
                e2 += (r0 + 3 * BLK < GS) ? v[3] : 0.0;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);
            if ((threadIdx.x & 63) == 0) s_wsum[threadIdx.x >> 6] = e2;
            MARL_STAMP45L(1);
            __syncthreads();
            if constexpr (DD) {   // domain decomposition: the slab's record and strips into the rank's message; nothing is decided here
                if (threadIdx.x == 0) {
                    double sumsq = 0.0;
#pragma unroll
                    for (int w = 0; w < NW; w++) sumsq += s_wsum[w];
                    send[0] = sumsq;
                    __hip_atomic_store(&sync->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
                    __hip_atomic_store(&sync->grab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (threadIdx.x >= 64 && threadIdx.x < 64 + NQ - 1) {
                    send[threadIdx.x - 64 + 1] = s_rec[threadIdx.x - 64 + 1];
                    __hip_atomic_store(mon + (threadIdx.x - 64) * MON_STRIDE, __builtin_inf(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                const int n = 2 * NF * halo;
                for (int i = threadIdx.x; i < n; i += BLK) {
                    const int aa = i / (NF * halo), f = (i / halo) % NF, j = i % halo;
                    const double* src = aa ? fout : yout;
                    send[NQ + i] = __hip_atomic_load(src + at<LAYOUT>(f, S.out_lo + j, S.ld), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    send[NQ + n + i] = __hip_atomic_load(src + at<LAYOUT>(f, S.out_hi - halo + j, S.ld), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                break;
            }
            if (threadIdx.x == 0) {
                double sumsq = 0.0;
#pragma unroll
                for (int w = 0; w < NW; w++) sumsq += s_wsum[w];
                const bool accepted = rk45_decide(sc, sumsq, &sb.T);
                if (!pause_on_event) {   // the status does not depend on this step's events: publish now, bookkeeping afterwards
                    if (accepted) rk45_advance_status_no_events(sc); else if (sc.status == ST_RUNNING) rk45_prepare_attempt(sc);
                } else {
                    s_rec[0] = 0.0;
                    const int fired = accepted ? rk45_events(sc, s_rec) : 0;
                    rk45_advance_status(sc, accepted, fired);
                }
                const unsigned long long hb = (unsigned long long)__double_as_longlong(sc.h_try);
                s_pub[0] = hb;
                s_pub[1] = pub_word(hb, epoch, sc.status, sc.cur);
                s_last = accepted ? 2 : 1;
            }
            MARL_STAMP45L(2);
            __syncthreads();
            if ((int)threadIdx.x < PUB_COPIES) {   // one 16-byte store per copy
                unsigned long long* p = sync->pub + threadIdx.x * PUB_STRIDE;
                const unsigned long long w0 = s_pub[0], w1 = s_pub[1];
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(__uint128_t(w0) | (__uint128_t(w1) << 64)) : "memory");
            }
            MARL_STAMP45L(3);
            // off the critical path: the driver's event bookkeeping of an accepted step (lanes 0..6: one monitor each), the extrema
            // re-armed for the attempt after next, the controller back to memory.  The next reader of either is the last workgroup
            // of a LATER barrier, which this workgroup joins only after these stores have been acknowledged; the host reads the
            // controller after the kernel.
            if (!pause_on_event && s_last == 2 && threadIdx.x < 7) rk45_event_one(sc, threadIdx.x, rk45_monitor_of_record(s_rec, threadIdx.x));
            if (threadIdx.x >= 64 && threadIdx.x < 64 + NQ - 1)
                __hip_atomic_store(mon + (threadIdx.x - 64) * MON_STRIDE, __builtin_inf(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if ((int)threadIdx.x < CTRL_WORDS)
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(ctrl) + threadIdx.x, reinterpret_cast<unsigned long long*>(&sc)[threadIdx.x],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            MARL_STAMP45L(4);
        }
        if constexpr (DD) break;   // (one attempt per launch: nothing to wait for)
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (true) {
                __uint128_t v;
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(my_pub) : "memory");
                const unsigned long long w0 = (unsigned long long)v, w1 = (unsigned long long)(v >> 64);
                if ((unsigned)(w1 >> 8) == epoch && w1 == pub_word(w0, epoch, (int)((w1 >> 1) & 7) - 1, (int)(w1 & 1))) {
                    s_pub[0] = w0;
                    s_pub[1] = w1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if (++spins > STREAM_SPIN_LIMIT || ((spins & 63u) == 0 && __hip_atomic_load(&sync->sticky, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(&sync->sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                    break;
                }
            }
        }
        __syncthreads();
        MARL_STAMP45(3);
        if (s_abort) break;
        // (LDS reads are not known to be uniform: without readfirstlane h, cur and the four buffer pointers live in vector registers)
        h = to_sgpr(__longlong_as_double((long long)s_pub[0]));
        const int32_t flags = to_sgpr((int32_t)(s_pub[1] & 0xff));
        status = ((flags >> 1) & 7) - 1;
        cur = flags & 1;
        // (the barrier in front of the next attempt's first tile - or the end of the kernel - separates these reads from thread 0's next writes)
    }
}

// U and W of a state (monitors zeros_U / zeros_W, LHeureux_model.py:567-593)
__device__ __forceinline__ void uw_point(double Phi, const DevConsts& C, const Tables& T, double& U, double& W)
{
    const double F = 1.0 - fast_exp(10.0 - 10.0 * rcp_nr(Phi), T);
    const double rF = C.hot.rhorat * F;
    U = C.hot.presum + rF * (Phi * Phi * Phi) * rcp_nr(1.0 - Phi);
    W = C.hot.presum - rF * Phi * Phi;
}

// Dense output of the step that starts at (yold, fold) with size h: replays the step's stages and
// writes  y(t_old + x h) = y_old + h sum_j w_j(x) K_j  (t_eval samples, ivp.py:706-723) and/or the
// monitors record of that state (event root finding, ivp.py:51-76).  yout may be NULL.
template <int BLK, int CPT, int LAYOUT, bool VD = false>
__global__ void __launch_bounds__(BLK) rk45_dense_kernel(const double* __restrict__ yold, const double* __restrict__ fold,
                                                         const DevConsts* __restrict__ consts, Slab S, double h,
                                                         DenseWeights dw, double* __restrict__ yout, double* __restrict__ part)
{
    constexpr int H = 6;
    constexpr int WIN = BLK * CPT;
    constexpr int V = WIN - 2 * H;
    using SB = StencilBlock<BLK, CPT, true, VD>;
    __shared__ double lds[SB::LDS_DOUBLES];
    const DevConsts& C = consts[0];
    const int64_t w0 = S.out_lo + (int64_t)blockIdx.x * V - H;
    const int64_t l0 = w0 + (int64_t)threadIdx.x * CPT;
    SB sb(lds, l0 + S.goff, consts);
    double y[CPT][NF], k1[CPT][NF], yn[CPT][NF], k7[CPT][NF], dsum[CPT][NF];
    PointAux aux[CPT];
    load_cells<CPT, LAYOUT>(yold, l0, S, C, y);
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int64_t l = l0 + c;
        const bool in = l >= 0 && l < S.n_buf;
#pragma unroll
        for (int f = 0; f < NF; f++) k1[c][f] = in ? fold[at<LAYOUT>(f, l, S.ld)] : 0.0;
    }
    dp45_attempt<BLK, CPT, true, SB>(sb, h, y, k1, yn, k7, dsum, aux, dw);
    double q[NQ];
    monitors_init(q);
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int wi = threadIdx.x * CPT + c;
        const int64_t l = l0 + c;
        if (wi >= H && wi < WIN - H && l >= S.out_lo && l < S.out_hi) {
            double d[NF];
#pragma unroll
            for (int f = 0; f < NF; f++) {
                d[f] = h * dsum[c][f] + y[c][f];
                if (yout) yout[at<LAYOUT>(f, l, S.ld)] = d[f];
            }
            double U, W;
            uw_point(d[4], C, sb.T, U, W);
            monitors_accumulate(q, d, U, W);
        }
    }
    block_reduce<BLK, NQ, NQMIN>(q, lds);   // the edge-exchange buffers are free now
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NQ; j++) part[(int64_t)blockIdx.x * NQ + j] = q[j];
    }
}

// ---------------------------------------------------------------------------------------------
// Domain decomposition (BASELINE config 5): a rank's slab buffer carries `halo` exchanged cells on each
// interior side.  pack: the slab's first / last `halo` OWNED cells of (y, f) -> contiguous strips
// [array(2)][field(5)][halo] for the neighbours; unpack: received strips -> the halo cells.
// `which`: 0 / 1 = explicit buffer; -1 = the buffer the attempt in flight wrote (cur ^ 1); -2 = cur.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int slab_which(int which, const Rk45Ctrl* ctrl)
{
    return which >= 0 ? which : (which == -1 ? (ctrl->cur ^ 1) : ctrl->cur);
}

__global__ void __launch_bounds__(128) slab_pack_kernel(const double* __restrict__ Y0, const double* __restrict__ Y1,
                                                        const double* __restrict__ F0, const double* __restrict__ F1,
                                                        const Rk45Ctrl* __restrict__ ctrl, int which, Slab S, int halo,
                                                        double* __restrict__ send_lo, double* __restrict__ send_hi)
{
    const int w = slab_which(which, ctrl);
    const double* Y = w ? Y1 : Y0;
    const double* F = w ? F1 : F0;
    const int n = 2 * NF * halo;
    for (int i = threadIdx.x; i < n; i += 128) {
        const int a = i / (NF * halo), f = (i / halo) % NF, j = i % halo;
        const double* src = a ? F : Y;
        if (send_lo) send_lo[i] = src[at<LAYOUT_FIELD_MAJOR>(f, S.out_lo + j, S.ld)];
        if (send_hi) send_hi[i] = src[at<LAYOUT_FIELD_MAJOR>(f, S.out_hi - halo + j, S.ld)];
    }
}

__global__ void __launch_bounds__(128) slab_unpack_kernel(double* __restrict__ Y0, double* __restrict__ Y1,
                                                          double* __restrict__ F0, double* __restrict__ F1,
                                                          const Rk45Ctrl* __restrict__ ctrl, int which, Slab S, int halo,
                                                          const double* __restrict__ recv_lo, const double* __restrict__ recv_hi)
{
    const int w = slab_which(which, ctrl);
    double* Y = w ? Y1 : Y0;
    double* F = w ? F1 : F0;
    const int n = 2 * NF * halo;
    for (int i = threadIdx.x; i < n; i += 128) {
        const int a = i / (NF * halo), f = (i / halo) % NF, j = i % halo;
        double* dst = a ? F : Y;
        if (recv_lo) dst[at<LAYOUT_FIELD_MAJOR>(f, S.out_lo - halo + j, S.ld)] = recv_lo[i];
        if (recv_hi) dst[at<LAYOUT_FIELD_MAJOR>(f, S.out_hi + j, S.ld)] = recv_hi[i];
    }
}


// ---- the per-attempt pair of the domain-decomposed loop (marl_slab_run) --------------------------------------------------
// After the attempt kernel: combine the slab's per-block records into ONE record and pack the strips the neighbours need, into
// the rank's message  send = [record (8) | lower strip (2*5*halo) | upper strip].  nb = 0: keep send[0..8) (set by the caller).
__global__ void __launch_bounds__(256) slab_reduce_pack_kernel(const double* __restrict__ Y0, const double* __restrict__ Y1,
                                                               const double* __restrict__ F0, const double* __restrict__ F1,
                                                               const Rk45Ctrl* __restrict__ ctrl, int which, Slab S, int halo,
                                                               const double* __restrict__ part, int64_t nb, double* __restrict__ send)
{
    __shared__ double scratch[NQ * 256];
    const int w = slab_which(which, ctrl);
    const double* Y = w ? Y1 : Y0;
    const double* F = w ? F1 : F0;
    const int n = 2 * NF * halo;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int a = i / (NF * halo), f = (i / halo) % NF, j = i % halo;
        const double* src = a ? F : Y;
        send[NQ + i] = src[at<LAYOUT_FIELD_MAJOR>(f, S.out_lo + j, S.ld)];
        send[NQ + n + i] = src[at<LAYOUT_FIELD_MAJOR>(f, S.out_hi - halo + j, S.ld)];
    }
    if (nb <= 0) return;
    double q[NQ];
    monitors_init(q);
    for (int64_t b = threadIdx.x; b < nb; b += 256) {
#pragma unroll
        for (int j = 0; j < NQ; j++) {
            const double o = part[b * NQ + j];
            q[j] = (j == 0) ? q[j] + o : (j <= NQMIN ? nanmin(q[j], o) : nanmax(q[j], o));
        }
    }
    block_reduce<256, NQ, NQMIN>(q, scratch);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int j = 0; j < NQ; j++) send[j] = q[j];
    }
}

// After the all-gather: copy the neighbours' strips out of the gathered messages into this slab's halo cells, then (do_control)
// combine the ranks' records IN RANK ORDER and finish the attempt - every rank computes the same decision from the same numbers.
__global__ void __launch_bounds__(256) slab_unpack_control_kernel(double* __restrict__ Y0, double* __restrict__ Y1, double* __restrict__ F0,
                                                                  double* __restrict__ F1, Rk45Ctrl* __restrict__ ctrl, int which, Slab S, int halo,
                                                                  const double* __restrict__ gathered, int rank, int world, int msg, int do_control)
{
    if (do_control && ctrl->status != ST_RUNNING) return;
    const int w = slab_which(which, ctrl);      // read before thread 0 may flip ctrl->cur below
    double* Y = w ? Y1 : Y0;
    double* F = w ? F1 : F0;
    const int n = 2 * NF * halo;
    const double* lo = (rank > 0 && S.out_lo > 0) ? gathered + (int64_t)(rank - 1) * msg + NQ + n : nullptr;          // lower neighbour's UPPER strip
    const double* hi = (rank < world - 1 && S.out_hi < S.n_buf) ? gathered + (int64_t)(rank + 1) * msg + NQ : nullptr;  // upper neighbour's LOWER strip
    for (int i = threadIdx.x; i < n; i += 256) {
        const int a = i / (NF * halo), f = (i / halo) % NF, j = i % halo;
        double* dst = a ? F : Y;
        if (lo) dst[at<LAYOUT_FIELD_MAJOR>(f, S.out_lo - halo + j, S.ld)] = lo[i];
        if (hi) dst[at<LAYOUT_FIELD_MAJOR>(f, S.out_hi + j, S.ld)] = hi[i];
    }
    __syncthreads();
    if (do_control && threadIdx.x == 0) {
        double q[NQ];
        monitors_init(q);
        for (int r = 0; r < world; r++) {
            const double* o = gathered + (int64_t)r * msg;
#pragma unroll
            for (int j = 0; j < NQ; j++) q[j] = (j == 0) ? q[j] + o[j] : (j <= NQMIN ? nanmin(q[j], o[j]) : nanmax(q[j], o[j]));
        }
        Rk45Ctrl c = *ctrl;
        rk45_finish_attempt(c, q);
        *ctrl = c;
    }
}

// records of all ranks (stride msg) -> dense [world][8]
__global__ void slab_records_kernel(const double* __restrict__ gathered, int world, int msg, double* __restrict__ recs)
{
    for (int i = threadIdx.x; i < world * NQ; i += blockDim.x) recs[i] = gathered[(int64_t)(i / NQ) * msg + (i % NQ)];
}

// owned cells of a slab <-> a dense [5][n_own] array
__global__ void __launch_bounds__(256) slab_copy_kernel(double* __restrict__ Y0, double* __restrict__ Y1,
                                                        const Rk45Ctrl* __restrict__ ctrl, int which, Slab S,
                                                        double* __restrict__ dense, int to_dense)
{
    double* Y = slab_which(which, ctrl) ? Y1 : Y0;
    const int64_t n_own = S.out_hi - S.out_lo;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_own) return;
#pragma unroll
    for (int f = 0; f < NF; f++) {
        if (to_dense) dense[f * n_own + i] = Y[at<LAYOUT_FIELD_MAJOR>(f, S.out_lo + i, S.ld)];
        else Y[at<LAYOUT_FIELD_MAJOR>(f, S.out_lo + i, S.ld)] = dense[f * n_own + i];
    }
}

// ---------------------------------------------------------------------------------------------
// Batched parameter sweep (BASELINE configs 3-4): ONE workgroup integrates ONE instance
// (N <= BLK*CPT cells) for all its steps.  State, stage vectors and the step controller stay
// on-chip for the whole integration; global memory is touched at entry and exit only.
//   Y: [batch][5][N] field-major per instance (the reference's layout, one instance after another).
// ---------------------------------------------------------------------------------------------
// (1024-thread workgroups cap a thread at 128 VGPRs, which this kernel overruns: 38 spilled VGPRs.  Parking the step's first state in
// LDS, and a 6th-order reuse tier for coarse grids, were measured and dropped - profiles/r02_lab_sweep_experiments.log)
//
// Two loop shapes, one arithmetic (dp45_attempt, the error norm, the monitors, rk45_decide / rk45_events / rk45_advance_status):
//   FAST = false  the round-1/2 loop: 8-quantity block reduction (4 barriers), scipy's controller on ONE lane while 1023 wait,
//                 2 more barriers to publish it.  Kept for single small runs that pause on a monitor sign change (root finding:
//                 the decision to pause needs the events of THIS step).
//   FAST = true   (round 3; sweeps, and single runs without event pauses) the step controller off the critical path: each wave
//                 butterflies its own sum of squares, ONE barrier makes the 16 partial sums visible, and EVERY wave adds them in
//                 wave order and runs decide + advance_status itself - replicated scalar work, but no single-lane section and no
//                 publish barriers; the controller's scalars are pushed into SGPRs (readfirstlane) so the replication costs no
//                 vector registers.  The seven monitors go through LDS columns written before that one barrier; waves 0..6 reduce
//                 one quantity each after it, and lanes 0..6 of wave 0 do the event bookkeeping (rk45_event_one) one barrier
//                 later - after the first evaluation of the NEXT attempt, whose exchange barrier orders it - or after the loop.
//                 Per attempt: 7 barriers instead of 12 (profiles/r03_lab_sweep.log).
// the fields of the step controller that rk45_decide / rk45_advance_status / rk45_prepare_attempt read or write
__device__ __forceinline__ void ctrl_to_sgpr(Rk45Ctrl& c)
{
    c.t = to_sgpr(c.t); c.h_abs = to_sgpr(c.h_abs); c.t_bound = to_sgpr(c.t_bound); c.h_try = to_sgpr(c.h_try); c.t_new = to_sgpr(c.t_new);
    c.t_old = to_sgpr(c.t_old); c.h_prev = to_sgpr(c.h_prev); c.pause_t = to_sgpr(c.pause_t);
    c.nfev = to_sgpr(c.nfev); c.n_acc = to_sgpr(c.n_acc); c.n_rej = to_sgpr(c.n_rej); c.attempts = to_sgpr(c.attempts);
    c.max_attempts = to_sgpr(c.max_attempts); c.n_total = to_sgpr(c.n_total);
    c.status = to_sgpr(c.status); c.cur = to_sgpr(c.cur); c.rejected = to_sgpr(c.rejected); c.accepted_last = to_sgpr(c.accepted_last);
    c.pause_on_event = to_sgpr(c.pause_on_event);
}

#ifndef MARL_SWEEP_PARK
#define MARL_SWEEP_PARK 2
#endif
template <int BLK, int CPT, bool VD, bool FAST>
__device__ __forceinline__ void rk45_sweep_body(double* __restrict__ Y, const DevConsts* __restrict__ consts, Rk45Ctrl* __restrict__ ctrls, int64_t N,
                                                double* __restrict__ Yold, double* __restrict__ Fold)
{
    using SB = StencilBlock<BLK, CPT, false, VD>;
    constexpr int NW = BLK / 64;
    constexpr int NMON = NQ - 1;                                  // the seven monitor extrema (record slots 1..7)
    constexpr int FAST_DOUBLES = FAST ? NMON * BLK + NW + NQ : 0; // monitor columns [slot][thread], per-wave sums, the reduced record
    // PARK fields of the step's first state y live in the thread's LDS column instead of registers (dp45_attempt): the 1024-thread
    // shape is capped at 128 VGPRs and every scratch reload in the stage loop stalls all 16 waves in front of their barrier
    // (measured: 6.6 -> 21.7 reloads per attempt = -17 %); two columns are what the 160 KB of LDS still hold beside the monitor columns
    constexpr int PARK = (FAST && BLK == 1024 && CPT == 1) ? MARL_SWEEP_PARK : 0;
    __shared__ double lds[SB::LDS_DOUBLES + FAST_DOUBLES + PARK * CPT * BLK];
    __shared__ Rk45Ctrl sc;
    const DevConsts& C = consts[blockIdx.x];
    double* yg = Y + (int64_t)blockIdx.x * NF * N;
    Slab S = {N, 0, N, 0, N};
    const int64_t l0 = (int64_t)threadIdx.x * CPT;
    SB sb(lds, l0, consts + blockIdx.x);
    if (threadIdx.x == 0) sc = ctrls[blockIdx.x];
    __syncthreads();
    if (sc.status != ST_RUNNING) return;
    const double rtol = FAST ? to_sgpr(sc.rtol) : sc.rtol, atol = FAST ? to_sgpr(sc.atol) : sc.atol;

    double y[CPT][NF], k1[CPT][NF], yn[CPT][NF], k7[CPT][NF], esum[CPT][NF];
    PointAux aux[CPT];
    load_cells<CPT, LAYOUT_FIELD_MAJOR>(yg, l0, S, C, y);
    sb.eval(y, k1, aux);  // f(t, y): RungeKutta.__init__ (first launch) or re-derived on resume
    double* pk = lds + SB::LDS_DOUBLES + FAST_DOUBLES + threadIdx.x;
    if constexpr (PARK > 0) {
#pragma unroll
        for (int c = 0; c < CPT; c++)
#pragma unroll
            for (int f = 0; f < PARK; f++) pk[(c * NF + f) * BLK] = y[c][f];
    }
#define MARL_Y(c, f) ((f) < PARK ? pk[((c) * NF + (f)) * BLK] : y[c][f])
#define MARL_KEEP_OLD_STEP()                                                                                               \
    _Pragma("unroll") for (int c = 0; c < CPT; c++) {                                                                      \
        if (l0 + c < N) {                                                                                                  \
            _Pragma("unroll") for (int f = 0; f < NF; f++) {                                                               \
                Yold[(int64_t)blockIdx.x * NF * N + at<LAYOUT_FIELD_MAJOR>(f, l0 + c, N)] = MARL_Y(c, f);                  \
                Fold[(int64_t)blockIdx.x * NF * N + at<LAYOUT_FIELD_MAJOR>(f, l0 + c, N)] = k1[c][f];                      \
            }                                                                                                              \
        }                                                                                                                  \
    }

    if constexpr (!FAST) {
        while (true) {
            const double h = sc.h_try;
            dp45_attempt<BLK, CPT, false, SB>(sb, h, y, k1, yn, k7, esum, aux);
            double q[NQ];
            monitors_init(q);
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                if (l0 + c < N) {
#pragma unroll
                    for (int f = 0; f < NF; f++) q[0] += dp45_err2(esum[c][f], h, MARL_Y(c, f), yn[c][f], rtol, atol);
                    monitors_accumulate(q, yn[c], aux[c].U, aux[c].W);
                }
            }
            block_reduce<BLK, NQ, NQMIN>(q, lds);   // the edge-exchange buffers are free now
            if (threadIdx.x == 0) rk45_finish_attempt(sc, q, &sb.T);
            __syncthreads();
            const int status = sc.status;
            if (sc.accepted_last) {
                if (status != ST_RUNNING && Yold) {
                    // keep (y_old, f_old) of the step just accepted: the host replays it for dense output
                    MARL_KEEP_OLD_STEP()
                }
#pragma unroll
                for (int c = 0; c < CPT; c++)
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        y[c][f] = yn[c][f];
                        k1[c][f] = k7[c][f];
                    }
            }
            __syncthreads();  // everyone has read sc before thread 0 may touch it again
            if (status != ST_RUNNING) break;
        }
    } else {
        double* mon = lds + SB::LDS_DOUBLES;         // [slot 1..7][thread]: never touched by the evaluations' edge exchange
        double* wsum = mon + NMON * BLK;             // [wave]
        double* rec = wsum + NW;                     // [NQ]: the reduced monitors of the last accepted step (slots 1..7)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        // the controller, replicated: every lane holds the same values; what lives across an attempt sits in SGPRs
        // (tried: the controller in a double-buffered LDS block re-read after the barrier - fewer SGPRs, but the register
        // allocator then spilled more vector registers: 21.7 instead of 6.6 - 15.7 scratch reloads per attempt, -17 %)
        Rk45Ctrl c = sc;
        ctrl_to_sgpr(c);
        const double inv_n = to_sgpr(1.0 / (double)c.n_total);
        bool events_pending = false;                 // an accepted step whose monitors (rec) have not been compared with sc.g yet
        auto event_bookkeeping = [&]() {             // lanes 0..6 of wave 0: one monitor each; sc.t_old / sc.h_prev / sc.t describe that step
            if (threadIdx.x < 7) rk45_event_one(sc, threadIdx.x, rk45_monitor_of_record(rec, threadIdx.x));
        };
        while (true) {
            const double h = c.h_try;
            // (the first evaluation's exchange barrier also orders the monitor reduction of the previous attempt before the
            // event bookkeeping below; dp45_attempt is opaque, so the bookkeeping follows the whole attempt's evaluations -
            // still before this attempt's own decision changes c.t_old / c.h_prev / c.t)
            dp45_attempt<BLK, CPT, false, SB, PARK>(sb, h, y, k1, yn, k7, esum, aux, DenseWeights{}, pk);
            if (events_pending) {                    // wave-uniform
                event_bookkeeping();
                events_pending = false;
            }
            double q[NQ];
            monitors_init(q);
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) {
                if (l0 + cc < N) {
#pragma unroll
                    for (int f = 0; f < NF; f++) q[0] += dp45_err2(esum[cc][f], h, MARL_Y(cc, f), yn[cc][f], rtol, atol);
                    monitors_accumulate<true>(q, yn[cc], aux[cc].U, aux[cc].W);   // (only read for accepted attempts: all finite)
                }
            }
            double e2 = q[0];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) e2 += __shfl_xor(e2, off, 64);   // every lane: the wave's sum (same order in every lane pair)
            if (lane == 0) wsum[wave] = e2;
#pragma unroll
            for (int j = 1; j < NQ; j++) mon[(j - 1) * BLK + threadIdx.x] = q[j];
            __syncthreads();                         // the ONE barrier between the attempt and the decision
            double sumsq = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w++) sumsq += wsum[w];                          // wave order: the same sum in every wave
            sumsq = to_sgpr(sumsq);
            const bool accepted = rk45_decide_lean(c, sumsq, inv_n, sb.T);
            if (accepted) {
                if (threadIdx.x == 0) { sc.t_old = c.t_old; sc.h_prev = c.h_prev; sc.t = c.t; }   // the step the pending events belong to
                if (c.t - c.t_bound >= 0.0) c.status = ST_DONE;   // (rk45_advance_status without the event pause, which this loop shape does not have)
                else if (c.t >= c.pause_t) c.status = ST_PAUSED;
            }
            if (c.status == ST_RUNNING) rk45_prepare_attempt_lean(c);
            c.t = to_sgpr(c.t); c.h_abs = to_sgpr(c.h_abs); c.h_try = to_sgpr(c.h_try); c.t_new = to_sgpr(c.t_new);
            const int status = c.status;
            if (accepted) {
                // monitors of y_new: wave w reduces record slot w + 1 (+ NW, ...) over the workgroup's columns
#pragma unroll
                for (int jj = 0; jj < (NMON + NW - 1) / NW; jj++) {
                    const int j = wave + jj * NW + 1;   // wave-uniform
                    if (j < NQ) {
                        // (finite values: see monitors_accumulate<true>; a maximum is the negated minimum of the negated values)
                        const double sgn = (j <= NQMIN) ? 1.0 : -1.0;
                        double a = sgn * mon[(j - 1) * BLK + lane];
#pragma unroll
                        for (int i = 1; i < NW; i++) a = __builtin_fmin(a, sgn * mon[(j - 1) * BLK + lane + 64 * i]);
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) a = __builtin_fmin(a, __shfl_xor(a, off, 64));
                        if (lane == 0) rec[j] = sgn * a;
                    }
                }
                events_pending = true;
                if (status != ST_RUNNING && Yold) {
                    MARL_KEEP_OLD_STEP()
                }
#pragma unroll
                for (int cc = 0; cc < CPT; cc++)
#pragma unroll
                    for (int f = 0; f < NF; f++) {
                        if (f < PARK) pk[(cc * NF + f) * BLK] = yn[cc][f]; else y[cc][f] = yn[cc][f];
                        k1[cc][f] = k7[cc][f];
                    }
            }
            if (status != ST_RUNNING) break;
        }
        __syncthreads();                             // the last accepted step's monitors are reduced
        if (events_pending) event_bookkeeping();
        __syncthreads();
        if (threadIdx.x == 0) {                      // the replicated controller back into the record that leaves the kernel
            sc.t = c.t; sc.h_abs = c.h_abs; sc.h_try = c.h_try; sc.t_new = c.t_new;
            sc.err_norm = sqrt(c.err_norm);          // (rk45_decide_lean keeps the mean square)
            sc.nfev = c.nfev; sc.n_acc = c.n_acc; sc.n_rej = c.n_rej; sc.attempts = c.attempts;
            sc.status = c.status; sc.rejected = c.rejected; sc.accepted_last = c.accepted_last; sc.cur = c.cur;
        }
    }
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        if (l0 + c < N) {
#pragma unroll
            for (int f = 0; f < NF; f++) yg[at<LAYOUT_FIELD_MAJOR>(f, l0 + c, N)] = MARL_Y(c, f);
        }
    }
#undef MARL_Y
#undef MARL_KEEP_OLD_STEP
    if (threadIdx.x == 0) {
        sc.cur = 0;
        ctrls[blockIdx.x] = sc;
    }
}

// sweeps and single small runs that never pause on a monitor sign change
template <int BLK, int CPT, bool VD = false>
__global__ void __launch_bounds__(BLK) rk45_sweep_kernel(double* __restrict__ Y, const DevConsts* __restrict__ consts,
                                                         Rk45Ctrl* __restrict__ ctrls, int64_t N,
                                                         double* __restrict__ Yold, double* __restrict__ Fold)
{
    rk45_sweep_body<BLK, CPT, VD, true>(Y, consts, ctrls, N, Yold, Fold);
}

// single small runs with event root finding (Rk45Ctrl.pause_on_event): the decision to pause needs this step's events
template <int BLK, int CPT, bool VD = false>
__global__ void __launch_bounds__(BLK) rk45_sweep_events_kernel(double* __restrict__ Y, const DevConsts* __restrict__ consts,
                                                                Rk45Ctrl* __restrict__ ctrls, int64_t N,
                                                                double* __restrict__ Yold, double* __restrict__ Fold)
{
    rk45_sweep_body<BLK, CPT, VD, false>(Y, consts, ctrls, N, Yold, Fold);
}

template <int BLK, int CPT, bool VD = false>
__global__ void __launch_bounds__(BLK) rk4_sweep_kernel(double* __restrict__ Y, const DevConsts* __restrict__ consts,
                                                        const double* __restrict__ dts, int64_t N, int64_t nsteps)
{
    using SB = StencilBlock<BLK, CPT, false, VD>;
    __shared__ double lds[SB::LDS_DOUBLES];
    const DevConsts& C = consts[blockIdx.x];
    double* yg = Y + (int64_t)blockIdx.x * NF * N;
    Slab S = {N, 0, N, 0, N};
    const int64_t l0 = (int64_t)threadIdx.x * CPT;
    SB sb(lds, l0, consts + blockIdx.x);
    const double dt = dts[blockIdx.x];
    const double h2 = 0.5 * dt, h6 = dt / 6.0;
    double y[CPT][NF], ys[CPT][NF], k[CPT][NF], acc[CPT][NF];
    PointAux aux[CPT];
    load_cells<CPT, LAYOUT_FIELD_MAJOR>(yg, l0, S, C, y);
#define MARL_CELLS _Pragma("unroll") for (int c = 0; c < CPT; c++) _Pragma("unroll") for (int f = 0; f < NF; f++)
#pragma unroll 1
    for (int64_t s = 0; s < nsteps; s++) {
        sb.template eval<TR_FILL>(y, k, aux);
        MARL_CELLS { acc[c][f] = k[c][f]; ys[c][f] = y[c][f] + h2 * k[c][f]; }
        sb.template eval<TR_REUSE>(ys, k, aux);
        MARL_CELLS { acc[c][f] = acc[c][f] + 2.0 * k[c][f]; ys[c][f] = y[c][f] + h2 * k[c][f]; }
        sb.template eval<TR_REUSE>(ys, k, aux);
        MARL_CELLS { acc[c][f] = acc[c][f] + 2.0 * k[c][f]; ys[c][f] = y[c][f] + dt * k[c][f]; }
        sb.template eval<TR_REUSE>(ys, k, aux);
        MARL_CELLS y[c][f] = y[c][f] + h6 * (acc[c][f] + k[c][f]);
    }
#undef MARL_CELLS
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        if (l0 + c < N) {
#pragma unroll
            for (int f = 0; f < NF; f++) yg[at<LAYOUT_FIELD_MAJOR>(f, l0 + c, N)] = y[c][f];
        }
    }
}

}  // namespace marl
