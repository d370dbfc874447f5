// marl_rk4_small.hip - the fused fixed-step kernel of SMALL grids (BASELINE configs[1]: N = 65 536; every grid of up to 98 304 cells takes
// the 16-steps-per-launch kernel), compiled as its own translation unit because it wants other choices than the rest of the library:
//
//   * at this size the grid is ~1 000 useful waves for 1 024 SIMDs - one or two resident waves per SIMD instead of four.  A register
//     cap of 128 (four waves per SIMD) buys nothing, and nothing hides an LDS round trip: all six values of the transcendental
//     expansions' centre live in VGPRs (MARL_CACHE_LDS_SLOTS = 0), the allocator gets the 256-VGPR budget (amdgpu_waves_per_eu(1, 2));
//   * the machine scheduler orders for instruction-level parallelism (-mllvm -amdgpu-sched-strategy=max-ilp, a whole-unit switch:
//     hence the unit) instead of for occupancy.
//   Measured (tools/rk4_lab.hip, N = 65 536, us per step; profiles/r04_lab_n65536.log): shipped shape 3.648 | 256-VGPR budget alone 3.616 |
//   all slots in VGPRs + budget 3.553 | + max-ilp 3.427 (+6.5 %).
//
// Every symbol of the shared headers lands in its own namespace (marl_small): device code of the two units is linked separately
// (no relocatable device code), and the host-side kernel stubs must not collide.
#define marl marl_small
#define MARL_CACHE_LDS_SLOTS 0
#define MARL_LAB_RK4_WAVES_MIN 1
#define MARL_LAB_RK4_WAVES_MAX 2
#include "marl_kernels.h"
#undef marl

// slab5: n_buf, goff, ld, out_lo, out_hi (marl::Slab, field for field); consts: the context's device constant block.
// Returns the hipGetLastError() of the launch.
extern "C" __attribute__((visibility("hidden"))) int marl_small_rk4_16(int tiled, unsigned grid, hipStream_t stream, const double* yin, double* yout,
                                                                       const void* consts, const int64_t slab5[5], double dt)
{
    using namespace marl_small;
    const Slab S{slab5[0], slab5[1], slab5[2], slab5[3], slab5[4]};
    const DevConsts* c = static_cast<const DevConsts*>(consts);
    if (tiled)
        hipLaunchKernelGGL((rk4_fused_kernel<256, 1, LAYOUT_TILED, 16>), dim3(grid), dim3(256), 0, stream, yin, yout, c, S, dt);
    else
        hipLaunchKernelGGL((rk4_fused_kernel<256, 1, LAYOUT_FIELD_MAJOR, 16>), dim3(grid), dim3(256), 0, stream, yin, yout, c, S, dt);
    return (int)hipGetLastError();
}
