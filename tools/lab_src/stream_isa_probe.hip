// tests/test_stream_isa.py compiles this translation unit (device code only) to check tools/check_stream_isa.py against a
// DELIBERATELY broken loop shape (-DMARL_LAB_BROKEN_STREAM_LATCH: the item grab in the loop latch, the shape that produced
// stale tiles in round 2) and against the shipped shape, for two depths.
#include "../../integrating-diagenetic-equations-using-python_amd/csrc/marl_kernels.h"

namespace marl {
template __global__ void rk4_stream_kernel<256, LAYOUT_TILED, 4, false>(double*, double*, const DevConsts*, Slab, double, unsigned, unsigned, unsigned*,
                                                                        unsigned*, unsigned*, unsigned, unsigned, double*);
template __global__ void rk4_stream_kernel<256, LAYOUT_TILED, 1, false>(double*, double*, const DevConsts*, Slab, double, unsigned, unsigned, unsigned*,
                                                                        unsigned*, unsigned*, unsigned, unsigned, double*);
}  // namespace marl
