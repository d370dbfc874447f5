// Compile-only probe: the persistent RK45 kernel (and the per-attempt kernel beside it) alone, for quick register / scratch / ISA checks:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -disable-machine-licm --cuda-device-only -c -Rpass-analysis=kernel-resource-usage \
//         -o /tmp/rk45_stream_only.o tools/lab_src/rk45_stream_only.hip
#include "../../integrating-diagenetic-equations-using-python_amd/csrc/marl_kernels.h"
template __global__ void marl::rk45_stream_kernel<256, marl::LAYOUT_TILED, false, false>(double*, double*, double*, double*, const marl::DevConsts*, marl::Slab,
                                                                                   marl::Rk45Ctrl*, double*, marl::Rk45Stream*, unsigned, unsigned, unsigned, unsigned, unsigned, double*, int);
template __global__ void marl::rk45_attempt_kernel<256, 1, marl::LAYOUT_TILED, false>(double*, double*, double*, double*, const marl::DevConsts*, marl::Slab,
                                                                                       const marl::Rk45Ctrl*, double*);
