// Compile-only probe: the one-workgroup sweep kernels alone (register / scratch / LDS checks in seconds):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -disable-machine-licm --cuda-device-only -c -Rpass-analysis=kernel-resource-usage -o /tmp/sweep_only.o tools/lab_src/sweep_only.hip
#include "../../integrating-diagenetic-equations-using-python_amd/csrc/marl_kernels.h"
template __global__ void marl::rk45_sweep_kernel<1024, 1, false>(double*, const marl::DevConsts*, marl::Rk45Ctrl*, int64_t, double*, double*);
