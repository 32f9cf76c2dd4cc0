// Compiles ONE kernel of the library in isolation, to read its ISA without the 2.5-minute build of every
// instantiation:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I fbus-ekf_amd/csrc -I include \
//                       -S --cuda-device-only -DKERNEL=1 tools/isa_one_kernel.hip -o /tmp/k.s
#include "ekf_kernels.hpp"
#ifndef KERNEL
#define KERNEL 1
#endif
#ifndef KJOINT
#define KJOINT true
#endif
const void* fbus_isa_keep()
{
#if KERNEL == 1
    return (const void*)predict_kernel<float, 18, 0, false>;
#elif KERNEL == 2
    return (const void*)correct_kernel<float, 18, 0, 0, KJOINT>;
#elif KERNEL == 3
    return (const void*)frame_kernel<float, 18, 0, 0, KJOINT>;
#endif
}
