// The four stages of one ImuUpdate compiled one by one, to count what each costs in VALU issue slots (scalar / packed):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I fbus-ekf_amd/csrc -I include -c tools/stage_isa.hip -o /tmp/stage.o
//   python tools/isa_stats.py /tmp/stage.o stage_
// Every kernel loads the nominal state and the packed covariance with plain loads, runs one stage and stores everything it may
// have changed; the ld/st kernel is the overhead to subtract.
#include "ekf_kernels.hpp"
#ifndef SN
#define SN 18
#endif
#ifndef SDIALECT
#define SDIALECT 0
#endif
namespace {
template <int STAGE>
__global__ void __launch_bounds__(64) stage_kernel(float* __restrict__ recs, const float* __restrict__ imu, int B)
{
    using RC = Rec<float, SN>;
    const int b = blockIdx.x * 64 + threadIdx.x;
    float* r = recs + (size_t)b * RC::NRECP;
    float nom[Lay<SN>::NNOM], P[RC::NCOVP];
#pragma unroll
    for (int i = 0; i < Lay<SN>::NNOM; ++i) nom[i] = r[i];
#pragma unroll
    for (int i = 0; i < RC::NCOVP; ++i) P[i] = r[Lay<SN>::NNOM + i];
    const float a[3] = { imu[b * 7], imu[b * 7 + 1], imu[b * 7 + 2] }, w[3] = { imu[b * 7 + 3], imu[b * 7 + 4], imu[b * 7 + 5] };
    const float dt = imu[b * 7 + 6];
    const float qd[4] = { imu[0], imu[1], imu[2], imu[3] };
    PredictCoef<float> k;
    if constexpr (STAGE == 0) {
        predict_nominal<float, SN, SDIALECT>(nom, a, w, dt, k);
        // the coefficient blocks leave through the covariance slots so that they are not dead
#pragma unroll
        for (int i = 0; i < 9; ++i) { P[i] = k.A[i]; P[9 + i] = k.Bm[i]; P[18 + i] = k.Th[i]; }
    } else if constexpr (STAGE >= 1 && STAGE <= 3) {
        // coefficients straight from memory: the stage alone
#pragma unroll
        for (int i = 0; i < 9; ++i) { k.A[i] = imu[B + b * 28 + i]; k.Bm[i] = imu[B + b * 28 + 9 + i]; k.Th[i] = imu[B + b * 28 + 18 + i]; }
        k.dt = dt;
        if constexpr (STAGE == 1) cov_stage_p<float, SN>(P, k);
        if constexpr (STAGE == 2) cov_stage_v<float, SN>(P, k, qd);
        if constexpr (STAGE == 3) cov_stage_th<float, SN>(P, k, qd);
    } else if constexpr (STAGE == 4) {
        predict_step<float, SN, SDIALECT>(nom, P, a, w, dt, qd);
    }
#pragma unroll
    for (int i = 0; i < Lay<SN>::NNOM; ++i) r[i] = nom[i];
#pragma unroll
    for (int i = 0; i < RC::NCOVP; ++i) r[Lay<SN>::NNOM + i] = P[i];
}
}
const void* fbus_stage_keep(int s)
{
    switch (s) {
        case 0: return (const void*)stage_kernel<0>;
        case 1: return (const void*)stage_kernel<1>;
        case 2: return (const void*)stage_kernel<2>;
        case 3: return (const void*)stage_kernel<3>;
        case 4: return (const void*)stage_kernel<4>;
        default: return (const void*)stage_kernel<5>;
    }
}
