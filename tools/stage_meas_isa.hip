// One marker of the reprojection fold compiled alone, to count what a corner costs in VALU issue slots (round 6, review item 4):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I fbus-ekf_amd/csrc -I include -c tools/stage_meas_isa.hip -o /tmp/stage_meas.o
//   python tools/isa_stats.py /tmp/stage_meas.o fold
// fold_kernel<NCAM, NMK>: NMK markers through pixel_fold_marker<NCAM, float, true> (square port, fp32 records) from plain loads into
// the 27 sums; count(NMK = 1) - count(NMK = 0) = instructions per marker.  corner_kernel<NMK>: the same for the corner-position fold
// (tri_corners_refractive + corner_fold_marker; python tools/isa_stats.py /tmp/stage_meas.o corner).
#include "ekf_kernels.hpp"
#include "ekf_meas.hpp"
namespace {
template <int NCAM, int NMK, bool CF = false>
__global__ void __launch_bounds__(64) fold_kernel(const double* __restrict__ in, const float* __restrict__ y, double* __restrict__ out, MeasConst mc)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    double p[3], R[9], pil[3];
    for (int i = 0; i < 3; ++i) p[i] = in[b * 32 + i];
    for (int i = 0; i < 9; ++i) R[i] = in[b * 32 + 3 + i];
    for (int i = 0; i < 3; ++i) pil[i] = in[b * 32 + 12 + i];
    PixAcc acc;
    acc.clear();
#pragma unroll 1
    for (int m = 0; m < NMK; ++m) {
        double mk[9];
        for (int i = 0; i < 9; ++i) mk[i] = mc.mkc[m * 9 + i];
        float yl[8], yr[8];
        for (int i = 0; i < 8; ++i) { yl[i] = y[(b * NMK + m) * 16 + i]; yr[i] = y[(b * NMK + m) * 16 + 8 + i]; }
        pixel_fold_marker<NCAM, float, true, 4, 0, CF>(acc, p, R, pil, mc, mk, yl, NCAM == 2 ? yr : yl, 0.15);
    }
    if constexpr (CF) acc.to_imu_frame(mc.adjL);
    for (int i = 0; i < PixAcc::NVAL; ++i) out[b * 32 + i] = acc.at(i);
}
// the corner-position fold (correct_corners): triangulation through the square port + 12 position rows per marker
template <int NMK>
__global__ void __launch_bounds__(64) corner_kernel(const double* __restrict__ in, const float* __restrict__ y, double* __restrict__ out, MeasConst mc,
                                                    VisConst<double> vc)
{
    __builtin_assume(vc.sqrt_minus0 && vc.sqrt_minus1);          // the reference's configuration: count the path that runs
    const int b = blockIdx.x * 64 + threadIdx.x;
    double p[3], R[9], pil[3];
    for (int i = 0; i < 3; ++i) p[i] = in[b * 32 + i];
    for (int i = 0; i < 9; ++i) R[i] = in[b * 32 + 3 + i];
    for (int i = 0; i < 3; ++i) pil[i] = in[b * 32 + 12 + i];
    PixAcc acc;
    acc.clear();
#pragma unroll 1
    for (int m = 0; m < NMK; ++m) {
        double mk[9];
        for (int i = 0; i < 9; ++i) mk[i] = mc.mkc[m * 9 + i];
        float yl[8], yr[8];
        for (int i = 0; i < 8; ++i) { yl[i] = y[(b * NMK + m) * 16 + i]; yr[i] = y[(b * NMK + m) * 16 + 8 + i]; }
        double C[4][3];
        tri_corners_refractive<float, true>(vc, yl, yr, C);
        corner_fold_marker(acc, p, R, pil, mc, mk, C, 0.15);
    }
    for (int i = 0; i < PixAcc::NVAL; ++i) out[b * 32 + i] = acc.at(i);
}
}
const void* fbus_stage_corner_keep(int s) { return s ? (const void*)corner_kernel<1> : (const void*)corner_kernel<0>; }
const void* fbus_stage_meas_keep(int s)
{
    switch (s) {
        case 0: return (const void*)fold_kernel<1, 0>;
        case 1: return (const void*)fold_kernel<1, 1>;
        case 2: return (const void*)fold_kernel<1, 2>;
        case 3: return (const void*)fold_kernel<2, 1>;
        case 5: return (const void*)fold_kernel<1, 0, true>;
        case 6: return (const void*)fold_kernel<1, 1, true>;
        default: return (const void*)fold_kernel<2, 2>;
    }
}
