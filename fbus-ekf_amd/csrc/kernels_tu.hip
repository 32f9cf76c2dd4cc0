// kernels_tu.hip -- one kernel family for one (scalar type, state size), both dialects.
// Compiled 16 times by build.py:  -DFBUS_TU_T=float|double  -DFBUS_TU_N=18|15  -DFBUS_TU_FAMILY=1..6
//   1 predict (per-call streamed kernel, both record-load policies, and predict_n)
//   2 correct (nearest / stacked x simple / Joseph)
//   3 fused frame (K predicts + correct in one launch)
//   4 (removed in round 4: the fp32-fold corner / pixel kernels; see 7)
//   5 frame window (F frames per launch)
//   6 team kernels (several waves per tile: predict, predict_n, frame window; fp32 only)
//   7 correct from corner pixels / from stereo corners (ekf_meas.hpp: double-precision fold, non-cancelling update)
//   9 correct from corner pixels with the update divided between the waves of a tile (ekf_meas_split.hpp; fp32 only)
//   8 fused frame with that update (K predicts + correct_pixels / correct_corners in one launch; fp32 only)
// gfx950 only.
#include <cstdlib>
#include "ekf_kernels.hpp"
#include "ekf_team.hpp"
#include "ekf_meas.hpp"
#if FBUS_TU_FAMILY == 9
#include "ekf_meas_split.hpp"
#endif
#include <atomic>
#include "ekf_launch.hpp"

#ifndef FBUS_TU_T
#error "kernels_tu.hip: define FBUS_TU_T, FBUS_TU_N and FBUS_TU_FAMILY (see build.py)"
#endif

namespace fbus {

// LaunchPolicy::two_wave_min_b (ekf_launch.hpp; default SIMDs x 64 + 1): from this many filters on a launch has more waves than
// the chip has SIMDs: some SIMDs hold two, and the instantiations written for at most 256 registers (row-split correct, parked
// predict_n / frame) let them run side by side instead of one after the other; up to one wave per SIMD every SIMD holds one wave
// whatever the register count, and the one-wave forms win.  r2 switched at 2048 waves; r3 measured the range in between
// (tools/r3_tail.sh, profiles/logs/r03_tail_sweep.txt): stacked correct at 73 728 filters 24.9 -> 21.4 us, fused frame +20 % from
// 69 632 to 114 688 filters.  r4: the threshold comes from the device (CU count) through the handle, FBUS_TWO_WAVE_MIN_B is read
// once at create.

#if FBUS_TU_FAMILY == 1
template <typename T, int N, int D>
void launch_predict_k(hipStream_t s, T* recs, int B, int K, int policy, const T* accel, const T* gyro, const T* dt,
                      int dt_stride, const DevConst<T>& dc, const LaunchPolicy& lp)
{
    const int grid = (B + BLOCK - 1) / BLOCK;
    // policy 0: nt loads and stores; 1: default-policy loads (first predict behind a kernel that stored the records with
    // the default policy); 2: default loads and stores (records do not fit the Infinity Cache) -- see predict_kernel
    if (K == 1) {
#define FBUS_LAUNCH_PREDICT(LD, ST)                                                                                     \
    hipLaunchKernelGGL((predict_kernel<T, N, D, false, LD, ST>), dim3(grid), dim3(BLOCK), 0, s, recs, B, K, accel, gyro, dt, \
                       dt_stride, dc)
        if (policy == 2) FBUS_LAUNCH_PREDICT(FBUS_X_PREDICT_LD_BIG, FBUS_X_PREDICT_ST_BIG);      // records larger than the Infinity Cache
        else if (policy == 1) FBUS_LAUNCH_PREDICT(FBUS_X_PREDICT_LD_WARM, AUX_NT);      // first predict behind a default-policy writer
        else FBUS_LAUNCH_PREDICT(FBUS_X_PREDICT_LD, FBUS_X_PREDICT_ST);
#undef FBUS_LAUNCH_PREDICT
    } else if constexpr (sizeof(T) == 8) {
        // fp64 (the reference's own arithmetic): K resident steps with rows p of the covariance and the whole nominal state parked in
        // LDS between their uses (StepPark; 512 registers, one wave per SIMD).  Rounds 1-3 ran predict_n as K launches of the per-call
        // kernel -- the resident loop spilled 580 bytes per lane; the parked form spills 68 (N = 18) / 0 (N = 15).
        hipLaunchKernelGGL((predict_kernel<T, N, D, true, AUX_NT, FBUS_X_PREDICT_ST, true>), dim3(grid), dim3(BLOCK), 0, s, recs, B,
                           K, accel, gyro, dt, dt_stride, dc);
    } else if (lp.two_wave(B)) {
        hipLaunchKernelGGL((predict_kernel<T, N, D, true, AUX_NT, FBUS_X_PREDICT_ST, true>), dim3(grid), dim3(BLOCK), 0, s, recs, B,
                           K, accel, gyro, dt, dt_stride, dc);
    } else {
        hipLaunchKernelGGL((predict_kernel<T, N, D, true>), dim3(grid), dim3(BLOCK), 0, s, recs, B, K, accel, gyro, dt,
                           dt_stride, dc);
    }
}
#define FBUS_INST(D)                                                                                                  \
    template void launch_predict_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, int, const FBUS_TU_T*,  \
                                                            const FBUS_TU_T*, const FBUS_TU_T*, int,                  \
                                                            const DevConst<FBUS_TU_T>&, const LaunchPolicy&);

#elif FBUS_TU_FAMILY == 2
template <typename T, int N, int D>
void launch_correct_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* pos, const T* quat, int mode,
                      bool joseph, const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc, const LaunchPolicy& lp)
{
    const int grid = (B + BLOCK - 1) / BLOCK;
    const bool joint = mode == MODE_STACKED;
    // measurement inputs as 16-byte loads where that is legal (groups of four markers, aligned arrays)
    if (M % 4 == 0 && ((reinterpret_cast<uintptr_t>(ids) | reinterpret_cast<uintptr_t>(pos) | reinterpret_cast<uintptr_t>(quat)) & 15) == 0 &&
        lp.meas_vec)
        mode |= MODE_MEAS_VEC;
#define FBUS_LAUNCH_CORRECT(COV, JOINT)                                                                              \
    hipLaunchKernelGGL((correct_kernel<T, N, D, COV, JOINT>), dim3(grid), dim3(BLOCK), 0, s, recs, B, M, ids, pos, quat, \
                       mode, skip, applied, dc)
    // fp32, stacked, simple form: from 1025 waves on (two on some SIMDs) the row-split instantiation (194 registers) is the
    // faster one -- see the LEAN comment in correct_kernel
    if (sizeof(T) == 4 && joint && !joseph && lp.two_wave(B)) {
        hipLaunchKernelGGL((correct_kernel<T, N, D, COV_SIMPLE, true, true>), dim3(grid), dim3(BLOCK), 0, s, recs, B, M, ids,
                           pos, quat, mode, skip, applied, dc);
        return;
    }
    if (joseph) { if (joint) FBUS_LAUNCH_CORRECT(COV_JOSEPH, true); else FBUS_LAUNCH_CORRECT(COV_JOSEPH, false); }
    else        { if (joint) FBUS_LAUNCH_CORRECT(COV_SIMPLE, true); else FBUS_LAUNCH_CORRECT(COV_SIMPLE, false); }
#undef FBUS_LAUNCH_CORRECT
}
#define FBUS_INST(D)                                                                                                   \
    template void launch_correct_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const int*,             \
                                                            const FBUS_TU_T*, const FBUS_TU_T*, int, bool,             \
                                                            const unsigned char*, unsigned char*,                      \
                                                            const DevConst<FBUS_TU_T>&, const LaunchPolicy&);

#elif FBUS_TU_FAMILY == 3
template <typename T, int N, int D>
void launch_frame_k(hipStream_t s, T* recs, int B, int K, const T* accel, const T* gyro, const T* dt, int dt_stride,
                    int M, const int* ids, const T* pos, const T* quat, int mode, bool joseph,
                    const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc, const LaunchPolicy& lp)
{
    const int grid = (B + BLOCK - 1) / BLOCK;
    const bool joint = mode == MODE_STACKED;
    if constexpr (sizeof(T) == 8) {
        // fp64: one fused kernel, the parked predict loop + the row-split passes (frame2_kernel: 512 registers, 39 KiB of LDS, one wave
        // per SIMD); stacked mode, simple form -- the caller (fbus_ekf.hip::launch_frame_t) runs every other combination as predict_n + correct
        hipLaunchKernelGGL((frame2_kernel<T, N, D>), dim3(grid), dim3(BLOCK), 0, s, recs, B, K, accel, gyro, dt, dt_stride, M, ids,
                           pos, quat, skip, applied, dc);
        (void)joint; (void)joseph; (void)lp;
    } else {
#define FBUS_LAUNCH_FRAME(COV, JOINT)                                                                                 \
    hipLaunchKernelGGL((frame_kernel<T, N, D, COV, JOINT>), dim3(grid), dim3(BLOCK), 0, s, recs, B, K, accel, gyro, dt, \
                       dt_stride, M, ids, pos, quat, mode, skip, applied, dc)
    // (Joseph form, nearest marker) is not built as a fused kernel (7 Joseph rank-2 passes with the record resident
    // spilled 280 bytes per lane): fbus_ekf.hip runs that combination as predict_n + correct
    // stacked mode, simple form, > 1024 waves: the two-waves-per-SIMD kernel (see frame2_kernel)
    if (joint && !joseph && lp.two_wave(B)) {
        hipLaunchKernelGGL((frame2_kernel<T, N, D>), dim3(grid), dim3(BLOCK), 0, s, recs, B, K, accel, gyro, dt, dt_stride, M, ids,
                           pos, quat, skip, applied, dc);
        return;
    }
    if (joseph) { FBUS_LAUNCH_FRAME(COV_JOSEPH, true); }
    else        { if (joint) FBUS_LAUNCH_FRAME(COV_SIMPLE, true); else FBUS_LAUNCH_FRAME(COV_SIMPLE, false); }
#undef FBUS_LAUNCH_FRAME
    }
}
#define FBUS_INST(D)                                                                                                  \
    template void launch_frame_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const FBUS_TU_T*,        \
                                                          const FBUS_TU_T*, const FBUS_TU_T*, int, int, const int*,   \
                                                          const FBUS_TU_T*, const FBUS_TU_T*, int, bool,              \
                                                          const unsigned char*, unsigned char*,                       \
                                                          const DevConst<FBUS_TU_T>&, const LaunchPolicy&);

#elif FBUS_TU_FAMILY == 5
template <typename T, int N, int D>
void launch_frames_k(hipStream_t s, T* recs, int B, int F, const unsigned char* kcount, const T* accel, const T* gyro,
                     const T* dt, int dt_stride, int M, const int* ids, const T* pos, const T* quat, int mode, bool joseph,
                     const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc)
{
    const int grid = (B + BLOCK - 1) / BLOCK;
    const bool joint = mode == MODE_STACKED;
    FrameCounts kc;
    for (int f = 0; f < FBUS_MAX_WINDOW_FRAMES; ++f) kc.k[f] = f < F ? kcount[f] : 0;
#define FBUS_LAUNCH_FRAMES(COV, JOINT)                                                                                \
    hipLaunchKernelGGL((frames_kernel<T, N, D, COV, JOINT>), dim3(grid), dim3(BLOCK), 0, s, recs, B, F, kc, accel, gyro, dt, \
                       dt_stride, M, ids, pos, quat, mode, skip, applied, dc)
    // (Joseph form, nearest marker) is not built with the record resident, as for frame_kernel: the caller runs that
    // combination frame by frame
    if (joseph) { FBUS_LAUNCH_FRAMES(COV_JOSEPH, true); }
    else        { if (joint) FBUS_LAUNCH_FRAMES(COV_SIMPLE, true); else FBUS_LAUNCH_FRAMES(COV_SIMPLE, false); }
#undef FBUS_LAUNCH_FRAMES
}
#define FBUS_INST(D)                                                                                                  \
    template void launch_frames_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const unsigned char*,   \
                                                           const FBUS_TU_T*, const FBUS_TU_T*, const FBUS_TU_T*, int,  \
                                                           int, const int*, const FBUS_TU_T*, const FBUS_TU_T*, int,   \
                                                           bool, const unsigned char*, unsigned char*,                 \
                                                           const DevConst<FBUS_TU_T>&);
#elif FBUS_TU_FAMILY == 6
// team kernels (ekf_team.hpp): fp32 only
template <typename T, int N, int D>
void launch_predict_team_k(hipStream_t s, T* recs, int B, int K, int roles, int policy, const T* accel, const T* gyro,
                           const T* dt, int dt_stride, const DevConst<T>& dc)
{
    const int tiles = (B + 63) / 64;
    if (K > 1) {
        hipLaunchKernelGGL((predict_n_team_kernel<T, N, D>), dim3(tiles), dim3(256), 0, s, recs, B, K, accel, gyro, dt, dt_stride, dc);
        return;
    }
    // cache policy of the record accesses as in launch_predict_k: 0 nt / nt, 1 default loads, 2 default loads and stores
#define FBUS_LAUNCH_PT(NR, LD, ST)                                                                                         \
    hipLaunchKernelGGL((predict_team_kernel<T, N, D, NR, LD, ST>), dim3(tiles), dim3(64 * NR), 0, s, recs, B, accel, gyro, dt, \
                       dt_stride, dc)
#define FBUS_LAUNCH_PT_POL(NR)                                                                                             \
    do {                                                                                                                   \
        if (policy == 2) FBUS_LAUNCH_PT(NR, AUX_DEFAULT, AUX_DEFAULT);                                                     \
        else if (policy == 1) FBUS_LAUNCH_PT(NR, AUX_DEFAULT, AUX_NT);                                                     \
        else FBUS_LAUNCH_PT(NR, AUX_NT, AUX_NT);                                                                           \
    } while (0)
    if (roles <= 2) FBUS_LAUNCH_PT_POL(2);
    else if (roles == 3) FBUS_LAUNCH_PT_POL(3);
    else FBUS_LAUNCH_PT_POL(4);
#undef FBUS_LAUNCH_PT_POL
#undef FBUS_LAUNCH_PT
}
template <typename T, int N, int D>
void launch_frames_team_k(hipStream_t s, T* recs, int B, int F, const unsigned char* kcount, const T* accel, const T* gyro,
                          const T* dt, int dt_stride, int M, const int* ids, const T* pos, const T* quat, int mode,
                          const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc)
{
    const int tiles = (B + 63) / 64;
    FrameCounts kc;
    for (int f = 0; f < FBUS_MAX_WINDOW_FRAMES; ++f) kc.k[f] = f < F ? kcount[f] : 0;
    // the image of the covariance, W and the exchange buffers: 80 KiB of LDS per workgroup, above the 64 KiB a kernel gets without asking
    constexpr size_t lds = FrameImage<T, N>::bytes();
    // (a per-device attribute: asked for once on every device this process launches the kernel on)
    static std::atomic<unsigned long long> asked{ 0 };
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(asked.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&frames_team_kernel<T, N, D>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            asked.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((frames_team_kernel<T, N, D>), dim3(tiles), dim3(256), lds, s, recs, B, F, kc, accel, gyro, dt, dt_stride, M,
                       ids, pos, quat, mode, skip, applied, dc);
}
#define FBUS_INST(D)                                                                                                   \
    template void launch_frames_team_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const unsigned char*, \
                                                                const FBUS_TU_T*, const FBUS_TU_T*, const FBUS_TU_T*, int, \
                                                                int, const int*, const FBUS_TU_T*, const FBUS_TU_T*, int, \
                                                                const unsigned char*, unsigned char*,                  \
                                                                const DevConst<FBUS_TU_T>&);                           \
    template void launch_predict_team_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, int, int, const FBUS_TU_T*, \
                                                                 const FBUS_TU_T*, const FBUS_TU_T*, int,              \
                                                                 const DevConst<FBUS_TU_T>&);
#elif FBUS_TU_FAMILY == 7
template <typename T, int N, int D>
void launch_pixels2_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* left, const T* right, int roles, double size,
                      double r_pix, const unsigned char* skip, unsigned char* applied, const short* id2slot, const MeasConst& mc)
{
    const int tiles = (B + 63) / 64;
    // the port square to the camera (normal exactly (0, 0, 1): the reference's configuration) has its own, shorter fold
    const bool nz = mc.n[0] == 0.0 && mc.n[1] == 0.0 && mc.n[2] == 1.0;
#define FBUS_LAUNCH_PX(NR)                                                                                               \
    do {                                                                                                                 \
        if (nz) hipLaunchKernelGGL((correct_pixels2_kernel<T, N, NR, true>), dim3(tiles), dim3(64 * NR), 0, s, recs, B, M, ids, left, \
                                   right, size, r_pix, skip, applied, id2slot, mc);                                      \
        else hipLaunchKernelGGL((correct_pixels2_kernel<T, N, NR, false>), dim3(tiles), dim3(64 * NR), 0, s, recs, B, M, ids, left, \
                                right, size, r_pix, skip, applied, id2slot, mc);                                         \
    } while (0)
    if (roles >= 3) { FBUS_LAUNCH_PX(4); return; }
    if (roles == 2) { FBUS_LAUNCH_PX(2); return; }
    {
        // (round 6) one wave per tile, square port -- the full-chip production case: the left-camera and the stereo update as kernels of their
        // own (CAM = 1 / 2: neither carries the other's image points and register pressure; EXPERIMENTS -1.7; fp64 records too: 16 slots
        // 92 -> 85-90 us left, 137 -> 125-132 us stereo, 4 slots stereo 69 -> 63 us, profiles/r06_f64_cam_ab.txt)
        if (nz) {
            if (right) hipLaunchKernelGGL((correct_pixels2_kernel<T, N, 1, true, 2>), dim3(tiles), dim3(64), 0, s, recs, B, M, ids, left, right,
                                          size, r_pix, skip, applied, id2slot, mc);
            else hipLaunchKernelGGL((correct_pixels2_kernel<T, N, 1, true, 1>), dim3(tiles), dim3(64), 0, s, recs, B, M, ids, left, right, size,
                                    r_pix, skip, applied, id2slot, mc);
            return;
        }
    }
    FBUS_LAUNCH_PX(1);
#undef FBUS_LAUNCH_PX
}
template <typename T, int N, int D>
void launch_corners2_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* left, const T* right, int geometry, int mode,
                       int roles, double size, double r_pos, double switch_thres, const unsigned char* skip, unsigned char* applied,
                       const short* id2slot, const MeasConst& mc, const VisConst<double>& vc, const VisConst<T>& vct)
{
    const int tiles = (B + 63) / 64;
    if (mode != MODE_STACKED) roles = 1;
    const bool nz = vc.nrm[0] == 0.0 && vc.nrm[1] == 0.0 && vc.nrm[2] == 1.0;       // the port square to the camera: the shorter triangulation
#define FBUS_LAUNCH_CR(NR)                                                                                               \
    do {                                                                                                                 \
        if (nz) hipLaunchKernelGGL((correct_corners2_kernel<T, N, NR, true>), dim3(tiles), dim3(64 * NR), 0, s, recs, B, M, ids, left, \
                                   right, geometry, mode, D, size, r_pos, switch_thres, skip, applied, id2slot, mc, vc, vct); \
        else hipLaunchKernelGGL((correct_corners2_kernel<T, N, NR, false>), dim3(tiles), dim3(64 * NR), 0, s, recs, B, M, ids, left, \
                                right, geometry, mode, D, size, r_pos, switch_thres, skip, applied, id2slot, mc, vc, vct); \
    } while (0)
    if (roles >= 3) FBUS_LAUNCH_CR(4);
    else if (roles == 2) FBUS_LAUNCH_CR(2);
    else FBUS_LAUNCH_CR(1);
#undef FBUS_LAUNCH_CR
}
#define FBUS_INST(D)                                                                                                   \
    template void launch_pixels2_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const int*, const FBUS_TU_T*, \
                                                            const FBUS_TU_T*, int, double, double, const unsigned char*, \
                                                            unsigned char*, const short*, const MeasConst&);           \
    template void launch_corners2_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const int*, const FBUS_TU_T*, \
                                                             const FBUS_TU_T*, int, int, int, double, double, double,  \
                                                             const unsigned char*, unsigned char*, const short*,       \
                                                             const MeasConst&, const VisConst<double>&, const VisConst<FBUS_TU_T>&);
#elif FBUS_TU_FAMILY == 8
template <typename T, int N, int D>
void launch_frame_meas_k(hipStream_t s, T* recs, int B, int F, const unsigned char* kcount, const T* accel, const T* gyro, const T* dt,
                         int dt_stride, int kind, int M, const int* ids, const T* left, const T* right, int geometry, int mode, double size,
                         double r_meas, double switch_thres, const unsigned char* skip, unsigned char* applied, const short* id2slot,
                         const MeasConst& mc, const VisConst<double>& vc, const VisConst<T>& vct, const T* qd)
{
    const int tiles = (B + 63) / 64;
    FrameCounts kc;
    for (int f = 0; f < FBUS_MAX_WINDOW_FRAMES; ++f) kc.k[f] = f < F ? kcount[f] : 0;
    QDiag<T> q;
    for (int i = 0; i < 4; ++i) q.qd[i] = qd[i];
    // the port square to the camera (the reference's configuration): the shorter fold / triangulation, as the per-call launchers choose
    const bool nz = (kind == MEAS_PIXELS) ? (mc.n[0] == 0.0 && mc.n[1] == 0.0 && mc.n[2] == 1.0)
                                          : (vc.nrm[0] == 0.0 && vc.nrm[1] == 0.0 && vc.nrm[2] == 1.0);
#define FBUS_LAUNCH_FM(KIND, NZF, CAM)                                                                                   \
    do {                                                                                                                 \
        if (F > 1) hipLaunchKernelGGL((frame_meas_kernel<T, N, D, KIND, NZF, true, CAM>), dim3(tiles), dim3(64), 0, s, recs, B, F, kc, accel, \
                                      gyro, dt, dt_stride, M, ids, left, right, geometry, mode, size, r_meas, switch_thres, skip, \
                                      applied, id2slot, mc, vc, vct, q);                                                 \
        else hipLaunchKernelGGL((frame_meas_kernel<T, N, D, KIND, NZF, false, CAM>), dim3(tiles), dim3(64), 0, s, recs, B, F, kc, accel, \
                                gyro, dt, dt_stride, M, ids, left, right, geometry, mode, size, r_meas, switch_thres, skip,  \
                                applied, id2slot, mc, vc, vct, q);                                                       \
    } while (0)
    // (round 6) pixel rows, square port: the left-camera and the stereo frame as kernels of their own (CAM = 1 / 2, as correct_pixels2_kernel)
    if (kind == MEAS_PIXELS) {
        if (nz) {
            // the left-camera and the stereo frame as kernels of their own (CAM = 1 / 2: +2-5 % for the windows and the stereo frame; the single
            // left-camera frame ran 1 % faster in the combined kernel until the fp32 start (EXPERIMENTS -1.10) pushed that kernel into 28 bytes of
            // scratch -- level since, profiles/r06_port_tangent_ab.txt, so the combined kernel is no longer built for the square port)
            if (right) FBUS_LAUNCH_FM(MEAS_PIXELS, true, 2);
            else FBUS_LAUNCH_FM(MEAS_PIXELS, true, 1);
        } else FBUS_LAUNCH_FM(MEAS_PIXELS, false, 0);
    }
    else                     { if (nz) FBUS_LAUNCH_FM(MEAS_CORNERS, true, 0); else FBUS_LAUNCH_FM(MEAS_CORNERS, false, 0); }
#undef FBUS_LAUNCH_FM
}
#define FBUS_INST(D)                                                                                                   \
    template void launch_frame_meas_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const unsigned char*, const FBUS_TU_T*, \
                                                               const FBUS_TU_T*, const FBUS_TU_T*, int, int, int, const int*, \
                                                               const FBUS_TU_T*, const FBUS_TU_T*, int, int, double, double, \
                                                               double, const unsigned char*, unsigned char*, const short*, \
                                                               const MeasConst&, const VisConst<double>&,               \
                                                               const VisConst<FBUS_TU_T>&, const FBUS_TU_T*);
#elif FBUS_TU_FAMILY == 9
template <typename T, int N, int D>
void launch_pixels_split_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* left, const T* right, int roles, double size,
                           double r_pix, const unsigned char* skip, unsigned char* applied, const short* id2slot, const MeasConst& mc)
{
    const int tiles = (B + 63) / 64;
    if (roles >= 3)
        hipLaunchKernelGGL((correct_pixels_split_kernel<T, N, 4>), dim3(tiles), dim3(256), 0, s, recs, B, M, ids, left, right, size,
                           r_pix, skip, applied, id2slot, mc);
    else
        hipLaunchKernelGGL((correct_pixels_split_kernel<T, N, 2>), dim3(tiles), dim3(128), 0, s, recs, B, M, ids, left, right, size,
                           r_pix, skip, applied, id2slot, mc);
}
#define FBUS_INST(D)                                                                                                   \
    template void launch_pixels_split_k<FBUS_TU_T, FBUS_TU_N, D>(hipStream_t, FBUS_TU_T*, int, int, const int*, const FBUS_TU_T*, \
                                                                 const FBUS_TU_T*, int, double, double, const unsigned char*, \
                                                                 unsigned char*, const short*, const MeasConst&);
#else
#error "FBUS_TU_FAMILY must be 1, 2, 3, 5, 6, 7, 8 or 9"
#endif

FBUS_INST(DIALECT_MATLAB)
FBUS_INST(DIALECT_CPP)

}  // namespace fbus
