// ekf_launch.hpp -- host-callable launchers of the four large kernel families.
//
// The library is built from several translation units so that the kernels compile in parallel (and a kernel under
// work rebuilds in seconds): kernels_tu.hip is compiled once per (scalar type, state size, family) with
// -DFBUS_TU_T / -DFBUS_TU_N / -DFBUS_TU_FAMILY and holds the explicit instantiations of the function templates
// declared here (both dialects); fbus_ekf.hip (handle, C ABI, the small kernels) only sees these declarations.
#pragma once
#include "ekf_device.hpp"
#include "vision_device.hpp"

#include <hip/hip_runtime.h>

namespace fbus {

// What the launchers need to know about the device and the process, decided ONCE per handle (fbus_ekf_create: device properties +
// environment overrides) -- nothing below reads the environment.
struct LaunchPolicy {
    int simds = 1024;               // SIMDs of the device (CUs x 4): one wave per SIMD = `simds` 64-filter tiles
    int two_wave_min_b = 1024 * 64 + 1;   // from this many filters on a launch has more waves than SIMDs: the <= 256-register forms
    bool meas_vec = true;           // measurement inputs of correct as 16-byte loads where legal
    int policy_b = 0;               // fbus_ekf_set_policy_batch: the batch the kernel-FORM choice is keyed on (0 = the launch's own B)
    // the <= 256-register forms (row-split correct, parked predict_n, frame2_kernel): chosen by the JOB's size, not the shard's, so
    // that every shard layout of one job runs the same instruction streams (the forms agree to 1 ulp only)
    bool two_wave(int B) const { return (policy_b > 0 ? policy_b : B) >= two_wave_min_b; }
};

// K == 1: the streamed per-call kernel (`policy`: 0 = nt loads and stores, 1 = default-policy loads, 2 = default loads
// and stores); K > 1: predict_n, K samples per launch with the record resident in registers
template <typename T, int N, int D>
void launch_predict_k(hipStream_t s, T* recs, int B, int K, int policy, const T* accel, const T* gyro, const T* dt,
                      int dt_stride, const DevConst<T>& dc, const LaunchPolicy& lp);

template <typename T, int N, int D>
void launch_correct_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* pos, const T* quat, int mode,
                      bool joseph, const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc, const LaunchPolicy& lp);

template <typename T, int N, int D>
void launch_frame_k(hipStream_t s, T* recs, int B, int K, const T* accel, const T* gyro, const T* dt, int dt_stride,
                    int M, const int* ids, const T* pos, const T* quat, int mode, bool joseph,
                    const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc, const LaunchPolicy& lp);

// a window of F frames in one launch (fp32; not (Joseph, nearest)); kcount: F host bytes
template <typename T, int N, int D>
void launch_frames_k(hipStream_t s, T* recs, int B, int F, const unsigned char* kcount, const T* accel, const T* gyro,
                     const T* dt, int dt_stride, int M, const int* ids, const T* pos, const T* quat, int mode, bool joseph,
                     const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc);

// ---- team kernels (ekf_team.hpp): several waves per 64-filter tile, fp32 only -------------------------------------------
// roles: waves per tile (predict 2..4, predict_n always 4); policy as in launch_predict_k
template <typename T, int N, int D>
void launch_predict_team_k(hipStream_t s, T* recs, int B, int K, int roles, int policy, const T* accel, const T* gyro,
                           const T* dt, int dt_stride, const DevConst<T>& dc);
template <typename T, int N, int D>
void launch_frames_team_k(hipStream_t s, T* recs, int B, int F, const unsigned char* kcount, const T* accel, const T* gyro,
                          const T* dt, int dt_stride, int M, const int* ids, const T* pos, const T* quat, int mode,
                          const unsigned char* skip, unsigned char* applied, const DevConst<T>& dc);

// ---- correct() from corner pixels, round-4 kernel (ekf_meas.hpp) ---------------------------------------------------------
constexpr int MKC_STRIDE = 10;      // doubles per map slot: corner 0 in the world (3), marker x axis (3), y axis (3), pad

// constants of the pixel fold (double; kernel argument)
struct MeasConst {
    double P_IL[3];
    double McL[9], McR[9], tR[3];   // a corner at t_I (IMU frame, t_I = R'(c_w - p) - P_IL) in the refraction frame of the left / right
                                    // camera:  XL = McL t_I,  XR = McR t_I + tR   (McL = F R_IL: vision.cpp:597-599 undone;
                                    // McR = R_RL^-1 McL, tR = -R_RL^-1 P_LR: vision.cpp:555-556 inverted)
    double nML[3], nMR[3];          // n' McL, n' McR
    double NI[6];                   // R_IL' R_IL (symmetric: 00 01 02 11 12 22): the N' of a corner's three position rows
    double n[3];                    // port normal
    double a0, a1, d_air, d_glass;  // n_air / n_glass, n_air / n_water
    double klim;                    // (round 6) (0.9 a1)^2 / (1 - a1^2): the field-of-view test of the fold (a division per marker when left to the kernel)
    double MiTL[9], adjL[9];        // (round 6) McL^-T and adj(McL) = det(McL) McL^-1: the camera-frame fold of the left-only kernels
                                    // (ekf_meas.hpp::pixel_fold_marker<..., CF = true>, PixAcc::to_imu_frame)
    float st[8];                    // (round 6) constants of the fp32 start of the port equation (ekf_meas.hpp::port_start_f32):
                                    // a1, a1^2, 1 - a1^2, 1 - a0^2, d_air, d_glass a0, (d_air + d_glass a0) / a1, 0
    const double* mkc;              // [FBUS_MAX_MARKERS][MKC_STRIDE]
};

// roles: waves per 64-filter tile (1, 2 or 4: the markers of a filter divided among them); right == nullptr: left camera only.
// The kernel does not depend on the dialect (the pixel rows have no quaternion part); D only keeps the instantiation macro uniform.
template <typename T, int N, int D>
void launch_pixels2_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* left, const T* right, int roles, double size,
                      double r_pix, const unsigned char* skip, unsigned char* applied, const short* id2slot, const MeasConst& mc);
// (round 5) the same update with the TAIL divided between the waves of a tile (ekf_meas_split.hpp): roles = 2 or 4 waves per tile
// (launches below half / a quarter of the chip's SIMDs in tiles).  fp32 records, square port only -- the caller checks.
template <typename T, int N, int D>
void launch_pixels_split_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* left, const T* right, int roles, double size,
                           double r_pix, const unsigned char* skip, unsigned char* applied, const short* id2slot, const MeasConst& mc);
// correct() from stereo corners, round-4 kernel: nearest marker (roles forced to 1) or stacked; dialect: hysteresis of the nearest mode
template <typename T, int N, int D>
void launch_corners2_k(hipStream_t s, T* recs, int B, int M, const int* ids, const T* left, const T* right, int geometry, int mode,
                       int roles, double size, double r_pos, double switch_thres, const unsigned char* skip, unsigned char* applied,
                       const short* id2slot, const MeasConst& mc, const VisConst<double>& vc, const VisConst<T>& vct);

// ---- one camera frame with the north star's MeasureUpdate in one launch (ekf_meas.hpp::frame_meas_kernel; fp32) -----------------
// kind: corner pixels (geometry / mode ignored; right == nullptr: left camera) or stereo corners (geometry, mode as correct_corners)
enum { MEAS_PIXELS = 0, MEAS_CORNERS = 1 };
// F = 1: one frame of kcount[0] predicts; F > 1: a window of F frames in one launch (kcount: F host bytes; measurements [F][B][M]...)
template <typename T, int N, int D>
void launch_frame_meas_k(hipStream_t s, T* recs, int B, int F, const unsigned char* kcount, const T* accel, const T* gyro, const T* dt, int dt_stride, int kind,
                         int M, const int* ids, const T* left, const T* right, int geometry, int mode, double size, double r_meas,
                         double switch_thres, const unsigned char* skip, unsigned char* applied, const short* id2slot,
                         const MeasConst& mc, const VisConst<double>& vc, const VisConst<T>& vct, const T* qd);

}  // namespace fbus
