// batched_filter.hpp -- header-only C++ host class over the C ABI (include/fbus_ekf.h).
//
// fbus::BatchedFilter is the batched, device-resident counterpart of the reference's
// FBUSEKF::FILTER (C++/include/filter.hpp:60-230): it owns the state of B filters and
// exposes the two steps of the hot path under the names BASELINE.json asks for:
//
//   predict(accel, gyro, dt)          == FILTER::UpdateCovariance + UpdateNominalState
//                                        (C++/src/filter.cpp:588-616, 533-582) / ImuUpdate.m:36
//   correct(M, ids, pos, quat, mode)  == FILTER::ObservationUpdate (filter.cpp:622-754) / MeasureUpdate.m:37
//
// plus the reference's input conventions: IMUData-style samples (common.hpp:176-193) and
// DetectionResult-style marker poses (filter.hpp:39-57).  Errors are C++ exceptions carrying the
// library's message; the reference's "silent early return" cases stay silent (query applied()).
#pragma once
#include "../fbus_ekf.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace fbus {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

template <typename Real = float>
class BatchedFilter {
    static_assert(sizeof(Real) == 4 || sizeof(Real) == 8, "Real must be float or double");

public:
    enum class Mode { Nearest = FBUS_MODE_NEAREST, Stacked = FBUS_MODE_STACKED };

    BatchedFilter(int batch, const fbus_params& prm, int device = 0, int nstate = 18)
        : batch_(batch), nstate_(nstate)
    {
        check(fbus_ekf_create(&h_, &prm, batch, device, int(sizeof(Real)) * 8, nstate), "fbus_ekf_create");
    }
    explicit BatchedFilter(int batch, int dialect = FBUS_DIALECT_CPP, int device = 0, int nstate = 18)
        : BatchedFilter(batch, defaults(dialect), device, nstate) {}
    ~BatchedFilter() { fbus_ekf_destroy(h_); }
    BatchedFilter(const BatchedFilter&) = delete;
    BatchedFilter& operator=(const BatchedFilter&) = delete;

    static fbus_params defaults(int dialect)
    {
        fbus_params p;
        if (fbus_params_default(&p, dialect) != FBUS_OK) throw Error(FBUS_ERR_INVALID, "fbus_params_default");
        return p;
    }

    int batch() const { return batch_; }
    int nstate() const { return nstate_; }
    fbus_ekf_t handle() const { return h_; }

    // ---- state (host vectors; B x 19, B x 9, B x N x N, B) -------------------------------------
    void set_state(const Real* nominal, const Real* rot, const Real* P, const int32_t* prev_id = nullptr)
    { check(fbus_ekf_set_state(h_, nominal, rot, P, prev_id), "set_state"); }
    void get_state(Real* nominal, Real* rot, Real* P, int32_t* prev_id = nullptr) const
    { check(fbus_ekf_get_state(h_, nominal, rot, P, prev_id), "get_state"); }
    void reset_covariance() { check(fbus_ekf_reset_cov(h_), "reset_cov"); }

    // ---- hot path, host pointers (staged) ---------------------------------------------------------
    void predict(const Real* accel, const Real* gyro, Real dt)
    { check(fbus_ekf_predict(h_, accel, gyro, &dt, 0), "predict"); }
    void predict(const Real* accel, const Real* gyro, const Real* dt_per_filter)
    { check(fbus_ekf_predict(h_, accel, gyro, dt_per_filter, 1), "predict"); }
    void correct(int M, const int32_t* ids, const Real* pos, const Real* quat, Mode mode = Mode::Nearest,
                 const uint8_t* skip = nullptr)
    { check(fbus_ekf_correct(h_, M, ids, pos, quat, int(mode), skip), "correct"); }

    // ---- (round 6) the same host-pointer calls without the wait: arrays taken by value (pageable: copied into a pinned ring of the
    //      handle's on return; pinned: in place, unchanged until inputs_consumed() / sync()), nothing waits for the device -- one call
    //      per IMU sample as FILTER::SetImuData / BatchImuProcessing issue them (filter.cpp:24-55,505-516)
    void predict_async(const Real* accel, const Real* gyro, Real dt)
    { check(fbus_ekf_predict_async(h_, accel, gyro, &dt, 0), "predict_async"); }
    void predict_n_async(int K, const Real* accel, const Real* gyro, const Real* dt, bool dt_per_filter = false)
    { check(fbus_ekf_predict_n_async(h_, K, accel, gyro, dt, dt_per_filter), "predict_n_async"); }
    void correct_async(int M, const int32_t* ids, const Real* pos, const Real* quat, Mode mode = Mode::Nearest,
                       const uint8_t* skip = nullptr)
    { check(fbus_ekf_correct_async(h_, M, ids, pos, quat, int(mode), skip), "correct_async"); }
    void correct_pixels_async(int M, const int32_t* ids, const Real* left, const Real* right = nullptr, const uint8_t* skip = nullptr)
    { check(fbus_ekf_correct_pixels_async(h_, M, ids, left, right, skip), "correct_pixels_async"); }
    void inputs_consumed() { check(fbus_ekf_async_inputs_consumed(h_), "async_inputs_consumed"); }

    // ---- hot path, device pointers (no copies, asynchronous until sync()) ----------------------------
    void predict_dev(const Real* accel, const Real* gyro, const Real* dt, bool dt_per_filter = false)
    { check(fbus_ekf_predict_dev(h_, accel, gyro, dt, dt_per_filter), "predict_dev"); }
    void predict_n_dev(int K, const Real* accel, const Real* gyro, const Real* dt, bool dt_per_filter = false)
    { check(fbus_ekf_predict_n_dev(h_, K, accel, gyro, dt, dt_per_filter), "predict_n_dev"); }
    void correct_dev(int M, const int32_t* ids, const Real* pos, const Real* quat, Mode mode = Mode::Nearest,
                     const uint8_t* skip = nullptr)
    { check(fbus_ekf_correct_dev(h_, M, ids, pos, quat, int(mode), skip), "correct_dev"); }
    // one camera frame: K per-sample predicts then one correct (filter.cpp:232-235)
    void frame_dev(int K, const Real* accel, const Real* gyro, const Real* dt, int M, const int32_t* ids,
                   const Real* pos, const Real* quat, Mode mode = Mode::Nearest)
    { check(fbus_ekf_frame_dev(h_, K, accel, gyro, dt, 0, M, ids, pos, quat, int(mode), nullptr), "frame_dev"); }

    // the same frame in one launch, and a window of frames in one launch (records resident in registers; offline replay)
    void frame_fused_dev(int K, const Real* accel, const Real* gyro, const Real* dt, int M, const int32_t* ids,
                         const Real* pos, const Real* quat, Mode mode = Mode::Nearest)
    { check(fbus_ekf_frame_fused_dev(h_, K, accel, gyro, dt, 0, M, ids, pos, quat, int(mode), nullptr), "frame_fused_dev"); }
    void frames_fused_dev(const std::vector<int32_t>& kcount, const Real* accel, const Real* gyro, const Real* dt, int M,
                          const int32_t* ids, const Real* pos, const Real* quat, Mode mode = Mode::Nearest,
                          const uint8_t* skip = nullptr)
    {
        check(fbus_ekf_frames_fused_dev(h_, int(kcount.size()), kcount.data(), accel, gyro, dt, 0, M, ids, pos, quat, int(mode), skip),
              "frames_fused_dev");
    }

    // correct() from corner pixels (north-star extension): left / right (B, M, 8) normalised image points, right may be null
    void correct_pixels(int M, const int32_t* ids, const Real* left, const Real* right = nullptr, const uint8_t* skip = nullptr)
    { check(fbus_ekf_correct_pixels(h_, M, ids, left, right, skip), "correct_pixels"); }

    // correct() from stereo corners (north-star extension): left / right (B, M, 8) normalised image points, or left = (B, M, 12)
    // triangulated corners with geometry = FBUS_VIS_CORNERS3D; the triangulation of vision.cpp:472-618 runs on the device
    void correct_corners(int M, const int32_t* ids, const Real* left, const Real* right, int geometry, Mode mode = Mode::Nearest,
                         const uint8_t* skip = nullptr)
    { check(fbus_ekf_correct_corners(h_, M, ids, left, right, geometry, int(mode), skip), "correct_corners"); }

    // (round 5) one camera frame with the north star's update in ONE launch (device pointers): K predicts, then correct_pixels
    // (kind = FBUS_MEAS_PIXELS; right may be null = left camera) or correct_corners (FBUS_MEAS_CORNERS with its geometry / mode) --
    // filter.cpp:232-235 with the reprojection rows in place of the pose rows
    void frame_meas_fused_dev(int K, const Real* accel, const Real* gyro, const Real* dt, int kind, int M, const int32_t* ids,
                              const Real* left, const Real* right = nullptr, int geometry = FBUS_VIS_REFRACTIVE,
                              Mode mode = Mode::Stacked, const uint8_t* skip = nullptr)
    {
        check(fbus_ekf_frame_meas_fused_dev(h_, K, accel, gyro, dt, 0, kind, M, ids, left, right, geometry, int(mode), skip),
              "frame_meas_fused_dev");
    }

    // ... and a window of such frames in one launch (offline replay of recorded corners; measurements [F][B][M]...)
    void frames_meas_fused_dev(const std::vector<int32_t>& kcount, const Real* accel, const Real* gyro, const Real* dt, int kind, int M,
                               const int32_t* ids, const Real* left, const Real* right = nullptr, int geometry = FBUS_VIS_REFRACTIVE,
                               Mode mode = Mode::Stacked, const uint8_t* skip = nullptr)
    {
        check(fbus_ekf_frames_meas_fused_dev(h_, int(kcount.size()), kcount.data(), accel, gyro, dt, 0, kind, M, ids, left, right, geometry,
                                             int(mode), skip), "frames_meas_fused_dev");
    }

    // waves per 64-filter tile (fbus_ekf_set_team): 0 = chosen per launch, 1 = always one, 2..4 = always that many
    void set_team(int predict_roles, int correct_roles) { check(fbus_ekf_set_team(h_, predict_roles, correct_roles), "set_team"); }

    std::vector<uint8_t> applied() const
    {
        std::vector<uint8_t> a(batch_);
        check(fbus_ekf_get_applied(h_, a.data()), "get_applied");
        return a;
    }
    // hip_stream is the hipStream_t itself (nullptr = HIP's legacy default stream); FBUS_STREAM_OWN = the handle's own
    void set_stream(void* hip_stream) { check(fbus_ekf_set_stream(h_, hip_stream), "set_stream"); }
    void wait_stream(void* other) { check(fbus_ekf_wait_stream(h_, other), "wait_stream"); }
    void signal_stream(void* other) { check(fbus_ekf_signal_stream(h_, other), "signal_stream"); }
    void sync() { check(fbus_ekf_sync(h_), "sync"); }

private:
    void check(int rc, const char* where) const
    {
        if (rc == FBUS_OK) return;
        std::string msg = std::string(where) + ": " + fbus_status_string(rc);
        if (h_) { const char* d = fbus_ekf_last_error(h_); if (d && *d) msg += std::string(" (") + d + ")"; }
        throw Error(rc, msg);
    }
    fbus_ekf_t h_ = nullptr;
    int batch_, nstate_;
};

}  // namespace fbus
