// frame_batcher.hpp -- host-side front door of the live pipeline.
//
// Mirrors what FBUSEKF::FILTER does around predict/correct in the reference's filter thread
// (paths relative to the upstream repository, C++/src/filter.cpp):
//   set_imu()        == FILTER::SetImuData :24-55         EMA pre-filter (coefficient 0.1 against the last
//                                                          BUFFERED sample), push, drop the oldest 500 when
//                                                          the buffer exceeds 2000 samples
//   on_detections()  == one pass of the thread loop :229-235
//                         BatchImuProcessing :483-531      samples with start <= t <= end, dt = t - state time,
//                                                          used samples (and older ones) leave the buffer
//                         ObservationUpdate  :622-754      -> Filter::correct (marker choice + hysteresis run
//                                                          inside the kernel)
//   on_corner_pixels()  the same pass with the detected markers' corner pixels -> Filter::correct_pixels (north star)
// (round 6) set_async(true): the same calls through Filter::predict_async / correct_async / correct_pixels_async where the filter has
// them (fbus::BatchedFilter does: fbus_ekf_*_async, arguments taken by value, nothing waits for the device) -- on_detections() then
// returns as soon as the frame's steps are queued, as the reference's callers do (SetImuData / SetDetectionResult hand the data over and
// return, filter.cpp:24-65); results by get_state() are unchanged and bit-equal to the synchronous sequence.
// Pure sequencing: no filter arithmetic happens here.  `Filter` is fbus::BatchedFilter<Real> (or anything with
// the same predict/correct members, which is how the CPU test drives it with a recorder).  All B filters of the
// handle receive the same sensor stream (B hypotheses of one robot); for independent streams call the batched
// device entry points directly.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace fbus {

template <typename Real, class Filter>
class FrameBatcher {
public:
    struct Sample { double t; Real accel[3]; Real gyro[3]; };

    FrameBatcher(Filter& filter, int batch, double state_time, bool ema = true, std::size_t max_buffer = 2000,
                 std::size_t trim = 500)
        : f_(filter), B_(batch), t_state_(state_time), ema_(ema), max_(max_buffer), trim_(trim) {}

    // FILTER::SetImuData (filter.cpp:24-55)
    void set_imu(double t, const Real accel[3], const Real gyro[3])
    {
        Sample s;
        s.t = t;
        for (int i = 0; i < 3; ++i) { s.accel[i] = accel[i]; s.gyro[i] = gyro[i]; }
        if (ema_ && !buf_.empty()) {
            const Real c = Real(0.1);
            const Sample& last = buf_.back();
            for (int i = 0; i < 3; ++i) {
                s.accel[i] = last.accel[i] * (Real(1) - c) + accel[i] * c;
                s.gyro[i] = last.gyro[i] * (Real(1) - c) + gyro[i] * c;
            }
        }
        buf_.push_back(s);
        if (buf_.size() > max_) buf_.erase(buf_.begin(), buf_.begin() + static_cast<std::ptrdiff_t>(trim_));
    }

    // (round 5) the same pass with what the cameras SAW instead of marker poses: the corner pixels of the M detected markers (left:
    // M x 8 normalised image coordinates x0 y0 .. x3 y3, the corners.txt layout of vision.cpp:111-119; right: the same from the right
    // camera, or nullptr) -> BatchImuProcessing, then Filter::correct_pixels -- the north star's MeasureUpdate in the live loop
    int on_corner_pixels(double t_frame, int M, const std::int32_t* ids, const Real* left, const Real* right)
    {
        const int used = advance_to(t_frame);
        if (M > 0) {
            ids_.assign(static_cast<std::size_t>(B_) * M, 0);
            pos_.assign(static_cast<std::size_t>(B_) * M * 8, Real(0));
            quat_.assign(right ? static_cast<std::size_t>(B_) * M * 8 : 0, Real(0));
            for (int b = 0; b < B_; ++b)
                for (int m = 0; m < M; ++m) {
                    ids_[static_cast<std::size_t>(b) * M + m] = ids[m];
                    for (int i = 0; i < 8; ++i) {
                        pos_[(static_cast<std::size_t>(b) * M + m) * 8 + i] = left[8 * m + i];
                        if (right) quat_[(static_cast<std::size_t>(b) * M + m) * 8 + i] = right[8 * m + i];
                    }
                }
            do_correct_pixels(f_, M, ids_.data(), pos_.data(), right ? quat_.data() : nullptr, 0);
        }
        return used;
    }

    // One detection result list stamped t_frame: BatchImuProcessing then ObservationUpdate.
    // Returns the number of predict steps issued.
    template <typename Mode>
    int on_detections(double t_frame, int M, const std::int32_t* ids, const Real* pos, const Real* quat, Mode mode)
    {
        const int used = advance_to(t_frame);
        if (M > 0) {
            ids_.assign(static_cast<std::size_t>(B_) * M, 0);
            pos_.assign(static_cast<std::size_t>(B_) * M * 3, Real(0));
            quat_.assign(static_cast<std::size_t>(B_) * M * 4, Real(0));
            for (int b = 0; b < B_; ++b)
                for (int m = 0; m < M; ++m) {
                    ids_[static_cast<std::size_t>(b) * M + m] = ids[m];
                    for (int i = 0; i < 3; ++i) pos_[(static_cast<std::size_t>(b) * M + m) * 3 + i] = pos[3 * m + i];
                    for (int i = 0; i < 4; ++i) quat_[(static_cast<std::size_t>(b) * M + m) * 4 + i] = quat[4 * m + i];
                }
            do_correct(f_, M, ids_.data(), pos_.data(), quat_.data(), mode, 0);
        }
        return used;
    }

    void set_async(bool on) { async_ = on; }
    bool async() const { return async_; }
    double state_time() const { return t_state_; }
    void set_state_time(double t) { t_state_ = t; }         // after an init / reset (filter.cpp:378,465)
    std::size_t buffered() const { return buf_.size(); }

private:
    // the asynchronous member where the filter type has one AND set_async(true) was called, the synchronous one otherwise
    // (overload ranking: int beats long, so the first form is taken whenever it compiles)
    template <class F> auto do_predict(F& f, const Real* a, const Real* g, Real dt, int) -> decltype(f.predict_async(a, g, dt), void())
    { if (async_) f.predict_async(a, g, dt); else f.predict(a, g, dt); }
    template <class F> void do_predict(F& f, const Real* a, const Real* g, Real dt, long) { f.predict(a, g, dt); }
    template <class F, typename Mode>
    auto do_correct(F& f, int M, const std::int32_t* ids, const Real* pos, const Real* quat, Mode mode, int)
        -> decltype(f.correct_async(M, ids, pos, quat, mode, nullptr), void())
    { if (async_) f.correct_async(M, ids, pos, quat, mode, nullptr); else f.correct(M, ids, pos, quat, mode, nullptr); }
    template <class F, typename Mode>
    void do_correct(F& f, int M, const std::int32_t* ids, const Real* pos, const Real* quat, Mode mode, long) { f.correct(M, ids, pos, quat, mode, nullptr); }
    template <class F> auto do_correct_pixels(F& f, int M, const std::int32_t* ids, const Real* l, const Real* r, int)
        -> decltype(f.correct_pixels_async(M, ids, l, r, nullptr), void())
    { if (async_) f.correct_pixels_async(M, ids, l, r, nullptr); else f.correct_pixels(M, ids, l, r, nullptr); }
    template <class F> void do_correct_pixels(F& f, int M, const std::int32_t* ids, const Real* l, const Real* r, long) { f.correct_pixels(M, ids, l, r, nullptr); }

    // BatchImuProcessing (filter.cpp:483-531): the buffered samples with state time <= t <= t_frame, one predict each
    int advance_to(double t_frame)
    {
        int used = 0, consumed = 0;
        for (const Sample& s : buf_) {                      // filter.cpp:493-517
            if (s.t < t_state_) { ++consumed; continue; }
            if (s.t > t_frame) break;
            ++consumed;
            const Real dt = static_cast<Real>(s.t - t_state_);
            acc_.assign(static_cast<std::size_t>(B_) * 3, Real(0));
            gyr_.assign(static_cast<std::size_t>(B_) * 3, Real(0));
            for (int b = 0; b < B_; ++b)
                for (int i = 0; i < 3; ++i) { acc_[3 * b + i] = s.accel[i]; gyr_[3 * b + i] = s.gyro[i]; }
            do_predict(f_, acc_.data(), gyr_.data(), dt, 0);    // UpdateCovariance + UpdateNominalState
            t_state_ = s.t;                                 // filter.cpp:516
            ++used;
        }
        buf_.erase(buf_.begin(), buf_.begin() + consumed);  // ClearImuBuffer, filter.cpp:520
        return used;
    }

    Filter& f_;
    int B_;
    double t_state_;
    bool ema_;
    bool async_ = false;
    std::size_t max_, trim_;
    std::vector<Sample> buf_;
    std::vector<Real> acc_, gyr_, pos_, quat_;
    std::vector<std::int32_t> ids_;
};

}  // namespace fbus
