"""Host-side front door of the live pipeline, the Python twin of
include/fbus/frame_batcher.hpp.

Mirrors what FBUSEKF::FILTER does around predict/correct in the reference's filter thread (C++/src/filter.cpp):
  set_imu()        == FILTER::SetImuData :24-55      EMA pre-filter (coefficient 0.1 against the last BUFFERED sample),
                                                     push, drop the oldest `trim` samples when the buffer exceeds `max_buffer`
  on_detections()  == one pass of the thread loop :229-235
                        BatchImuProcessing :483-531   samples with start <= t <= end, dt = t - state time
                        ObservationUpdate  :622-754   -> filter.correct (marker choice + hysteresis run in the kernel)
Pure sequencing: no filter arithmetic happens here.  `flt` is a BatchedFilter (or anything with predict/correct); all B
filters of the handle receive the same sensor stream (B hypotheses of one robot).
"""
import numpy as np


class FrameBatcher:
    def __init__(self, flt, batch, state_time, ema=True, max_buffer=2000, trim=500, dtype=np.float32):
        self.flt, self.B, self.t_state = flt, int(batch), float(state_time)
        self.ema, self.max_buffer, self.trim, self.dtype = bool(ema), int(max_buffer), int(trim), dtype
        self.buf = []                                   # (t, accel[3], gyro[3])
        self.use_async = False                          # set_async(True): predict_async / correct_async / correct_pixels_async where the filter has them

    def set_async(self, on):
        """(round 6) queue the calls through the filter's *_async members (fbus_ekf_*_async: arguments by value, nothing waits for the
        device); results through get_state() are bit-equal to the synchronous sequence"""
        self.use_async = bool(on)

    def _call(self, name, *args):
        fn = getattr(self.flt, name + "_async", None) if self.use_async else None
        return (fn or getattr(self.flt, name))(*args)

    def set_imu(self, t, accel, gyro):
        """FILTER::SetImuData (filter.cpp:24-55)"""
        a, w = np.asarray(accel, self.dtype).copy(), np.asarray(gyro, self.dtype).copy()
        if self.ema and self.buf:
            c = self.dtype(0.1)
            a = self.buf[-1][1] * (self.dtype(1) - c) + a * c
            w = self.buf[-1][2] * (self.dtype(1) - c) + w * c
        self.buf.append((float(t), a, w))
        if len(self.buf) > self.max_buffer:
            del self.buf[:self.trim]

    def on_detections(self, t_frame, ids, pos, quat, mode):
        """One detection list stamped t_frame: BatchImuProcessing, then ObservationUpdate.  Returns the number of
        predict steps issued."""
        used = self._advance_to(t_frame)
        ids = np.asarray(ids, np.int32).reshape(-1)
        if ids.size:
            M = ids.size
            self._call("correct", np.tile(ids, (self.B, 1)), np.tile(np.asarray(pos, self.dtype).reshape(1, M, 3), (self.B, 1, 1)),
                             np.tile(np.asarray(quat, self.dtype).reshape(1, M, 4), (self.B, 1, 1)), mode)
        return used

    def on_corner_pixels(self, t_frame, ids, left, right=None):
        """(round 5) the same pass with what the cameras saw: the corner pixels of the detected markers (left / right: (M, 8) normalised
        image coordinates, the corners.txt layout) -> BatchImuProcessing, then filter.correct_pixels -- the north star's MeasureUpdate."""
        used = self._advance_to(t_frame)
        ids = np.asarray(ids, np.int32).reshape(-1)
        if ids.size:
            M = ids.size
            rgt = None if right is None else np.tile(np.asarray(right, self.dtype).reshape(1, M, 8), (self.B, 1, 1))
            self._call("correct_pixels", np.tile(ids, (self.B, 1)), np.tile(np.asarray(left, self.dtype).reshape(1, M, 8), (self.B, 1, 1)), rgt)
        return used

    def _advance_to(self, t_frame):
        """BatchImuProcessing (filter.cpp:483-531)"""
        used = consumed = 0
        for (ts, a, w) in self.buf:                     # filter.cpp:493-517
            if ts < self.t_state:
                consumed += 1
                continue
            if ts > t_frame:
                break
            consumed += 1
            dt = self.dtype(ts - self.t_state)
            self._call("predict", np.tile(a, (self.B, 1)), np.tile(w, (self.B, 1)), float(dt))
            self.t_state = ts                           # filter.cpp:516
            used += 1
        del self.buf[:consumed]                         # ClearImuBuffer, filter.cpp:520
        return used

    @property
    def buffered(self):
        return len(self.buf)
