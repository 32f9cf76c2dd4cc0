"""BatchedFilter: the reference filter's predict()/correct() contract for B filters on one GPU.

Mirrors `State = ImuUpdate(State, accel, gyro, dt)` (matlab/ImuUpdate.m:36) and
`State = MeasureUpdate(State, visionMeas, markerMap, cameraInfo)`
(matlab/MeasureUpdate.m:37) / FILTER::UpdateCovariance+UpdateNominalState and
FILTER::ObservationUpdate (C++/src/filter.cpp:588-616,533-582,622-754), with the
state owned by the object as in the C++ FILTER class.  All arithmetic happens in
libfbus_ekf.so; this class only converts arguments.

Host (numpy) arrays go through the staging entry points; device arrays (anything
with `.data_ptr()`, e.g. torch tensors on the handle's GPU) go to the `_dev` entry
points without copies.
"""
import ctypes as C

import numpy as np

from . import capi


def _is_dev(x):
    return hasattr(x, "data_ptr")


class BatchedFilter:
    def __init__(self, batch, params=None, dialect=capi.DIALECT_MATLAB, device=0, dtype=32, nstate=18,
                 stream=None, order_streams=True):
        self._lib = capi.load_library()
        self._h = C.c_void_p()
        self.params = params if params is not None else capi.default_params(dialect)
        self.B, self.device, self.dtype, self.N = int(batch), int(device), int(dtype), int(nstate)
        self.np_dtype = np.float32 if dtype == 32 else np.float64
        rc = self._lib.fbus_ekf_create_checked(C.byref(self._h), C.byref(self.params), C.sizeof(capi.FbusParams), capi.ABI_VERSION,
                                               self.B, self.device, self.dtype, self.N)
        if rc != 0:
            self._h = C.c_void_p()
            raise capi.FbusError(rc, "fbus_ekf_create", self._lib.fbus_status_string(rc).decode())
        self._keep = []          # device arrays that must outlive asynchronous launches
        # The handle starts on its own NON-BLOCKING stream: device arrays handed to predict/correct/frame must be
        # complete before the call and results are complete after sync() (or order the streams with
        # wait_stream()/signal_stream(), or share the caller's stream with set_stream()).
        self._own_stream = True
        # order_streams (default): while the handle runs on its own stream, every device-array call first makes that
        # stream wait for the caller's current torch stream (inputs produced there, e.g. an upload or a .to(dtype), are
        # complete before a kernel reads them) and afterwards makes the caller's stream wait for the call (results are
        # visible to work queued there).  A caller that synchronises by itself (bench.py's timed region) switches it
        # off: the two event markers per call cost launch-stream time.
        self.order_streams = bool(order_streams)
        self._capturing = False
        if stream is not None:
            self.set_stream(stream)

    # ---- plumbing -------------------------------------------------------------
    def _check(self, rc, where):
        if rc != 0:
            detail = self._lib.fbus_ekf_last_error(self._h).decode() or self._lib.fbus_status_string(rc).decode()
            raise capi.FbusError(rc, where, detail)

    def close(self):
        if self._h:
            self._lib.fbus_ekf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream):
        """stream: an int/hipStream_t handle or an object with `.cuda_stream` (torch.cuda.Stream); the handle is passed
        as it is -- 0 is HIP's legacy default stream (what torch.cuda.current_stream() is unless the caller switched
        streams), so the launches are ordered with the caller's own work.  stream=None restores the handle's own stream."""
        if stream is None:
            handle = capi.STREAM_OWN
        else:
            handle = int(getattr(stream, "cuda_stream", stream))
        self._check(self._lib.fbus_ekf_set_stream(self._h, C.c_void_p(handle)), "set_stream")
        self._own_stream = stream is None

    def set_team(self, predict_roles=0, correct_roles=0):
        """waves per 64-filter tile of predict / correct: 0 = chosen per launch (default), 1 = one wave per tile, 2..4 fixed.
        predict_roles also governs predict_n and the fused frame / frame window entry points (the four-role pipeline), correct_roles also
        correct_corners (stacked mode) and correct_pixels (2 = two waves per tile, 3..4 = four) -- include/fbus_ekf.h, DESIGN.md 4.5"""
        self._check(self._lib.fbus_ekf_set_team(self._h, int(predict_roles), int(correct_roles)), "set_team")

    def set_policy_batch(self, total_filters):
        """the batch the automatic kernel-family choice is keyed on: the WHOLE job when this handle holds one shard of it
        (every shard layout then runs the same kernels: bit-equal results); 0 = this handle's own batch"""
        self._check(self._lib.fbus_ekf_set_policy_batch(self._h, int(total_filters)), "fbus_ekf_set_policy_batch")

    def launch_info(self, what, arg=0):
        v = C.c_int(0)
        self._check(self._lib.fbus_ekf_launch_info(self._h, int(what), int(arg), C.byref(v)), "fbus_ekf_launch_info")
        return v.value

    def launch_policy(self, M=4, K=7):
        """the handle's launch policy as a dict (what bench.py records beside its numbers)"""
        c = capi
        return {"simds": self.launch_info(c.INFO_SIMDS), "one_round_filters": self.launch_info(c.INFO_ONE_ROUND_FILTERS),
                "two_wave_min_b": self.launch_info(c.INFO_TWO_WAVE_MIN_B), "big_records_MB": self.launch_info(c.INFO_BIG_RECORDS_MB),
                "mall_MB": self.launch_info(c.INFO_MALL_MB), "l2_KB": self.launch_info(c.INFO_L2_KB),
                "policy_batch": self.launch_info(c.INFO_POLICY_BATCH), "roles_predict": self.launch_info(c.INFO_ROLES_PREDICT, 1),
                "roles_predict_n": self.launch_info(c.INFO_ROLES_PREDICT, K), "roles_meas": self.launch_info(c.INFO_ROLES_MEAS, M),
                "team_frames": bool(self.launch_info(c.INFO_TEAM_FRAMES)), "meas_split": self.launch_info(c.INFO_MEAS_SPLIT, M)}

    def wait_stream(self, stream):
        """work submitted to this filter from now on starts after everything already queued on `stream`"""
        self._check(self._lib.fbus_ekf_wait_stream(self._h, C.c_void_p(int(getattr(stream, "cuda_stream", stream)))), "wait_stream")

    def signal_stream(self, stream):
        """work submitted to `stream` from now on starts after everything already queued on this filter"""
        self._check(self._lib.fbus_ekf_signal_stream(self._h, C.c_void_p(int(getattr(stream, "cuda_stream", stream)))), "signal_stream")

    def sync(self):
        self._check(self._lib.fbus_ekf_sync(self._h), "sync")
        self._keep.clear()

    def _order_in(self, *arrays):
        """own stream + order_streams: wait for the caller's current stream; returns it for _order_out (else None)"""
        if not (self._own_stream and self.order_streams) or self._capturing:
            return None
        dev = next((a for a in arrays if a is not None and _is_dev(a) and hasattr(a, "device")), None)
        if dev is None:
            return None
        try:
            import torch
            cur = torch.cuda.current_stream(dev.device)
        except Exception:           # a non-torch device array: the caller orders the streams
            return None
        self.wait_stream(cur)
        return cur

    def _order_out(self, cur):
        if cur is not None:
            self.signal_stream(cur)

    def _host(self, a, shape, dtype=None):
        a = np.ascontiguousarray(a, dtype or self.np_dtype)
        if a.size != int(np.prod(shape)):
            raise ValueError(f"expected {shape}, got {a.shape}")
        return a

    @staticmethod
    def _p(a):
        if a is None:
            return None
        if _is_dev(a):
            return C.c_void_p(a.data_ptr())
        return a.ctypes.data_as(C.c_void_p)

    def _dev_checked(self, a, numel, what):
        if a.numel() != numel:
            raise ValueError(f"{what}: expected {numel} elements, got {a.numel()}")
        if hasattr(a, "is_contiguous") and not a.is_contiguous():
            raise ValueError(f"{what}: device array must be contiguous")
        self._keep.append(a)
        return a

    # ---- state ----------------------------------------------------------------
    def set_state(self, nominal=None, rot=None, P=None, prev_id=None):
        B, N = self.B, self.N
        if any(_is_dev(x) for x in (nominal, rot, P, prev_id) if x is not None):
            cur = self._order_in(nominal, rot, P, prev_id)
            rc = self._lib.fbus_ekf_set_state_dev(self._h, self._p(nominal), self._p(rot), self._p(P), self._p(prev_id))
            self._check(rc, "set_state_dev")
            return self._order_out(cur)
        nominal = None if nominal is None else self._host(nominal, (B, 19))
        rot = None if rot is None else self._host(rot, (B, 9))
        P = None if P is None else self._host(P, (B, N, N))
        prev_id = None if prev_id is None else self._host(prev_id, (B,), np.int32)
        rc = self._lib.fbus_ekf_set_state(self._h, self._p(nominal), self._p(rot), self._p(P), self._p(prev_id))
        self._check(rc, "set_state")

    def get_state(self):
        B, N = self.B, self.N
        nominal = np.empty((B, 19), self.np_dtype)
        rot = np.empty((B, 9), self.np_dtype)
        P = np.empty((B, N, N), self.np_dtype)
        prev = np.empty(B, np.int32)
        rc = self._lib.fbus_ekf_get_state(self._h, self._p(nominal), self._p(rot), self._p(P), self._p(prev))
        self._check(rc, "get_state")
        return nominal, rot, P, prev

    def reset_cov(self):
        self._check(self._lib.fbus_ekf_reset_cov(self._h), "reset_cov")

    def records(self):
        """(device pointer, bytes per filter, total bytes) of the packed records."""
        ptr, bpf, tot = C.c_void_p(), C.c_size_t(), C.c_size_t()
        self._check(self._lib.fbus_ekf_records(self._h, C.byref(ptr), C.byref(bpf), C.byref(tot)), "records")
        return ptr.value, bpf.value, tot.value

    def attach_records(self, dev_array):
        """Make the handle keep its records inside a caller-owned device array (e.g. a torch uint8 tensor)."""
        nbytes = dev_array.numel() * dev_array.element_size()
        self._check(self._lib.fbus_ekf_attach_records(self._h, self._p(dev_array), nbytes), "attach_records")
        self._records_owner = dev_array

    # ---- multi-GPU: the one collective (RCCL inside the library) -------------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte ncclUniqueId (bytes): rank 0 creates it, every rank passes it to comm_init"""
        lib = capi.load_library()
        buf = C.create_string_buffer(128)
        rc = lib.fbus_ekf_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise capi.FbusError(rc, "comm_unique_id", lib.fbus_status_string(rc).decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self._lib.fbus_ekf_comm_init(self._h, C.cast(buf, C.c_void_p), int(rank), int(world)), "comm_init")

    def copy_records(self, dst, dst_device=0, byte_offset=0):
        """this handle's packed records into device memory `dst` (+ byte_offset) on `dst_device`, on the handle's stream
        (fbus_ekf_copy_records: the peer-copy form of the gather, what fbus::NodeFilter::gather_to issues per shard).
        dst: a device tensor or a raw device pointer (int)."""
        ptr = dst if isinstance(dst, int) else dst.data_ptr()
        if not isinstance(dst, int):
            self._keep.append(dst)
        self._check(self._lib.fbus_ekf_copy_records(self._h, C.c_void_p(ptr + int(byte_offset)), int(dst_device)), "copy_records")

    def gather(self, out, bytes_of_rank=None):
        """all ranks' packed records into the device array `out` (uint8, sum of the ranks' record bytes) on every rank"""
        arr = None
        if bytes_of_rank is not None:
            arr = (C.c_size_t * len(bytes_of_rank))(*[int(x) for x in bytes_of_rank])
        self._keep.append(out)
        self._check(self._lib.fbus_ekf_gather(self._h, self._p(out), arr), "gather")

    # ---- predict == ImuUpdate -----------------------------------------------------
    def predict(self, accel, gyro, dt):
        return self.predict_n(accel, gyro, dt, K=1)

    def predict_n(self, accel, gyro, dt, K=None):
        B = self.B
        if _is_dev(accel):
            K = K if K is not None else accel.numel() // (3 * B)
            per = 1 if dt.numel() == K * B and not (B == 1 and dt.numel() == K) else 0
            if not per and dt.numel() != K:
                raise ValueError("dt must have K or K*B elements")
            self._dev_checked(accel, K * B * 3, "accel"); self._dev_checked(gyro, K * B * 3, "gyro")
            self._dev_checked(dt, dt.numel(), "dt")
            cur = self._order_in(accel, gyro, dt)
            rc = self._lib.fbus_ekf_predict_n_dev(self._h, K, self._p(accel), self._p(gyro), self._p(dt), per)
            self._check(rc, "predict_n_dev")
            return self._order_out(cur)
        accel = np.ascontiguousarray(accel, self.np_dtype)
        K = K if K is not None else accel.size // (3 * B)
        accel = self._host(accel, (K, B, 3))
        gyro = self._host(gyro, (K, B, 3))
        dt = np.ascontiguousarray(np.atleast_1d(dt), self.np_dtype)
        per = 1 if (dt.size == K * B and B > 1) else 0
        if not per and dt.size != K:
            raise ValueError("dt must have K or K*B elements")
        rc = self._lib.fbus_ekf_predict_n(self._h, K, self._p(accel), self._p(gyro), self._p(dt), per)
        self._check(rc, "predict_n")

    # ---- the host-pointer calls without the wait (fbus_ekf_*_async) --------------------------------
    def _host_any(self, x, shape, dtype=None):
        """numpy array or a CPU torch tensor (pinned tensors are transferred in place by the library)"""
        if _is_dev(x):
            if x.is_cuda:
                raise ValueError("the _async calls take HOST arrays (device arrays go to predict / correct: already asynchronous)")
            x = x.numpy()              # shares the (possibly pinned) memory
        return self._host(x, shape, dtype)

    def predict_async(self, accel, gyro, dt, K=1):
        """fbus_ekf_predict_n_async: host arrays taken by value, nothing waits for the device (results: sync() / get_state())"""
        B = self.B
        accel, gyro = self._host_any(accel, (K, B, 3)), self._host_any(gyro, (K, B, 3))
        dt = np.ascontiguousarray(np.atleast_1d(dt), self.np_dtype)
        per = 1 if (dt.size == K * B and B > 1) else 0
        if not per and dt.size != K:
            raise ValueError("dt must have K or K*B elements")
        self._check(self._lib.fbus_ekf_predict_n_async(self._h, K, self._p(accel), self._p(gyro), self._p(dt), per), "predict_n_async")

    def correct_async(self, ids, pos, quat, mode=capi.MODE_NEAREST, skip=None):
        B = self.B
        ids = np.ascontiguousarray(ids.numpy() if _is_dev(ids) else ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        pos, quat = self._host_any(pos, (B, M, 3)), self._host_any(quat, (B, M, 4))
        skip = None if skip is None else self._host(skip, (B,), np.uint8)
        self._check(self._lib.fbus_ekf_correct_async(self._h, M, self._p(ids), self._p(pos), self._p(quat), mode, self._p(skip)), "correct_async")

    def correct_pixels_async(self, ids, left, right=None, skip=None):
        B = self.B
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        left = self._host_any(left, (B, M, 8))
        right = None if right is None else self._host_any(right, (B, M, 8))
        skip = None if skip is None else self._host(skip, (B,), np.uint8)
        self._check(self._lib.fbus_ekf_correct_pixels_async(self._h, M, self._p(ids), self._p(left), self._p(right), self._p(skip)),
                    "correct_pixels_async")

    def async_inputs_consumed(self):
        """every H2D copy of the _async calls so far is done: pinned input arrays may be rewritten"""
        self._check(self._lib.fbus_ekf_async_inputs_consumed(self._h), "async_inputs_consumed")

    def async_stats(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self._lib.fbus_ekf_async_stats(self._h, C.byref(a), C.byref(b), C.byref(c)), "async_stats")
        return {"calls": a.value, "waits": b.value, "direct_pieces": c.value}

    # ---- correct == MeasureUpdate -----------------------------------------------------
    def correct(self, ids, pos, quat, mode=capi.MODE_NEAREST, skip=None):
        B = self.B
        if _is_dev(ids):
            M = ids.numel() // B
            self._dev_checked(ids, B * M, "ids"); self._dev_checked(pos, B * M * 3, "pos")
            self._dev_checked(quat, B * M * 4, "quat")
            if skip is not None:
                self._dev_checked(skip, B, "skip")
            cur = self._order_in(ids, pos, quat, skip)
            rc = self._lib.fbus_ekf_correct_dev(self._h, M, self._p(ids), self._p(pos), self._p(quat), mode, self._p(skip))
            self._check(rc, "correct_dev")
            return self._order_out(cur)
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        pos = self._host(pos, (B, M, 3))
        quat = self._host(quat, (B, M, 4))
        skip = None if skip is None else self._host(skip, (B,), np.uint8)
        rc = self._lib.fbus_ekf_correct(self._h, M, self._p(ids), self._p(pos), self._p(quat), mode, self._p(skip))
        self._check(rc, "correct")

    def correct_corners(self, ids, left, right=None, geometry=capi.VIS_REFRACTIVE, mode=capi.MODE_NEAREST, skip=None):
        """correct() from stereo corners (north-star extension, no reference counterpart): the corners are
        triangulated on the device and each corner position is a 3-row measurement (12 rows per marker).
        left/right: (B, M, 8) normalised corner coordinates, or left = (B, M, 12) with VIS_CORNERS3D."""
        B = self.B
        w = 12 if geometry == capi.VIS_CORNERS3D else 8
        if _is_dev(ids):
            M = ids.numel() // B
            self._dev_checked(ids, B * M, "ids"); self._dev_checked(left, B * M * w, "left")
            if right is not None:
                self._dev_checked(right, B * M * 8, "right")
            if skip is not None:
                self._dev_checked(skip, B, "skip")
            cur = self._order_in(ids, left, right, skip)
            rc = self._lib.fbus_ekf_correct_corners_dev(self._h, M, self._p(ids), self._p(left), self._p(right),
                                                        geometry, mode, self._p(skip))
            self._check(rc, "correct_corners_dev")
            return self._order_out(cur)
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        left = self._host(left, (B, M, w))
        right = None if right is None else self._host(right, (B, M, 8))
        skip = None if skip is None else self._host(skip, (B,), np.uint8)
        rc = self._lib.fbus_ekf_correct_corners(self._h, M, self._p(ids), self._p(left), self._p(right), geometry, mode,
                                                self._p(skip))
        self._check(rc, "correct_corners")

    def correct_pixels(self, ids, left, right=None, skip=None):
        """correct() from corner PIXELS (north-star extension, no reference counterpart): the flat-port reprojection of
        the four corners of every visible marker, 2 rows per corner (left camera) or 4 (left and right).
        left/right: (B, M, 8) normalised image points x0 y0 .. x3 y3."""
        B = self.B
        if _is_dev(ids):
            M = ids.numel() // B
            self._dev_checked(ids, B * M, "ids"); self._dev_checked(left, B * M * 8, "left")
            if right is not None:
                self._dev_checked(right, B * M * 8, "right")
            if skip is not None:
                self._dev_checked(skip, B, "skip")
            cur = self._order_in(ids, left, right, skip)
            rc = self._lib.fbus_ekf_correct_pixels_dev(self._h, M, self._p(ids), self._p(left), self._p(right), self._p(skip))
            self._check(rc, "correct_pixels_dev")
            return self._order_out(cur)
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        left = self._host(left, (B, M, 8))
        right = None if right is None else self._host(right, (B, M, 8))
        skip = None if skip is None else self._host(skip, (B,), np.uint8)
        rc = self._lib.fbus_ekf_correct_pixels(self._h, M, self._p(ids), self._p(left), self._p(right), self._p(skip))
        self._check(rc, "correct_pixels")

    def applied(self):
        out = np.empty(self.B, np.uint8)
        self._check(self._lib.fbus_ekf_get_applied(self._h, self._p(out)), "get_applied")
        return out

    def frame(self, accel, gyro, dt, ids, pos, quat, mode=capi.MODE_NEAREST, skip=None, fused=False):
        """K per-sample predict launches followed by one correct launch (device arrays only);
        fused=True: the same frame as ONE launch with the records resident in registers."""
        B = self.B
        K = accel.numel() // (3 * B) if accel is not None else 0
        M = ids.numel() // B if ids is not None else 0
        per = 0
        if K > 0:
            per = 1 if (dt.numel() == K * B and B > 1) else 0
            if not per and dt.numel() < K:
                raise ValueError("dt must have K or K*B elements")
            self._dev_checked(accel, K * B * 3, "accel"); self._dev_checked(gyro, K * B * 3, "gyro")
            self._dev_checked(dt, dt.numel(), "dt")
        if M > 0:
            self._dev_checked(ids, B * M, "ids"); self._dev_checked(pos, B * M * 3, "pos")
            self._dev_checked(quat, B * M * 4, "quat")
        if skip is not None:
            self._dev_checked(skip, B, "skip")
        fn = self._lib.fbus_ekf_frame_fused_dev if fused else self._lib.fbus_ekf_frame_dev
        cur = self._order_in(accel, gyro, dt, ids, pos, quat, skip)
        rc = fn(self._h, K, self._p(accel), self._p(gyro), self._p(dt), per, M,
                                          self._p(ids), self._p(pos), self._p(quat), mode, self._p(skip))
        self._check(rc, "frame_dev")
        self._order_out(cur)

    def frame_meas(self, accel, gyro, dt, ids, left, right=None, kind=capi.MEAS_PIXELS, geometry=capi.VIS_REFRACTIVE,
                   mode=capi.MODE_STACKED, skip=None):
        """One camera frame with the north star's MeasureUpdate in ONE launch (fbus_ekf_frame_meas_fused_dev; device arrays):
        K predicts, then correct_pixels (kind = MEAS_PIXELS; right=None: left camera) or correct_corners (MEAS_CORNERS, with its
        geometry / mode).  accel, gyro: (K, B, 3); dt: (K,) or (K, B); ids: (B, M); left / right: (B, M, 8) [(B, M, 12) corner
        positions for VIS_CORNERS3D]."""
        B = self.B
        K = accel.numel() // (3 * B) if accel is not None else 0
        M = ids.numel() // B if ids is not None else 0
        per = 0
        if K > 0:
            per = 1 if (dt.numel() == K * B and B > 1) else 0
            if not per and dt.numel() < K:
                raise ValueError("dt must have K or K*B elements")
            self._dev_checked(accel, K * B * 3, "accel"); self._dev_checked(gyro, K * B * 3, "gyro")
            self._dev_checked(dt, dt.numel(), "dt")
        if M > 0:
            lw = 12 if (kind == capi.MEAS_CORNERS and geometry == capi.VIS_CORNERS3D) else 8
            self._dev_checked(ids, B * M, "ids"); self._dev_checked(left, B * M * lw, "left")
            if right is not None:
                self._dev_checked(right, B * M * 8, "right")
        if skip is not None:
            self._dev_checked(skip, B, "skip")
        cur = self._order_in(accel, gyro, dt, ids, left, right, skip)
        rc = self._lib.fbus_ekf_frame_meas_fused_dev(self._h, K, self._p(accel), self._p(gyro), self._p(dt), per, kind, M,
                                                     self._p(ids), self._p(left), self._p(right), geometry, mode, self._p(skip))
        self._check(rc, "frame_meas_fused_dev")
        self._order_out(cur)

    def frames_meas(self, kcount, accel, gyro, dt, ids, left, right=None, kind=capi.MEAS_PIXELS, geometry=capi.VIS_REFRACTIVE,
                    mode=capi.MODE_STACKED, skip=None):
        """A window of camera frames with the north star's MeasureUpdate in ONE launch (fbus_ekf_frames_meas_fused_dev; device arrays):
        len(kcount) times { kcount[f] predicts, correct_pixels / correct_corners }.  accel, gyro: (sum kcount, B, 3); dt: (sum kcount,) or
        (sum kcount, B); ids: (F, B, M); left / right: (F, B, M, 8) [(F, B, M, 12) for VIS_CORNERS3D]; skip: (F, B) or None."""
        B = self.B
        kcount = np.ascontiguousarray(kcount, np.int32)
        F, Kt = int(kcount.size), int(kcount.sum())
        if F > capi.MAX_WINDOW_FRAMES:
            raise ValueError(f"at most {capi.MAX_WINDOW_FRAMES} frames per window")
        M = ids.numel() // (B * F) if (ids is not None and F > 0) else 0
        per = 0
        if Kt > 0:
            per = 1 if (dt.numel() == Kt * B and B > 1) else 0
            if not per and dt.numel() < Kt:
                raise ValueError("dt must have sum(kcount) or sum(kcount)*B elements")
            self._dev_checked(accel, Kt * B * 3, "accel"); self._dev_checked(gyro, Kt * B * 3, "gyro")
            self._dev_checked(dt, dt.numel(), "dt")
        if M > 0:
            lw = 12 if (kind == capi.MEAS_CORNERS and geometry == capi.VIS_CORNERS3D) else 8
            self._dev_checked(ids, F * B * M, "ids"); self._dev_checked(left, F * B * M * lw, "left")
            if right is not None:
                self._dev_checked(right, F * B * M * 8, "right")
        if skip is not None:
            self._dev_checked(skip, F * B, "skip")
        cur = self._order_in(accel, gyro, dt, ids, left, right, skip)
        rc = self._lib.fbus_ekf_frames_meas_fused_dev(self._h, F, kcount.ctypes.data_as(C.POINTER(C.c_int32)), self._p(accel), self._p(gyro),
                                                      self._p(dt), per, kind, M, self._p(ids), self._p(left), self._p(right), geometry, mode,
                                                      self._p(skip))
        self._check(rc, "frames_meas_fused_dev")
        self._order_out(cur)

    def frames(self, kcount, accel, gyro, dt, ids, pos, quat, mode=capi.MODE_NEAREST, skip=None):
        """A window of camera frames in ONE launch (device arrays): len(kcount) times { kcount[f] predicts, one correct }
        with the records resident in registers in between -- the frame loop of FBUS_EKF.m:151-210 over a recorded stretch.
        accel, gyro: (sum kcount, B, 3); dt: (sum kcount,) or (sum kcount, B); ids: (F, B, M); pos: (F, B, M, 3);
        quat: (F, B, M, 4); skip: (F, B) or None.  applied() afterwards reports the last frame."""
        B = self.B
        kcount = np.ascontiguousarray(kcount, np.int32)
        F, Kt = int(kcount.size), int(kcount.sum())
        if F > capi.MAX_WINDOW_FRAMES:
            raise ValueError(f"at most {capi.MAX_WINDOW_FRAMES} frames per window")
        M = ids.numel() // (B * F) if (ids is not None and F > 0) else 0
        per = 0
        if Kt > 0:
            per = 1 if (dt.numel() == Kt * B and B > 1) else 0
            if not per and dt.numel() < Kt:
                raise ValueError("dt must have sum(kcount) or sum(kcount)*B elements")
            self._dev_checked(accel, Kt * B * 3, "accel"); self._dev_checked(gyro, Kt * B * 3, "gyro")
            self._dev_checked(dt, dt.numel(), "dt")
        if M > 0:
            self._dev_checked(ids, F * B * M, "ids"); self._dev_checked(pos, F * B * M * 3, "pos")
            self._dev_checked(quat, F * B * M * 4, "quat")
        if skip is not None:
            self._dev_checked(skip, F * B, "skip")
        cur = self._order_in(accel, gyro, dt, ids, pos, quat, skip)
        rc = self._lib.fbus_ekf_frames_fused_dev(self._h, F, kcount.ctypes.data_as(C.POINTER(C.c_int32)), self._p(accel),
                                                 self._p(gyro), self._p(dt), per, M, self._p(ids), self._p(pos), self._p(quat),
                                                 mode, self._p(skip))
        self._check(rc, "frames_fused_dev")
        self._order_out(cur)

    # ---- init / reset / front door (host arrays) --------------------------------------------
    def init_gravity_bias(self, accel, gyro):
        """InitGravityAndGyrobias.m:36-40: accel, gyro (T, B, 3) -> g, bg of every filter."""
        accel = np.ascontiguousarray(accel, self.np_dtype)
        T = accel.size // (3 * self.B)
        gyro = self._host(gyro, (T, self.B, 3))
        self._check(self._lib.fbus_ekf_init_gravity_bias(self._h, T, self._p(accel), self._p(gyro)), "init_gravity_bias")

    def pose_init(self, ids, pos, quat, what=capi.POSE_INIT, mask=None):
        """InitPositionAndQuaternion.m / ResetState.m (what = POSE_INIT / POSE_RESET) from the nearest marker;
        returns the per-filter applied flags."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(self.B, -1)
        M = ids.shape[1]
        pos = self._host(pos, (self.B, M, 3)); quat = self._host(quat, (self.B, M, 4))
        mask = None if mask is None else self._host(mask, (self.B,), np.uint8)
        rc = self._lib.fbus_ekf_pose_init(self._h, M, self._p(ids), self._p(pos), self._p(quat), what, self._p(mask))
        self._check(rc, "pose_init")
        return self.applied()                      # 0 where nothing happened (no marker in range / in the map / masked)

    def vision_only_pose(self, ids, pos, quat):
        """ComputeVisionOnlyResults.m:39-79 -> (B, 7) [p3, q4]; the state is not touched."""
        ids = np.ascontiguousarray(ids, np.int32).reshape(self.B, -1)
        M = ids.shape[1]
        pos = self._host(pos, (self.B, M, 3)); quat = self._host(quat, (self.B, M, 4))
        out = np.zeros((self.B, 7), self.np_dtype)
        rc = self._lib.fbus_ekf_vision_only_pose(self._h, M, self._p(ids), self._p(pos), self._p(quat), self._p(out))
        self._check(rc, "vision_only_pose")
        return out

    def imu_ema(self, accel, gyro, restart=False):
        """IMU pre-filter of FILTER::SetImuData (filter.cpp:36-47); returns filtered copies (T, B, 3)."""
        accel = np.array(accel, self.np_dtype, order="C", copy=True)
        gyro = np.array(gyro, self.np_dtype, order="C", copy=True)
        T = accel.size // (3 * self.B)
        rc = self._lib.fbus_ekf_imu_ema(self._h, T, self._p(accel), self._p(gyro), 1 if restart else 0)
        self._check(rc, "imu_ema")
        return accel.reshape(T, self.B, 3), gyro.reshape(T, self.B, 3)

    # ---- marker pose from stereo corners (vision.cpp:472-759) ---------------------------
    def marker_pose(self, left, right=None, geometry=capi.VIS_REFRACTIVE, want_corners=False):
        """left/right: (n, 8) normalised corner coordinates (or left = (n, 12) 3-D corners with
        geometry VIS_CORNERS3D).  Host arrays in -> (pos (n,3), quat (n,4)[, corners (n,4,3)]) host arrays out;
        device arrays in -> device arrays of the same kind out (torch)."""
        w = 12 if geometry == capi.VIS_CORNERS3D else 8
        if _is_dev(left):
            import torch
            n = left.numel() // w
            pos = torch.empty((n, 3), dtype=left.dtype, device=left.device)
            quat = torch.empty((n, 4), dtype=left.dtype, device=left.device)
            c3 = torch.empty((n, 4, 3), dtype=left.dtype, device=left.device) if want_corners else None
            self._keep += [left, right, pos, quat, c3]
            cur = self._order_in(left, right)       # inputs come from / outputs go to torch's stream: order both ways
            rc = self._lib.fbus_ekf_marker_pose_dev(self._h, n, geometry, self._p(left), self._p(right),
                                                    self._p(pos), self._p(quat), self._p(c3))
            self._check(rc, "marker_pose_dev")
            self._order_out(cur)
            return (pos, quat, c3) if want_corners else (pos, quat)
        left = np.ascontiguousarray(left, self.np_dtype).reshape(-1, w)
        n = left.shape[0]
        right = None if right is None else self._host(right, (n, 8))
        pos = np.empty((n, 3), self.np_dtype)
        quat = np.empty((n, 4), self.np_dtype)
        c3 = np.empty((n, 4, 3), self.np_dtype) if want_corners else None
        rc = self._lib.fbus_ekf_marker_pose(self._h, n, geometry, self._p(left), self._p(right), self._p(pos),
                                            self._p(quat), self._p(c3))
        self._check(rc, "marker_pose")
        return (pos, quat, c3) if want_corners else (pos, quat)

    # ---- L0 helpers one by one (unit-test hook) -------------------------------------------
    def l0_eval(self, op, a, b=None):
        """one of the device inline helpers (capi.L0_*) on n rows of host input; see include/fbus_ekf.h"""
        wa, wb, wo = (4, 4, 4, 4, 3, 3, 1)[op], (4, 0, 0, 0, 1, 0, 0)[op], (4, 9, 9, 4, 9, 4, 4)[op]
        a = np.ascontiguousarray(a, self.np_dtype).reshape(-1, wa)
        n = a.shape[0]
        b = None if not wb else self._host(b, (n, wb))
        out = np.empty((n, wo), self.np_dtype)
        self._check(self._lib.fbus_ekf_l0_eval(self._h, op, n, self._p(a), self._p(b), self._p(out)), "l0_eval")
        return out

    # ---- HIP graphs ------------------------------------------------------------------------
    def graph_capture(self, fn):
        """Runs fn() (device-array calls on this filter only) under stream capture; returns a graph id."""
        self._check(self._lib.fbus_ekf_graph_begin(self._h), "graph_begin")
        self._capturing = True          # no cross-stream markers inside a capture: the caller orders the graph launch
        try:
            fn()
        finally:
            self._capturing = False
            gid = C.c_int(-1)
            rc = self._lib.fbus_ekf_graph_end(self._h, C.byref(gid))
        self._check(rc, "graph_end")
        self._graph_keep = getattr(self, "_graph_keep", []) + list(self._keep)   # captured pointers must stay alive
        return gid.value

    def graph_launch(self, gid):
        self._check(self._lib.fbus_ekf_graph_launch(self._h, gid), "graph_launch")

    # ---- timing -------------------------------------------------------------------------
    def timing_enable(self, on=True, stride=1):
        """stride: frame() brackets only every stride-th frame with HIP events"""
        self._check(self._lib.fbus_ekf_timing_enable(self._h, int(stride) if on else 0), "timing_enable")

    def timing_reset(self):
        self._check(self._lib.fbus_ekf_timing_reset(self._h), "timing_reset")

    def timing_read(self, kernel):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.fbus_ekf_timing_read(self._h, kernel, C.byref(ms), C.byref(n)), "timing_read")
        return ms.value, n.value
