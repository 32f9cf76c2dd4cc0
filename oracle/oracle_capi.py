"""oracle_capi.py -- TEST INFRASTRUCTURE: ctypes loader for oracle/_build/libfbus_oracle*.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  The product package (fbus-ekf_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FBO_MAX_MARKERS = 32
MATLAB, CPP = 0, 1
NEAREST, STACKED = 0, 1
SIMPLE, JOSEPH = 0, 1


class FboParams(C.Structure):
    _fields_ = [
        ("dialect", C.c_int), ("nstate", C.c_int),
        ("q_diag", C.c_double * 4), ("r_pos", C.c_double), ("r_quat", C.c_double),
        ("R_IL", C.c_double * 9), ("P_IL", C.c_double * 3), ("Q_IL", C.c_double * 4),
        ("n_markers", C.c_int), ("marker_id", C.c_int * FBO_MAX_MARKERS),
        ("marker_pos", (C.c_double * 3) * FBO_MAX_MARKERS),
        ("marker_quat", (C.c_double * 4) * FBO_MAX_MARKERS),
        ("switch_thres", C.c_double), ("cov_form", C.c_int),
    ]


class FboState(C.Structure):
    _fields_ = [
        ("p", C.c_double * 3), ("v", C.c_double * 3), ("q", C.c_double * 4), ("ba", C.c_double * 3),
        ("bg", C.c_double * 3), ("g", C.c_double * 3), ("R", C.c_double * 9), ("P", C.c_double * (18 * 18)),
        ("prev_id", C.c_int),
    ]


class FbvParams(C.Structure):
    _fields_ = [
        ("R_IL", C.c_double * 9), ("P_LI", C.c_double * 3),
        ("R_IR", C.c_double * 9), ("P_RI", C.c_double * 3),
        ("n_air", C.c_double), ("n_glass", C.c_double), ("n_water", C.c_double),
        ("d_air", C.c_double), ("d_glass", C.c_double), ("normal", C.c_double * 3),
    ]


def build(native=False):
    """make the oracle library; returns its path."""
    target = ["native"] if native else []
    subprocess.run(["make", "-C", _HERE] + target, check=True, stdout=subprocess.DEVNULL)
    name = "libfbus_oracle_native.so" if native else "libfbus_oracle.so"
    return os.path.join(_HERE, "_build", name)


_libs = {}


def load(native=False):
    key = bool(native)
    if key in _libs:
        return _libs[key]
    name = "libfbus_oracle_native.so" if native else "libfbus_oracle.so"
    path = os.path.join(_HERE, "_build", name)
    if os.environ.get("FBUS_ORACLE_LIB"):          # e.g. the -fsanitize=address,undefined build (tests/test_sanitizers_cpu.py)
        path = os.environ["FBUS_ORACLE_LIB"]
    elif not os.path.exists(path):
        path = build(native)
    lib = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    lib.fbo_default_params.argtypes = [C.POINTER(FboParams), C.c_int, C.c_int]
    lib.fbo_default_P0.argtypes = [C.POINTER(FboParams), dp]
    lib.fbo_predict_batch.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(FboParams), dp, dp, dp, C.c_int, C.c_int]
    lib.fbo_correct_batch.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(FboParams), C.c_int, ip, dp, dp,
                                      C.c_int, ip, C.c_int]
    lib.fbo_frame_batch.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(FboParams), C.c_int, dp, dp, dp, C.c_int,
                                    ip, dp, dp, C.c_int, C.c_int]
    lib.fbo_schedule_batch.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(FboParams), C.c_int, ip, C.c_int, dp, dp, dp,
                                       C.c_int, ip, dp, dp, C.c_int, C.c_int]
    lib.fbo_correct_corners_batch.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(FboParams), C.c_int, ip, dp,
                                              C.c_double, C.c_int, ip]
    lib.fbo_transition.argtypes = [C.POINTER(FboState), C.POINTER(FboParams), dp, dp, C.c_double, dp]
    lib.fbo_measurement.argtypes = [C.POINTER(FboState), C.POINTER(FboParams), C.c_int, dp, dp, dp, dp, dp]
    lib.fbo_predict.argtypes = [C.POINTER(FboState), C.POINTER(FboParams), dp, dp, C.c_double]
    lib.fbo_correct_pixels_batch.argtypes = [C.c_int, dp, dp, dp, ip, C.POINTER(FboParams), C.POINTER(FbvParams), C.c_int, ip,
                                             dp, dp, C.c_double, C.c_double, ip]
    lib.fbo_correct_pixels_analytic_batch.argtypes = lib.fbo_correct_pixels_batch.argtypes
    lib.fbv_project_camera_jac.argtypes = [C.POINTER(FbvParams), dp, C.c_int, dp, dp]
    lib.fbv_project_camera_jac.restype = C.c_int
    lib.fbv_project_camera.argtypes = [C.POINTER(FbvParams), dp, C.c_int, dp]
    lib.fbv_project_camera.restype = C.c_int
    lib.fbv_project_stereo.argtypes = [C.POINTER(FbvParams), dp, dp, dp]
    lib.fbv_project_stereo.restype = C.c_int
    lib.fbv_refraction_project.argtypes = [C.POINTER(FbvParams), dp, dp]
    lib.fbv_refraction_project.restype = C.c_int
    u8 = C.POINTER(C.c_ubyte)
    lib.fbo_init_gravity_bias.argtypes = [C.c_int, dp, dp, dp, dp]
    lib.fbo_pose_init_batch.argtypes = [C.c_int, dp, dp, C.POINTER(FboParams), C.c_int, ip, dp, dp, C.c_int, C.c_double,
                                        u8, dp, ip]
    lib.fbo_imu_ema.argtypes = [C.c_int, dp, dp, C.c_int]
    lib.fbv_default_params.argtypes = [C.POINTER(FbvParams)]
    lib.fbv_refraction_triangulate.argtypes = [C.POINTER(FbvParams), dp, dp, dp]
    lib.fbv_normal_triangulate.argtypes = [C.POINTER(FbvParams), dp, dp, dp]
    lib.fbv_marker_pose.argtypes = [dp, dp, dp, dp]
    _libs[key] = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Oracle:
    """Batched fp64 oracle over flat numpy arrays (nominal Bx19, rot Bx9, P Bxnxn, prev B)."""

    def __init__(self, dialect=MATLAB, nstate=18, cov_form=SIMPLE, native=False, nthreads=1):
        self.lib = load(native)
        self.prm = FboParams()
        self.lib.fbo_default_params(C.byref(self.prm), dialect, nstate)
        self.prm.cov_form = cov_form
        self.n = nstate
        self.nthreads = nthreads

    def P0(self):
        P = np.zeros((self.n, self.n))
        self.lib.fbo_default_P0(C.byref(self.prm), _dp(P))
        return P

    # ---- single-state views of the linearisation (finite-difference tests) ----
    def _state(self, nominal, rot):
        st = FboState()
        x = np.asarray(nominal, np.float64).reshape(19)
        st.p[:] = x[0:3]; st.v[:] = x[3:6]; st.q[:] = x[6:10]; st.ba[:] = x[10:13]; st.bg[:] = x[13:16]; st.g[:] = x[16:19]
        st.R[:] = np.asarray(rot, np.float64).reshape(9)
        return st

    def transition(self, nominal, rot, accel, gyro, dt):
        """Fx (n x n) the oracle's predict uses at this state (ImuUpdate.m:63-69 / filter.cpp:597-604)"""
        st = self._state(nominal, rot)
        Fx = np.zeros((self.n, self.n))
        self.lib.fbo_transition(C.byref(st), C.byref(self.prm), _dp(np.ascontiguousarray(accel, np.float64)),
                                _dp(np.ascontiguousarray(gyro, np.float64)), float(dt), _dp(Fx))
        return Fx

    def predict_nominal(self, nominal, rot, accel, gyro, dt):
        """the nominal-state kinematics of ONE filter through fbo_predict -> (nominal 19, rot 9)"""
        st = self._state(nominal, rot)
        self.lib.fbo_predict(C.byref(st), C.byref(self.prm), _dp(np.ascontiguousarray(accel, np.float64)),
                             _dp(np.ascontiguousarray(gyro, np.float64)), float(dt))
        out = np.concatenate([np.array(st.p), np.array(st.v), np.array(st.q), np.array(st.ba), np.array(st.bg), np.array(st.g)])
        return out, np.array(st.R)

    def measurement(self, nominal, rot, marker_id, yp, yq):
        """(h 7, H 7 x n, r 7) the oracle's correct uses for this marker (MeasureUpdate.m:67-88 / filter.cpp:684-721)"""
        st = self._state(nominal, rot)
        h, H, r = np.zeros(7), np.zeros((7, self.n)), np.zeros(7)
        ok = self.lib.fbo_measurement(C.byref(st), C.byref(self.prm), int(marker_id),
                                      _dp(np.ascontiguousarray(yp, np.float64)), _dp(np.ascontiguousarray(yq, np.float64)),
                                      _dp(h), _dp(H), _dp(r))
        if not ok:
            raise KeyError(marker_id)
        return h, H, r

    def predict(self, nominal, rot, P, prev, accel, gyro, dt):
        B = nominal.shape[0]
        accel = np.ascontiguousarray(accel, np.float64)
        gyro = np.ascontiguousarray(gyro, np.float64)
        dt = np.ascontiguousarray(np.atleast_1d(dt), np.float64)
        stride = 1 if dt.size == B and B > 1 else 0
        for a in (nominal, rot, P):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        assert prev.dtype == np.int32
        self.lib.fbo_predict_batch(B, _dp(nominal), _dp(rot), _dp(P), _ip(prev), C.byref(self.prm),
                                   _dp(accel), _dp(gyro), _dp(dt), stride, self.nthreads)

    def correct(self, nominal, rot, P, prev, ids, pos, quat, mode=NEAREST):
        B = nominal.shape[0]
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        pos = np.ascontiguousarray(pos, np.float64).reshape(B, M, 3)
        quat = np.ascontiguousarray(quat, np.float64).reshape(B, M, 4)
        applied = np.zeros(B, np.int32)
        self.lib.fbo_correct_batch(B, _dp(nominal), _dp(rot), _dp(P), _ip(prev), C.byref(self.prm), M,
                                   _ip(ids), _dp(pos), _dp(quat), mode, _ip(applied), self.nthreads)
        return applied


    def frame(self, nominal, rot, P, prev, accel, gyro, dt, ids, pos, quat, mode=NEAREST):
        """K predicts + one correct per filter, threads spawned once (CPU-baseline driver)."""
        B = nominal.shape[0]
        accel = np.ascontiguousarray(accel, np.float64)
        K = accel.size // (3 * B)
        gyro = np.ascontiguousarray(gyro, np.float64)
        dt = np.ascontiguousarray(dt, np.float64)
        assert dt.size == K
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        pos = np.ascontiguousarray(pos, np.float64)
        quat = np.ascontiguousarray(quat, np.float64)
        self.lib.fbo_frame_batch(B, _dp(nominal), _dp(rot), _dp(P), _ip(prev), C.byref(self.prm), K, _dp(accel),
                                 _dp(gyro), _dp(dt), M, _ip(ids), _dp(pos), _dp(quat), mode, self.nthreads)


    def correct_corners(self, nominal, rot, P, prev, ids, corners, size, mode=NEAREST):
        """corner-row model (no reference counterpart): corners (B, M, 4, 3) triangulated positions"""
        B = nominal.shape[0]
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        corners = np.ascontiguousarray(corners, np.float64).reshape(B, M, 12)
        applied = np.zeros(B, np.int32)
        self.lib.fbo_correct_corners_batch(B, _dp(nominal), _dp(rot), _dp(P), _ip(prev), C.byref(self.prm), M, _ip(ids),
                                           _dp(corners), float(size), mode, _ip(applied))
        return applied

    def correct_pixels(self, nominal, rot, P, prev, ids, left, right, size, r_pix, vision=None, analytic=False):
        """pixel-row model (no reference counterpart): left / right (B, M, 8) normalised corner image points, right may be
        None (left camera only: 2 rows per corner).  analytic: d pi / d X in closed form (fbo_correct_pixels_analytic) instead of
        central differences"""
        B = nominal.shape[0]
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        left = np.ascontiguousarray(left, np.float64).reshape(B, M, 8)
        right = None if right is None else np.ascontiguousarray(right, np.float64).reshape(B, M, 8)
        vp = vision if vision is not None else vision_params()
        applied = np.zeros(B, np.int32)
        fn = self.lib.fbo_correct_pixels_analytic_batch if analytic else self.lib.fbo_correct_pixels_batch
        fn(B, _dp(nominal), _dp(rot), _dp(P), _ip(prev), C.byref(self.prm), C.byref(vp), M,
           _ip(ids), _dp(left), None if right is None else _dp(right), float(size), float(r_pix), _ip(applied))
        return applied

    def schedule(self, nominal, rot, P, prev, Ks, reps, accel, gyro, dt, ids, pos, quat, mode=NEAREST):
        """reps x (frames of Ks[f] predicts + one correct) per filter inside ONE thread team (CPU baseline)."""
        B = nominal.shape[0]
        Ks = np.ascontiguousarray(Ks, np.int32)
        accel = np.ascontiguousarray(accel, np.float64); gyro = np.ascontiguousarray(gyro, np.float64)
        dt = np.ascontiguousarray(dt, np.float64)
        ids = np.ascontiguousarray(ids, np.int32)
        M = ids.size // (len(Ks) * B)
        pos = np.ascontiguousarray(pos, np.float64); quat = np.ascontiguousarray(quat, np.float64)
        assert accel.size == int(Ks.sum()) * B * 3 and dt.size == int(Ks.sum())
        self.lib.fbo_schedule_batch(B, _dp(nominal), _dp(rot), _dp(P), _ip(prev), C.byref(self.prm), len(Ks), _ip(Ks),
                                    reps, _dp(accel), _dp(gyro), _dp(dt), M, _ip(ids), _dp(pos), _dp(quat), mode,
                                    self.nthreads)

    def pose_init(self, nominal, rot, ids, pos, quat, what, max_dist=2.0, mask=None):
        """what 0 init / 1 reset / 2 vision-only (returns out7); modifies nominal/rot in place for 0/1."""
        B = nominal.shape[0]
        ids = np.ascontiguousarray(ids, np.int32).reshape(B, -1)
        M = ids.shape[1]
        pos = np.ascontiguousarray(pos, np.float64); quat = np.ascontiguousarray(quat, np.float64)
        out7 = np.zeros((B, 7))
        applied = np.zeros(B, np.int32)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.lib.fbo_pose_init_batch(B, _dp(nominal), _dp(rot), C.byref(self.prm), M, _ip(ids), _dp(pos), _dp(quat),
                                     what, max_dist, None if m is None else m.ctypes.data_as(C.POINTER(C.c_ubyte)),
                                     _dp(out7), _ip(applied))
        return applied, out7


# ---- L0 helpers of the oracle (matlab/quaternion_*.m, axisangle_to_quaternion.m ; matrix_math.hpp:26-99) ----
def l0(name, *args, out):
    """calls fbo_<name>(args..., out) with double arrays; out = number of doubles returned"""
    lib = load()
    fn = getattr(lib, "fbo_" + name)
    fn.restype = None
    res = np.zeros(out)
    cargs = []
    for x in args:
        if np.isscalar(x):
            cargs.append(C.c_double(float(x)))
        else:
            cargs.append(_dp(np.ascontiguousarray(x, np.float64)))
    fn(*cargs, _dp(res))
    return res


def init_gravity_bias(accel, gyro):
    """accel, gyro (T, 3) of ONE filter -> g, bg"""
    lib = load()
    a = np.ascontiguousarray(accel, np.float64); w = np.ascontiguousarray(gyro, np.float64)
    g, bg = np.zeros(3), np.zeros(3)
    lib.fbo_init_gravity_bias(a.shape[0], _dp(a), _dp(w), _dp(g), _dp(bg))
    return g, bg


def imu_ema(x6, carry=None):
    """x6 (T, 6) of ONE filter, filtered copy; carry (6,) previous filtered sample or None"""
    lib = load()
    x = np.array(x6, np.float64, order="C", copy=True)
    c = np.zeros(6) if carry is None else np.array(carry, np.float64, copy=True)
    lib.fbo_imu_ema(x.shape[0], _dp(x), _dp(c), 0 if carry is None else 1)
    return x, c


def vision_params():
    lib = load()
    p = FbvParams()
    lib.fbv_default_params(C.byref(p))
    return p


def refraction_triangulate(p, left8, right8):
    lib = load()
    out = np.zeros(12)
    lib.fbv_refraction_triangulate(C.byref(p), _dp(np.ascontiguousarray(left8, np.float64)),
                                   _dp(np.ascontiguousarray(right8, np.float64)), _dp(out))
    return out.reshape(4, 3)


def project_stereo(p, Xcam, stereo=True):
    """forward flat-port projection of points (n, 3) of the left camera frame -> (uvL (n,2), uvR (n,2) or None, ok (n,))"""
    lib = load()
    X = np.ascontiguousarray(Xcam, np.float64).reshape(-1, 3)
    uvL, uvR, ok = np.zeros((len(X), 2)), np.zeros((len(X), 2)), np.zeros(len(X), bool)
    for i in range(len(X)):
        ok[i] = bool(lib.fbv_project_stereo(C.byref(p), _dp(X[i]), _dp(uvL[i]), _dp(uvR[i]) if stereo else None))
    return uvL, (uvR if stereo else None), ok


def marker_pose(corners12):
    lib = load()
    c = np.ascontiguousarray(corners12, np.float64).reshape(12)
    pos, quat, rot = np.zeros(3), np.zeros(4), np.zeros(9)
    lib.fbv_marker_pose(_dp(c), _dp(pos), _dp(quat), _dp(rot))
    return pos, quat, rot.reshape(3, 3)
