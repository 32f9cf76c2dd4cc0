"""ekf_oracle_np.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Independent numpy/scipy twin of oracle/ekf_oracle.c, written in matrix form
straight from the reference's Matlab (`expm`, `inv`, whole-matrix products) so
that two separately written fp64 restatements can be cross-checked
(tests/test_oracle_cpu.py).  PARITY STATUS: "parity unpinned" for the EKF
arithmetic -- the reference has no known-answer test for it and cannot be run
here (no Matlab/Octave, no Eigen).  It also generates the golden vectors in
tests/golden/ (tests/golden/make_golden.py).

Reference lines followed:
  predict : matlab/ImuUpdate.m:36-82 ; C++/src/filter.cpp:533-616
  correct : matlab/MeasureUpdate.m:37-103 ; C++/src/filter.cpp:622-741
"""
import numpy as np
from scipy.linalg import expm

MATLAB, CPP = 0, 1
NEAREST, STACKED = 0, 1

# matlab/config/camerainfo.yml:11-15 (== C++/config/camerainfo1.yml), raw left TSC
TSC_LEFT_RAW = np.array([[-0.999862, 0.015685, -0.00548, 0.059967],
                         [-0.015639, -0.999843, -0.00827, 0.000127837],
                         [-0.005609, -0.008183, 0.999951, -0.002],
                         [0, 0, 0, 1.0]])

_I3 = np.eye(3)
_RA = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0.0]])
_RB = np.array([[1, 0, 0], [0, -1, 0], [0, 0, -1.0]])
# matlab/GetMarkerMap.m:1-63 == C++/config/markersetup.yml
MARKER_MAP = [
    (0, [0, 0, 0], _I3), (1, [0, 0.61, 0.285], _RA), (2, [0, 0.61, 1.185], _RA),
    (3, [0, 0.61, 2.085], _RA), (4, [0, 0.61, 2.985], _RA), (5, [0, 0.265, 4.12], _RB),
    (6, [0, -0.635, 4.12], _RB), (7, [0, -1.535, 4.12], _RB), (8, [0, -2.435, 4.12], _RB),
    (16, [0, -2.7, 0], _I3), (17, [0, -1.8, 0], _I3), (18, [0, -0.9, 0], _I3),
]


def qmul(p, q):  # quaternion_add.m:22-28
    pw, px, py, pz = p
    qw, qx, qy, qz = q
    return np.array([pw * qw - px * qx - py * qy - pz * qz,
                     pw * qx + px * qw + py * qz - pz * qy,
                     pw * qy - px * qz + py * qw + pz * qx,
                     pw * qz + px * qy - py * qx + pz * qw])


def aa2q(axis, angle):  # axisangle_to_quaternion.m:22-29
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(angle / 2)], axis * np.sin(angle / 2)])


def q2R(q):  # quaternion_to_rotmat.m:22-33
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def q2R_eigen(q):  # Eigen toRotationMatrix
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def R2q(R):  # trace based (Eigen Quaterniond(Matrix3d))
    R = np.asarray(R, float)
    t = np.trace(R)
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1:] = [(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t]
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1)
        q[1 + i] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def skew(v):  # vector_to_crossmat.m:22-30
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def Lq(q):  # quaternion_left_product_matrix.m
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]])


def Rq(q):  # quaternion_right_product_matrix.m
    w, x, y, z = q
    return np.array([[w, -x, -y, -z], [x, w, z, -y], [y, -z, w, x], [z, y, -x, w]])


class Params:
    def __init__(self, dialect=MATLAB, nstate=18):
        self.dialect, self.n = dialect, nstate
        self.q_diag = np.array([1e-3, 1e-4, 1e-3, 1e-4])
        self.r_pos, self.r_quat = (0.01, 0.01) if dialect == MATLAB else (0.001, 0.001)
        self.switch_thres = 0.5
        self.joseph = False
        T = np.diag([-1.0, -1, 1, 1]) @ TSC_LEFT_RAW  # FBUS_EKF.m:68
        self.R_IL = T[:3, :3]
        self.P_IL = -self.R_IL.T @ T[:3, 3]
        self.Q_IL = R2q(self.R_IL)
        self.markers = {i: (np.array(p, float), R2q(r)) for i, p, r in MARKER_MAP}

    def P0(self):
        d = ([1e-4, 0.1, 1e-4, 1e-3, 1e-3, 100.0] if self.dialect == MATLAB
             else [1e-4, 1e-2, 1e-4, 1e-2, 1e-2, 100.0])
        return np.diag(np.repeat(d, 3)[:self.n])


class State:
    def __init__(self, n=18):
        self.p = np.zeros(3); self.v = np.zeros(3); self.q = np.array([1.0, 0, 0, 0])
        self.ba = np.zeros(3); self.bg = np.zeros(3); self.g = np.zeros(3)
        self.R = np.eye(3); self.P = np.eye(n); self.prev_id = 0

    def copy(self):
        s = State(self.P.shape[0])
        for k in ("p", "v", "q", "ba", "bg", "g", "R", "P"):
            setattr(s, k, getattr(self, k).copy())
        s.prev_id = self.prev_id
        return s


def predict(s, prm, accel, gyro, dt):
    n = prm.n
    a = accel - s.ba
    w = gyro - s.bg
    Fx = np.eye(n)
    Fx[0:3, 3:6] = np.eye(3) * dt
    Fx[3:6, 6:9] = -s.R @ skew(a) * dt
    Fx[3:6, 9:12] = -s.R * dt
    if n == 18:
        Fx[3:6, 15:18] = np.eye(3) * dt
    Fx[6:9, 6:9] = expm(-skew(w) * dt) if prm.dialect == MATLAB else np.eye(3) - skew(w) * dt
    Fx[6:9, 12:15] = -np.eye(3) * dt
    Fi = np.zeros((n, 12))
    Fi[3:15, :] = np.eye(12)
    Q = np.diag(np.repeat(prm.q_diag, 3))
    P = Fx @ s.P @ Fx.T + Fi @ Q @ Fi.T
    if prm.dialect == MATLAB:
        dth = np.linalg.norm(w * dt)
        qT = qmul(s.q, aa2q(w, dth))
        qH = qmul(s.q, aa2q(w, dth / 2))
        R0, RH, RT = s.R, q2R(qH), q2R(qT)
        qnew = qT / np.linalg.norm(qT)
    else:
        wn = np.linalg.norm(w)
        R0 = q2R_eigen(s.q)
        if wn > 10e-5:
            qH = qmul(s.q, aa2q(w / wn, wn * dt / 2))
            qT = qmul(s.q, aa2q(w / wn, wn * dt))
        else:
            qH = qmul(s.q, np.concatenate([[1.0], 0.5 * dt * w / 2]))
            qT = qmul(s.q, np.concatenate([[1.0], 0.5 * dt * w]))
        qH = qH / np.linalg.norm(qH)
        qT = qT / np.linalg.norm(qT)
        RH, RT = q2R_eigen(qH), q2R_eigen(qT)
        qnew = qT
    kv1 = R0 @ a + s.g
    kv2 = RH @ a + s.g
    kv3 = kv2
    kv4 = RT @ a + s.g
    v = s.v + dt / 6 * (kv1 + 2 * kv2 + 2 * kv3 + kv4)
    kp1 = s.v
    kp2 = s.v + kv1 * dt / 2
    kp3 = s.v + kv2 * dt / 2
    kp4 = s.v + kv3 * dt / 2
    p = s.p + dt / 6 * (kp1 + 2 * kp2 + 2 * kp3 + kp4)
    s.q, s.R, s.v, s.p = qnew, RT, v, p
    s.P = (P + P.T) / 2
    return s


def _rows(s, prm, mid, yp, yq):
    n = prm.n
    Pm, Qm = prm.markers[mid]
    L1 = np.zeros((4, 3)); L1[1:, :] = 0.5 * np.eye(3)
    L2 = np.diag([1.0, -1, -1, -1])
    RR = s.R @ prm.R_IL.T
    hp = RR.T @ (Pm - s.p - s.R @ prm.P_IL)
    hq = qmul(qmul(prm.Q_IL, s.q * np.array([1, -1, -1, -1.0])), Qm)
    H = np.zeros((7, n))
    H[0:3, 0:3] = -RR.T
    H[0:3, 6:9] = prm.R_IL @ skew(s.R.T @ (Pm - s.p))
    H[3:7, 6:9] = Rq(Qm) @ Lq(prm.Q_IL) @ L2 @ Lq(s.q) @ L1
    if np.linalg.norm(yq - hq) > np.linalg.norm(yq + hq):
        hq = -hq
        H[3:7, 6:9] = -H[3:7, 6:9]
    r = np.concatenate([yp - hp, (yq - hq) if prm.dialect == CPP else np.zeros(4)])
    return H, r


def correct(s, prm, ids, pos, quat, mode=NEAREST):
    n = prm.n
    ids = list(ids)
    pos = np.asarray(pos, float).reshape(-1, 3)
    quat = np.asarray(quat, float).reshape(-1, 4)
    if mode == NEAREST:
        mi, md, pi_, pd = -1, 10.0, -1, 0.0
        for i, mid in enumerate(ids):
            if mid < 0:
                continue
            d = np.linalg.norm(pos[i])
            if d < md:
                md, mi = d, i
            if prm.dialect == CPP and mid == s.prev_id:
                pd, pi_ = d, i
        if mi < 0:
            return False
        if prm.dialect == CPP and abs(pd - md) < prm.switch_thres and pd != 0:
            mi = pi_
        if ids[mi] not in prm.markers:
            return False
        if prm.dialect == CPP:
            s.prev_id = ids[mi]
        sel = [mi]
    else:
        sel = [i for i, mid in enumerate(ids) if mid >= 0 and mid in prm.markers]
        if not sel:
            return False
    Hs, rs = zip(*[_rows(s, prm, ids[i], pos[i], quat[i]) for i in sel])
    H = np.vstack(Hs)
    r = np.concatenate(rs)
    Rm = np.diag(np.tile([prm.r_pos] * 3 + [prm.r_quat] * 4, len(sel)))
    S = H @ s.P @ H.T + Rm
    if prm.dialect == MATLAB:
        K = s.P @ H.T @ np.linalg.inv(S)
    else:
        K = np.linalg.solve(S, H @ s.P).T
    dx = K @ r
    s.p = s.p + dx[0:3]
    s.v = s.v + dx[3:6]
    q = qmul(s.q, aa2q(dx[6:9], np.linalg.norm(dx[6:9])))
    s.q = q / np.linalg.norm(q)
    s.ba = s.ba + dx[9:12]
    s.bg = s.bg + dx[12:15]
    if n == 18:
        s.g = s.g + dx[15:18]
    IKH = np.eye(n) - K @ H
    P = IKH @ s.P
    if prm.joseph:
        P = P @ IKH.T + K @ Rm @ K.T
    s.P = (P + P.T) / 2
    return True
