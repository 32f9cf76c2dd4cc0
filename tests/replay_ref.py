"""Test helper: drives fbus_ekf.replay with the fp64 oracle as the engine."""
import numpy as np

import oracle_capi as oc
from fbus_ekf import capi, replay


class OracleEngine:
    """BatchedFilter-shaped adapter around oracle_capi.Oracle (tests only)."""

    def __init__(self, batch, dialect=0, nstate=18, cov_form=0):
        self.B, self.N = batch, nstate
        self.orc = oc.Oracle(dialect, nstate, cov_form)
        self.nominal = np.zeros((batch, 19))
        self.rot = np.zeros((batch, 9))
        self.P = np.zeros((batch, nstate, nstate))
        self.prev = np.zeros(batch, np.int32)

    def set_state(self, nominal=None, rot=None, P=None, prev_id=None):
        if nominal is not None: self.nominal[...] = np.asarray(nominal, float).reshape(self.B, 19)
        if rot is not None: self.rot[...] = np.asarray(rot, float).reshape(self.B, 9)
        if P is not None: self.P[...] = np.asarray(P, float).reshape(self.B, self.N, self.N)
        if prev_id is not None: self.prev[...] = prev_id

    def get_state(self):
        return self.nominal.copy(), self.rot.copy(), self.P.copy(), self.prev.copy()

    def predict(self, accel, gyro, dt):
        self.orc.predict(self.nominal, self.rot, self.P, self.prev,
                         np.asarray(accel, float).reshape(self.B, 3), np.asarray(gyro, float).reshape(self.B, 3),
                         np.asarray(dt, float))

    def init_gravity_bias(self, accel, gyro):
        accel = np.asarray(accel, float).reshape(-1, self.B, 3)
        gyro = np.asarray(gyro, float).reshape(-1, self.B, 3)
        for b in range(self.B):
            g, bg = oc.init_gravity_bias(accel[:, b], gyro[:, b])
            self.nominal[b, 16:19], self.nominal[b, 13:16] = g, bg

    def pose_init(self, ids, pos, quat, what=0, mask=None):
        return self.orc.pose_init(self.nominal, self.rot, ids, pos, quat, what, 2.0, mask)[0]

    def correct(self, ids, pos, quat, mode=0, skip=None):
        return self.orc.correct(self.nominal, self.rot, self.P, self.prev, ids, pos, quat, mode)

    # the north star's update on the oracle side: marker_size / r_pix are the handle's parameters (set them from fbus_params)
    marker_size, r_pix = 0.28, 1e-6

    def correct_pixels(self, ids, left, right=None, skip=None):
        ids = np.ascontiguousarray(ids, np.int32).reshape(self.B, -1)
        M = ids.shape[1]
        left = np.ascontiguousarray(left, float).reshape(self.B, M, 8)
        right = None if right is None else np.ascontiguousarray(right, float).reshape(self.B, M, 8)
        return self.orc.correct_pixels(self.nominal, self.rot, self.P, self.prev, ids, left, right, self.marker_size, self.r_pix)


def replay_with_oracle(imu, image, dialect, nframes):
    eng = OracleEngine(1, dialect, 18)
    prm = capi.default_params(dialect)
    return replay.replay(eng, imu, image, prm, max_frames=nframes)
