"""Config 5 (84 stacked pose rows): how much of the fp32-vs-fp64 gap of one frame is the fp32 REPRESENTATION of the predicted
state, with exact arithmetic?  The oracle (fp64) corrects the exact predicted state and the same state rounded to fp32 (nominal
state / position only / covariance only / everything); the parity figures of tests/util.py between the two posteriors.
CPU only; needs the built library only for the default parameters.  Result (4000 filters): all rounded: sigma-aware 1.5e-6,
plain 3.7e-5, both in block v -- the update gain from position to velocity (~17 / s) times the 6e-8 m quantum of p and R."""
import sys
sys.path.insert(0, '/root/repo/fbus-ekf_amd'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_capi as oc
from fbus_ekf import synth, capi
from util import parity_errors
f32 = np.float32
r32 = lambda a: np.asarray(a, np.float64).astype(f32).astype(np.float64)
orc = oc.Oracle(0, 18)
prm = capi.default_params(0)
B, M = 4000, 12
p0 = [1e-4, 0.1, 1e-4, 1e-3, 1e-3, 100]
nom, rot, P, prev = synth.initial_state(0, B, p0, 18)
nom, rot, P = (np.ascontiguousarray(r32(x)) for x in (nom, rot, P))
prev = np.zeros(B, np.int32)
nom0 = nom.copy()
acc, gyr = synth.imu_samples(0, B, 0, 7, nom)
for k in range(7):
    orc.predict(nom, rot, P, prev, r32(acc[k]), r32(gyr[k]), np.array([0.005]))
ids, pos, quat = synth.marker_frame(0, B, 0, M, nom0, prm)
pos, quat = r32(pos), r32(quat)
# path A: exact fp64 predicted state -> correct ; path B: the predicted state rounded to fp32 (nominal, rot, P), then correct
A = [x.copy() for x in (nom, rot, P)]
Bq = [np.ascontiguousarray(r32(x)) for x in (nom, rot, P)]
for name, S in (("nominal only rounded", [np.ascontiguousarray(r32(nom)), rot.copy(), P.copy()]),
                ("p only rounded", [np.ascontiguousarray(np.concatenate([r32(nom[:, :3]), nom[:, 3:]], axis=1)), rot.copy(), P.copy()]),
                ("P only rounded", [nom.copy(), rot.copy(), np.ascontiguousarray(r32(P))]),
                ("all rounded", Bq)):
    a = [x.copy() for x in A]
    orc.correct(a[0], a[1], a[2], prev.copy(), ids, pos, quat, 1)
    s = [x.copy() for x in S]
    orc.correct(s[0], s[1], s[2], prev.copy(), ids, pos, quat, 1)
    e = parity_errors((s[0], s[1], s[2], prev), (a[0], a[1], a[2], prev))
    print(f"{name:24s} literal {e['literal']:.2e} sigma {e['sigma']:.2e} ({e['sigma_block']}) plain {e['plain']:.2e} ({e['plain_block']}) cov_block {e['cov_block']:.2e}")
