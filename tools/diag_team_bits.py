"""which covariance elements of the team predict differ from the one-wave kernel's (bit level)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
from fbus_ekf import BatchedFilter, capi, synth
r32 = lambda a: np.asarray(a, np.float64).astype(np.float32).astype(np.float64)
B = 311
for dialect in (0, 1):
    prm = capi.default_params(dialect)
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18, mixed_cov=True)
    nom, rot, P = r32(nom), r32(rot), r32(P)
    acc, gyr = synth.imu_samples(0, B, 0, 1, nom)
    acc, gyr = r32(acc), r32(gyr)
    DT = np.array([np.float64(np.float32(0.005))])
    out = {}
    for roles in (1, 2, 3, 4):
        with BatchedFilter(B, prm) as flt:
            flt.set_team(roles, 1)
            flt.set_state(nom, rot, P, prev)
            flt.predict(acc[0], gyr[0], DT)
            out[roles] = flt.get_state()
    for roles in (2, 3, 4):
        d = out[roles][2] != out[1][2]
        nz = np.argwhere(d.any(axis=0))
        print("dialect", dialect, "roles", roles, "nominal equal", np.array_equal(out[roles][0], out[1][0]),
              "differing (i,j):", sorted({(min(i, j), max(i, j)) for i, j in nz}), "filters", int(d.any(axis=(1, 2)).sum()))
        if len(nz):
            i, j = nz[0]
            b = int(np.argwhere(d[:, i, j])[0][0])
            print("   e.g. filter", b, (i, j), repr(out[roles][2][b, i, j]), repr(out[1][2][b, i, j]))
