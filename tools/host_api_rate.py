#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-pointer entry points (fbus_ekf_predict / fbus_ekf_correct):
every call stages its inputs H2D and synchronises.  Informational; never bench `value`."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
from fbus_ekf import BatchedFilter, capi, synth

B, M = 65536, 4
prm = capi.default_params(0)
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
acc, gyr = synth.imu_samples(0, B, 0, 20, nom)
acc, gyr = acc.astype(np.float32), gyr.astype(np.float32)
ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
pos, quat = pos.astype(np.float32), quat.astype(np.float32)
dt = np.array([0.005], np.float32)
with BatchedFilter(B, prm) as flt:
    flt.set_state(nom, rot, P, prev)
    for k in range(3):
        flt.predict(acc[k], gyr[k], dt)
    flt.correct(ids, pos, quat, 1)
    t0 = time.perf_counter()
    for rep in range(5):
        for k in range(20):
            flt.predict(acc[k], gyr[k], dt)
    t1 = time.perf_counter()
    for rep in range(20):
        flt.correct(ids, pos, quat, 1)
    t2 = time.perf_counter()
    t3 = time.perf_counter()
    flt.predict_n(acc, gyr, np.full(20, 0.005, np.float32))
    t4 = time.perf_counter()
print(f"host-pointer predict : {(t1-t0)/100*1e6:8.1f} us/call  -> {B*100/(t1-t0):.3g} EKF steps/s (PCIe + sync inclusive)")
print(f"host-pointer correct : {(t2-t1)/20*1e6:8.1f} us/call  -> {B*20/(t2-t1):.3g} EKF steps/s")
print(f"host-pointer predict_n(K=20): {(t4-t3)*1e6:8.1f} us/call -> {B*20/(t4-t3):.3g} EKF steps/s")
