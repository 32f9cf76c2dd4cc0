#!/usr/bin/env python3
"""Host-side cost of the per-call path with a launch-bound batch (B = 64): per frame through the Python wrapper, per
frame through ctypes with pre-extracted pointers, and per launch inside the C ABI (one frame_dev call with K = 64)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
B, M, K = 64, 4, 64
prm = capi.default_params(0)
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
acc, gyr = synth.imu_samples(0, B, 0, K, nom)
ids, pos, quat = synth.marker_frame(0, B, 0, M, nom, prm)
dev = torch.device("cuda:0")
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
d_acc, d_gyr, d_dt = f32(acc), f32(gyr), f32(np.full(K, 0.005))
d_ids, d_pos, d_quat = torch.from_numpy(ids).to(dev), f32(pos), f32(quat)
with BatchedFilter(B, prm) as flt:
    flt.set_state(nom, rot, P, prev)
    def timed(fn, n):
        for _ in range(20): fn()
        flt.sync(); t0 = time.perf_counter()
        for _ in range(n): fn()
        t1 = time.perf_counter(); flt.sync(); t2 = time.perf_counter()
        return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
    a7, g7, dt7 = d_acc[:7], d_gyr[:7], d_dt[:7]
    h, w = timed(lambda: flt.frame(a7, g7, dt7, d_ids, d_pos, d_quat, 1), 2000)
    print(f"python wrapper, frame of 7 predicts + correct : host {h:7.1f} us  wall {w:7.1f} us per frame = {w / 8:5.2f} us per launch")
    lib, hnd, p = flt._lib, flt._h, flt._p
    args = (hnd, 7, p(a7), p(g7), p(dt7), 0, M, p(d_ids), p(d_pos), p(d_quat), 1, None)
    h, w = timed(lambda: lib.fbus_ekf_frame_dev(*args), 2000)
    print(f"ctypes, pointers pre-extracted, same frame     : host {h:7.1f} us  wall {w:7.1f} us per frame = {w / 8:5.2f} us per launch")
    args = (hnd, K, p(d_acc), p(d_gyr), p(d_dt), 0, M, p(d_ids), p(d_pos), p(d_quat), 1, None)
    h, w = timed(lambda: lib.fbus_ekf_frame_dev(*args), 300)
    print(f"ctypes, one call with K = {K} predicts + correct : host {h:7.1f} us  wall {w:7.1f} us per call  = {w / (K + 1):5.2f} us per launch")
    flt._keep.clear()
