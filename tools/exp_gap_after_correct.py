#!/usr/bin/env python3
"""Experiment: is the slow first predict after a correct (15.8 us against 12.2 us steady) caused by work the correct
kernel leaves running in the background (then a small unrelated kernel in between should absorb it) or by the cache
state it leaves behind (then it should not)?  Run under `rocprofv3 --kernel-trace` and reduce with
tools/trace_positions.py.   usage: exp_gap_after_correct.py [gap|nogap]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
gap = len(sys.argv) > 1 and sys.argv[1] == "gap"
B, M, K = 65536, 4, 7
prm = capi.default_params(0)
nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
dev = torch.device("cuda:0")
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
pool = []
for s in range(4):
    acc, gyr = synth.imu_samples(0, B, s * K, K, nom)
    ids, pos, quat = synth.marker_frame(0, B, s, M, nom, prm)
    pool.append((f32(acc), f32(gyr), torch.from_numpy(ids).to(dev), f32(pos), f32(quat)))
d_dt = f32(np.full(K, 0.005))
c = np.load(os.path.join(ROOT, "tests", "golden", "vision_water.npz"))["corners"][:256]
d_l, d_r = f32(c[:, 2:10]), f32(c[:, 10:18])
with BatchedFilter(B, prm) as flt:
    flt.set_state(nom, rot, P, prev)
    for f in range(60):
        a, g, i, p, q = pool[f % 4]
        for k in range(K):
            flt.predict(a[k], g[k], d_dt[k:k + 1])
        flt.correct(i, p, q, 1)
        if gap:
            flt.marker_pose(d_l, d_r)
    flt.sync()
    flt._keep.clear()
print("done", "gap" if gap else "nogap")
