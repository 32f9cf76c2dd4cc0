#!/usr/bin/env python3
"""Experiment: the bench pattern on ONE handle of B filters vs S handles of B/S filters on S streams (independent
launch chains whose kernel boundaries and load/compute/store phases interleave).  Prints EKF steps/s for each S."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth
PATTERN = (7, 7, 6)
B = int(os.environ.get("B", 65536)); STEPS = 60
dev = torch.device("cuda:0")
prm = capi.default_params(0)
f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
for S in (1, 2, 4, 1, 2, 4):
    hs = []
    for s in range(S):
        lo, hi = s * B // S, (s + 1) * B // S
        nom, rot, P, prev = synth.initial_state(lo, hi, list(prm.p0_diag), 18)
        acc, gyr = synth.imu_samples(lo, hi, 0, 20, nom)
        frames = [synth.marker_frame(lo, hi, f, 4, nom, prm) for f in range(3)]
        flt = BatchedFilter(hi - lo, prm)
        flt.set_state(nom, rot, P, prev)
        hs.append((flt, f32(acc), f32(gyr), [(torch.from_numpy(i).to(dev), f32(p), f32(q)) for i, p, q in frames]))
    d_dt = f32(np.full(7, 0.005))
    torch.cuda.synchronize()
    def step():
        k = 0
        for f, K in enumerate(PATTERN):
            for flt, acc, gyr, frames in hs:
                ids, pos, quat = frames[f]
                flt.frame(acc[k:k + K], gyr[k:k + K], d_dt[:K], ids, pos, quat, 1)
            k += K
    for _ in range(5): step()
    torch.cuda.synchronize()
    for h in hs: h[0]._keep.clear()
    t0 = time.perf_counter()
    for _ in range(STEPS): step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"B {B} streams {S}: {B * 23 * STEPS / el:.4g} EKF steps/s, {el / STEPS * 1e3:.4f} ms per bench step", flush=True)
    for h in hs: h[0].close()
