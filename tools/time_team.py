#!/usr/bin/env python3
"""Team kernels (several waves per 64-filter tile) against the one-wave kernels, launch by launch (HIP-event bracket per launch,
which adds ~2-3 us to every figure).  python tools/time_team.py [B ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fbus-ekf_amd"))
import torch
from fbus_ekf import BatchedFilter, capi, synth

dev = torch.device("cuda:0")
prm = capi.default_params(0)
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.float32)


def timed(flt, kind, fn, reps):
    for _ in range(5):
        fn()
    flt.sync()
    flt.timing_enable(True); flt.timing_reset()
    for _ in range(reps):
        fn()
    ms, n = flt.timing_read(kind)
    flt.timing_enable(False)
    return ms / n * 1e3


Bs = [int(x) for x in sys.argv[1:]] or [4096, 16384, 32768, 49152, 65536, 73728]
print("B        kernel                 " + "".join(f"roles={r:<8d}" for r in (1, 2, 3, 4)) + "   (us per launch)")
for B in Bs:
    nom, rot, P, prev = synth.initial_state(0, B, list(prm.p0_diag), 18)
    acc, gyr = synth.imu_samples(0, B, 0, 8, nom)
    d_acc, d_gyr, d_dt = up(acc), up(gyr), up(np.full(8, 0.005))
    ids, pos, quat = synth.marker_frame(0, B, 0, 4, nom, prm)
    d_ids, d_pos, d_quat = torch.from_numpy(ids).to(dev), up(pos), up(quat)
    rows = {"predict": [], "predict_n K=8": [], "correct stacked M=4": [], "correct nearest M=4": []}
    with BatchedFilter(B, prm, order_streams=False) as flt:
        flt.set_state(nom, rot, P, prev)
        for r in (1, 2, 3, 4):
            flt.set_team(r, r)
            rows["predict"].append(timed(flt, capi.KERNEL_PREDICT, lambda: flt.predict(d_acc[0], d_gyr[0], d_dt[:1]), 300))
            if r in (1, 4):
                rows["predict_n K=8"].append(timed(flt, capi.KERNEL_PREDICT_N, lambda: flt.predict_n(d_acc, d_gyr, d_dt), 100))
            else:
                rows["predict_n K=8"].append(float("nan"))
            rows["correct stacked M=4"].append(timed(flt, capi.KERNEL_CORRECT, lambda: flt.correct(d_ids, d_pos, d_quat, 1), 200))
            rows["correct nearest M=4"].append(timed(flt, capi.KERNEL_CORRECT, lambda: flt.correct(d_ids, d_pos, d_quat, 0), 200))
        # fused frame (K = 7 + stacked correct) and a window of 8 such frames: one-wave kernels (set_team 1) and the team kernel (4)
        kc = [7] * 8
        acc8, gyr8 = synth.imu_samples(0, B, 0, 56, nom)
        w_acc, w_gyr, w_dt = up(acc8), up(gyr8), up(np.full(56, 0.005))
        fr = [synth.marker_frame(0, B, f, 4, nom, prm) for f in range(8)]
        w_ids = torch.from_numpy(np.stack([x[0] for x in fr])).to(dev)
        w_pos, w_quat = up(np.stack([x[1] for x in fr])), up(np.stack([x[2] for x in fr]))
        rows["fused frame K=7 M=4"], rows["frame window 8x(7+1)"] = [], []
        for r in (1, 2, 3, 4):
            if r in (1, 4):
                flt.set_team(r, 1)
                rows["fused frame K=7 M=4"].append(timed(flt, capi.KERNEL_FRAME, lambda: flt.frame(d_acc[:7], d_gyr[:7], d_dt[:7], d_ids, d_pos, d_quat, 1, fused=True), 60))
                rows["frame window 8x(7+1)"].append(timed(flt, capi.KERNEL_FRAME, lambda: flt.frames(kc, w_acc, w_gyr, w_dt, w_ids, w_pos, w_quat, 1), 20))
            else:
                rows["fused frame K=7 M=4"].append(float("nan")); rows["frame window 8x(7+1)"].append(float("nan"))
    for name, v in rows.items():
        print(f"{B:<8d} {name:<22s} " + "".join(f"{x:<14.2f}" for x in v))
